"""CPU: the key ranges MAGI's chunk scheduler gives the attention layer (inferix_amd/magi/kv_ranges.py) against a golden the
reference's own `SampleTransport` methods produced (oracle/gen_golden_magi_kvrange.py) — bit-exact integer work."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
from fixture_io import golden  # noqa: E402


def test_kv_ranges_match_reference_golden():
    from inferix_amd.magi import kv_ranges as KR
    fx = golden("magi_kvrange.npz")
    n = int(fx["n_cases"])
    assert n >= 7
    for i in range(n):
        cw, lh, lw, patch, num_steps, clean, sp, dn = [int(v) for v in fx[f"c{i}_args"].tolist()]
        n2c = [int(v) for v in fx[f"c{i}_n2c"].tolist()]
        steps = [int(v) for v in fx[f"c{i}_steps"].tolist()]
        tokens = KR.chunk_token_nums(cw, lh, lw, patch)
        assert tokens == int(fx[f"c{i}_tokens"])
        got = KR.generate_kvrange_for_denoising_video(tokens, sp, dn, steps, num_steps, n2c, clean)
        assert got.dtype == torch.int32 and torch.equal(got, fx[f"c{i}_denoising"].to(torch.int32)), i
        assert torch.equal(KR.generate_default_kvrange(tokens, sp, dn), fx[f"c{i}_default"].to(torch.int32)), i
        assert torch.equal(KR.generate_kvrange_for_prefix_video(tokens, sp + dn, n2c, clean), fx[f"c{i}_prefix"].to(torch.int32)), i


def test_bench_leg_ranges_are_the_schedule_of_the_named_config():
    """bench.py's `magi_cp8_emulated` leg attends [0, (2 + i) * 12150): that is the 4.5B distill config's schedule
    (noise2clean_kvrange [5, 4, 3, 2], clean_chunk_kvrange 1) for four chunks at steps 48 / 32 / 16 / 0 of 64 behind one clean chunk."""
    from inferix_amd.magi import kv_ranges as KR
    tokens = KR.chunk_token_nums(6, 90, 90, 2)
    assert tokens == 12150
    kr = KR.generate_kvrange_for_denoising_video(tokens, 1, 4, [48, 32, 16, 0], 64, [5, 4, 3, 2], 1)
    assert kr.tolist() == [[0, (2 + i) * 12150] for i in range(4)]
