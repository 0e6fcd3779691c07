"""CPU: MAGI's chunk schedule (inferix_amd/magi/schedule.py) against a golden the reference's own `SampleTransport` methods produced
for every forward step of five clips (oracle/gen_golden_magi_schedule.py) — integers bit-exact, the time grid and the integration
deltas bit-exact in fp32 (same torch expressions)."""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
from fixture_io import golden  # noqa: E402

TCFG = [({}, "8,16,16"), ({"tSchedulerFunc": "square"}, ""), ({"tSchedulerFunc": "sd3", "shift": 2.0}, "16,16,8"),
        ({"tSchedulerFunc": "piecewise"}, "8,16,16"), ({"tSchedulerFunc": "identity"}, "")]


def test_schedule_matches_reference_golden_for_every_step():
    from inferix_amd.magi import schedule as S
    from inferix_amd.magi.kv_ranges import chunk_token_nums
    fx = golden("magi_schedule.npz")
    n = int(fx["n_cases"])
    assert n == len(TCFG)
    for c in range(n):
        num_steps, window, chunk_num, cw, off, lh, lw, clean = [int(v) for v in fx[f"c{c}_args"].tolist()]
        n2c = [int(v) for v in fx[f"c{c}_n2c"].tolist()]
        tcfg, shortcut = TCFG[c]
        t_total = S.init_t(tcfg, num_steps, "cpu", shortcut)
        assert torch.equal(t_total, fx[f"c{c}_t_total"]), c
        assert torch.equal(S.init_interval(num_steps, "cpu", shortcut).float(), fx[f"c{c}_interval"]), c
        assert torch.tensor(S.generate_sequences(chunk_num, window, off)).tolist() == fx[f"c{c}_sequences"].tolist(), c
        sch = S.ChunkSchedule(num_steps, window, chunk_num, cw, off)
        assert sch.total_forward_step() == int(fx[f"c{c}_total"])
        tokens = chunk_token_nums(cw, lh, lw, 2)
        status, lens = fx[f"c{c}_status"].tolist(), fx[f"c{c}_len"].tolist()
        at = 0
        kat = 0
        dat = 0
        for step, (row, ln) in enumerate(zip(status, lens)):
            p = sch.plan(step)
            got = [p.denoise_step_per_stage, p.denoise_stage, p.denoise_idx, off, p.chunk_start, p.chunk_end, p.t_start, p.t_end,
                   int(p.fwd_extra_1st_chunk), p.slice_point, p.denoising_range_num]
            assert got == [int(v) for v in row], (c, step)
            assert p.range_num == p.chunk_end and len(p.denoise_step_of_each_chunk) == ln == p.denoising_range_num
            assert p.denoise_step_of_each_chunk == [int(v) for v in fx[f"c{c}_steps_of"][at:at + ln].tolist()], (c, step)
            assert torch.equal(sch.timestep(t_total, p, 0.9999), fx[f"c{c}_t"][at:at + ln]), (c, step)
            kv = sch.kv_range(p, tokens, n2c, clean)
            assert torch.equal(kv, fx[f"c{c}_kv"][kat:kat + ln].to(torch.int32)), (c, step)
            nd = p.chunk_end - p.chunk_start
            dt = sch.timestep(t_total, p, 0.9999, advance=1) - t_total[p.t_index]
            assert torch.equal(dt, fx[f"c{c}_dt"][dat:dat + nd]), (c, step)
            at, kat, dat = at + ln, kat + ln, dat + nd
        assert at == fx[f"c{c}_t"].numel()


class _Recorder:
    """Stands in for the model: records what the loop hands over, returns a constant velocity."""

    def __init__(self):
        self.runtime_config = SimpleNamespace(clean_t=0.9999, noise2clean_kvrange=[5, 4, 3, 2], clean_chunk_kvrange=1, cfg_number=1)
        self.engine_config = SimpleNamespace(shortcut_mode="8,16,16", distill_nearly_clean_chunk_threshold=0.3)
        self.model_config = SimpleNamespace(patch_size=2)
        self.calls = []

    def forward_dispatcher(self, x, timestep, y, mask, kv_range, inference_params, **kw):
        self.calls.append((tuple(x.shape), timestep.clone(), tuple(y.shape), tuple(mask.shape), kv_range.clone(), dict(kw)))
        return torch.ones_like(x)


def test_run_loop_walks_every_chunk_from_noise_to_clean():
    """With velocity == 1 every chunk ends at x0 + (t[num_steps] - t[0]) = x0 + 1 after its 64 steps; the calls carry the extra
    clean chunk exactly at the first step of a stage whose window no longer starts at chunk 0."""
    from inferix_amd.magi import schedule as S
    sch = S.ChunkSchedule(64, 4, 4, 6)
    m = _Recorder()
    x0 = torch.randn(1, 16, 24, 4, 6)
    x = torch.cat([x0, x0], 0)
    y = torch.randn(2, 4, 5, 8)
    masks = torch.ones(2, 4, 5)
    out = sch.run(m, x.clone(), y, masks, inference_params=None)
    assert torch.allclose(out, x + 1.0, atol=1e-5)
    assert len(m.calls) == 112
    extras = [i for i, c in enumerate(m.calls) if c[5]["fwd_extra_1st_chunk"]]
    assert extras == [64, 80, 96]                                  # stages 4, 5, 6: windows starting at chunks 1, 2, 3
    shape, t, ys, ms, kv, kw = m.calls[64]
    assert shape == (2, 16, 24, 4, 6) and t.shape == (2, 4) and abs(t[0, 0].item() - 0.9999) < 1e-7
    assert ys == (8, 1, 5, 8) and ms == (8, 1, 5) and kv.shape == (4, 2)
    assert kw["slice_point"] == 0 and kw["denoising_range_num"] == 4 and kw["range_num"] == 4
    # chunk counts per stage: 1 2 3 4 (+clean) 3 (+clean) 2 (+clean) 1
    assert [m.calls[s * 16 + 1][0][2] // 6 for s in range(7)] == [1, 2, 3, 4, 3, 2, 1]
    assert [m.calls[s * 16][0][2] // 6 for s in range(7)] == [1, 2, 3, 4, 4, 3, 2]
    # the nearly-clean flag follows the oldest denoising chunk's time
    for shape, t, ys, ms, kv, kw in m.calls:
        assert kw["distill_nearly_clean_chunk"] == (t[0, int(kw["fwd_extra_1st_chunk"])].item() > 0.3)
