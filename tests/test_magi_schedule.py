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


def test_run_loop_with_a_prefix_video():
    """Video continuation (one whole prefix chunk) and image-to-video (one prefix frame): the prefix chunks go through the model once at
    clean_t with the null caption and `extract_prefix_video_feature`; afterwards every forward that overlaps prefix frames carries them
    instead of its noise, at t = 1 where a whole chunk is covered (video_generate.py:391-454)."""
    from inferix_amd.magi import schedule as S
    import pytest
    cw = 2
    x0 = torch.randn(1, 16, 8, 4, 6)
    x = torch.cat([x0, x0], 0)
    y = torch.randn(2, 4, 5, 8)
    masks = torch.ones(2, 4, 5)
    prefix = torch.full((2, 16, 2, 4, 6), 7.0)
    m = _Recorder()
    kept = []
    orig = m.forward_dispatcher
    m.forward_dispatcher = lambda **kw: (kept.append(kw["x"].clone()), orig(**kw))[1]
    sch = S.ChunkSchedule(8, 4, 4, cw, chunk_offset=1)
    out = sch.run(m, x.clone(), y, masks, inference_params=None, prefix_video=prefix)
    assert len(m.calls) == 1 + sch.total_forward_step() == 1 + 2 * 6
    shape, t, ys, ms, kv, kw = m.calls[0]                              # the extraction pass
    assert kw["extract_prefix_video_feature"] and kw["denoising_range_num"] == kw["range_num"] == 1 and kw["slice_point"] == 0
    assert shape == (2, 16, 2, 4, 6) and torch.all(kept[0] == 7.0) and torch.allclose(t, torch.full((2, 1), 0.9999))
    assert kv.tolist() == [[0, cw * 2 * 3]]
    # stage 0 of a schedule with offset 1 denoises chunk 1 only; nothing overlaps the prefix until a clean chunk 0 would be needed — it is
    # never re-forwarded (chunk_start > chunk_offset is what asks for the extra chunk): no later forward holds prefix frames
    assert all(not torch.any(k == 7.0) for k in kept[1:])
    assert all(c[5]["slice_point"] >= 1 for c in m.calls[1:])
    assert torch.equal(out[:, :, :cw], x[:, :, :cw]), "the prefix chunk's noise is not touched"
    assert torch.allclose(out[:, :, cw:], x[:, :, cw:] + 1.0, atol=1e-5)
    with pytest.raises(ValueError):
        S.ChunkSchedule(8, 4, 4, cw).run(m, x.clone(), y, masks, inference_params=None, prefix_video=prefix)
    # image-to-video: a single prefix frame, no whole chunk -> no extraction pass, the frame is pasted into every forward of chunk 0
    m2, kept2 = _Recorder(), []
    orig2 = m2.forward_dispatcher
    m2.forward_dispatcher = lambda **kw: (kept2.append((kw["x"].clone(), kw["timestep"].clone(), kw["slice_point"])), orig2(**kw))[1]
    sch2 = S.ChunkSchedule(8, 4, 4, cw)
    sch2.run(m2, x.clone(), y, masks, inference_params=None, prefix_video=prefix[:, :, :1])
    assert len(kept2) == sch2.total_forward_step()
    for xk, tk, sp in kept2:
        assert bool(torch.all(xk[:, :, 0] == 7.0)) == (sp == 0) and not torch.any(xk[:, :, 1:] == 7.0)
        assert not torch.any(tk == 1.0)                                # half a chunk is not a clean chunk
