#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/magi_schedule.npz: MAGI's chunk schedule (`generate_sequences`, `init_t`,
`init_intervel`, `SampleTransport.generate_denoise_status_and_sequences`, `get_timestep`, `get_denoise_step_of_each_chunk`,
`total_forward_step`, `generate_kvrange_for_denoising_video`, `integrate`; inferix/pipeline/magi/video_generate.py:166-236,324-359,
513-585), computed by the REFERENCE's own functions on CPU for every forward step of a table of clips.  The methods read a handful of
attributes, so the transport object is built without running the pipeline's constructor (as gen_golden_magi_kvrange.py does).

usage (build container only; /root/reference must exist):  python oracle/gen_golden_magi_schedule.py
"""
from __future__ import annotations

import os
import sys
from types import SimpleNamespace

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstub  # noqa: E402
from fixture_io import GOLDEN_DIR, save_npz  # noqa: E402

# (num_steps, window_size, chunk_num, chunk_width, chunk_offset, latent H, latent W, noise2clean_kvrange, clean_chunk_kvrange,
#  t schedule config, shortcut_mode)
CASES = [
    (64, 4, 4, 6, 0, 90, 90, [5, 4, 3, 2], 1, {}, "8,16,16"),                      # example/magi/configs/4.5B distill: 96 frames at 720 x 720
    (64, 4, 7, 6, 1, 90, 160, [5, 4, 3, 2], -1, {"tSchedulerFunc": "square"}, ""),  # a prefix chunk in front
    (12, 4, 5, 6, 0, 60, 104, [3, 2], 1, {"tSchedulerFunc": "sd3", "shift": 2.0}, "16,16,8"),
    (12, 4, 2, 4, 0, 60, 104, [], -1, {"tSchedulerFunc": "piecewise"}, "8,16,16"),  # fewer chunks than the window
    (16, 4, 3, 6, 0, 30, 52, [], 4, {"tSchedulerFunc": "identity"}, ""),
]


def load_reference():
    _refstub.install_magi()
    import importlib
    import types
    import inferix.pipeline as _pl
    pkg = types.ModuleType("inferix.pipeline.magi")
    pkg.__path__ = [os.path.join(os.path.dirname(_pl.__file__), "magi")]
    sys.modules["inferix.pipeline.magi"] = pkg
    pp = types.ModuleType("inferix.pipeline.magi.prompt_process")
    pp.get_negative_special_token_keys = pp.get_special_token_keys = pp.pad_special_token = lambda *a, **k: None
    sys.modules["inferix.pipeline.magi.prompt_process"] = pp
    return importlib.import_module("inferix.pipeline.magi.video_generate")


def main():
    if not _refstub.available():
        raise SystemExit("reference tree not available: fixtures can only be generated in the build container")
    vg = load_reference()
    cpu = torch.device("cpu")
    fx = {"n_cases": torch.tensor(len(CASES))}
    for c, (num_steps, window, chunk_num, cw, off, lh, lw, n2c, clean, tcfg, shortcut) in enumerate(CASES):
        st = object.__new__(vg.SampleTransport)
        st.chunk_width, st.window_size, st.device = cw, window, cpu
        st.model_config = SimpleNamespace(patch_size=2)
        st.runtime_config = SimpleNamespace(noise2clean_kvrange=list(n2c), clean_chunk_kvrange=clean, clean_t=0.9999)
        prefix = torch.zeros(1, 16, off * cw, 2, 2) if off else None
        st.transport_inputs = [SimpleNamespace(latent_size=(1, 16, chunk_num * cw, lh, lw), num_steps=num_steps, chunk_num=chunk_num,
                                               prefix_video=prefix)]
        t_total = vg.init_t(dict(tcfg), num_steps, cpu, shortcut_mode=shortcut)
        fx[f"c{c}_args"] = torch.tensor([num_steps, window, chunk_num, cw, off, lh, lw, clean])
        fx[f"c{c}_n2c"] = torch.tensor(n2c, dtype=torch.int64)
        fx[f"c{c}_t_total"] = t_total
        fx[f"c{c}_interval"] = vg.init_intervel(num_steps, cpu, shortcut_mode=shortcut).to(torch.float32)
        seqs = vg.generate_sequences(chunk_num, window, off)
        fx[f"c{c}_sequences"] = torch.tensor(seqs)
        total = st.total_forward_step(0)
        fx[f"c{c}_total"] = torch.tensor(total)
        status, tcat, tlen, scat, kcat, dcat = [], [], [], [], [], []
        for step in range(total):
            (per, stage, idx), (o, cs, ce, ts, te) = st.generate_denoise_status_and_sequences(0, step)
            extra = cs > o and idx == 0
            sp, dn = (cs - 1, ce - cs + 1) if extra else (cs, ce - cs)
            steps_of = st.get_denoise_step_of_each_chunk(0, per, ts, te, idx, has_clean_t=extra)
            t = st.get_timestep(t_total, per, ts, te, idx, has_clean_t=extra)
            kv = st.generate_kvrange_for_denoising_video(infer_idx=0, slice_point=sp, denoising_range_num=dn,
                                                         denoise_step_of_each_chunk=steps_of)
            # integrate on a ramp: the per-chunk time deltas are all it adds
            xc = torch.zeros(1, 1, (ce - cs) * cw, 1, 1)
            dt = st.integrate(xc, torch.ones_like(xc), t_total, per, ts, te, idx)[0, 0, ::cw, 0, 0]
            status.append([per, stage, idx, o, cs, ce, ts, te, int(extra), sp, dn])
            tcat.append(t); tlen.append(len(t)); scat.append(torch.tensor(steps_of)); kcat.append(kv); dcat.append(dt)
        fx[f"c{c}_status"] = torch.tensor(status)
        fx[f"c{c}_len"] = torch.tensor(tlen)
        fx[f"c{c}_t"] = torch.cat(tcat)
        fx[f"c{c}_steps_of"] = torch.cat(scat)
        fx[f"c{c}_kv"] = torch.cat(kcat)
        fx[f"c{c}_dt"] = torch.cat(dcat)
        print(c, "forwards", total, "ranges", int(sum(tlen)))
    path = os.path.join(GOLDEN_DIR, "magi_schedule.npz")
    save_npz(path, fx)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
