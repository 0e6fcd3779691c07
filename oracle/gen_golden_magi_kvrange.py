#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/magi_kvrange.npz: the key ranges MAGI's chunk scheduler hands to the attention
layer (`SampleTransport.generate_default_kvrange`, `generate_noise2clean_kvrange`, `generate_kvrange_for_prefix_video`,
inferix/pipeline/magi/video_generate.py:373-529), computed by the REFERENCE's own methods on CPU for a table of cases.  The
methods only read a handful of attributes, so the object is built without running the pipeline's constructor.

usage (build container only; /root/reference must exist):  python oracle/gen_golden_magi_kvrange.py
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

# (chunk_width, latent H, latent W, patch, num_steps, noise2clean_kvrange, clean_chunk_kvrange, slice_point, steps of each chunk)
CASES = [
    (6, 90, 90, 2, 64, [5, 4, 3, 2], 1, 1, [48, 32, 16, 0]),      # 4.5B distill config at 720 x 720: 12150 tokens per chunk
    (6, 90, 90, 2, 64, [5, 4, 3, 2], 1, 0, [32, 16, 0]),
    (6, 90, 90, 2, 64, [5, 4, 3, 2], -1, 3, [64, 48, 32, 16]),     # a finished chunk still in the window: clean_chunk_kvrange
    (6, 90, 90, 2, 64, [5, 4, 3, 2], 2, 6, [64, 63, 17, 15]),
    (4, 60, 104, 2, 12, [3, 2], 1, 2, [11, 6, 5, 0]),
    (6, 90, 160, 2, 16, [], -1, 2, [12, 8, 4, 0]),                 # no schedule: the default causal ranges
    (6, 90, 160, 2, 16, [], 4, 0, [4, 0]),
]


def main():
    if not _refstub.available():
        raise SystemExit("reference tree not available: fixtures can only be generated in the build container")
    _refstub.install_magi()
    import importlib
    import types
    # the package's __init__ pulls the whole MAGI pipeline (T5, VAE, transformers): register the package by path only, so that the
    # scheduler module and its relative import (`.prompt_process`) load without it
    import inferix.pipeline as _pl
    pkg = types.ModuleType("inferix.pipeline.magi")
    pkg.__path__ = [os.path.join(os.path.dirname(_pl.__file__), "magi")]
    sys.modules["inferix.pipeline.magi"] = pkg
    # prompt_process imports the T5 tokenizer stack (transformers probes the flash_attn stand-in and fails); the key-range methods
    # never touch its three helpers
    pp = types.ModuleType("inferix.pipeline.magi.prompt_process")
    pp.get_negative_special_token_keys = pp.get_special_token_keys = pp.pad_special_token = lambda *a, **k: None
    sys.modules["inferix.pipeline.magi.prompt_process"] = pp
    vg = importlib.import_module("inferix.pipeline.magi.video_generate")
    fx = {"n_cases": torch.tensor(len(CASES))}
    for i, (cw, lh, lw, patch, num_steps, n2c, clean, sp, steps) in enumerate(CASES):
        st = object.__new__(vg.SampleTransport)
        st.chunk_width = cw
        st.device = torch.device("cpu")
        st.model_config = SimpleNamespace(patch_size=patch)
        st.runtime_config = SimpleNamespace(noise2clean_kvrange=list(n2c), clean_chunk_kvrange=clean)
        st.transport_inputs = [SimpleNamespace(latent_size=(1, 16, 24, lh, lw), num_steps=num_steps)]
        dn = len(steps)
        fx[f"c{i}_args"] = torch.tensor([cw, lh, lw, patch, num_steps, clean, sp, dn])
        fx[f"c{i}_n2c"] = torch.tensor(n2c, dtype=torch.int64)
        fx[f"c{i}_steps"] = torch.tensor(steps)
        fx[f"c{i}_tokens"] = torch.tensor(st.get_batch_size_and_chunk_token_nums(0)[1])
        fx[f"c{i}_denoising"] = st.generate_kvrange_for_denoising_video(0, sp, dn, list(steps))
        fx[f"c{i}_default"] = st.generate_default_kvrange(0, sp, dn)
        fx[f"c{i}_prefix"] = st.generate_kvrange_for_prefix_video(0, sp + dn)
    path = os.path.join(GOLDEN_DIR, "magi_kvrange.npz")
    save_npz(path, fx)
    print("wrote", path, os.path.getsize(path), "bytes")
    for i in range(len(CASES)):
        print(i, fx[f"c{i}_tokens"].item(), fx[f"c{i}_denoising"].tolist())


if __name__ == "__main__":
    main()
