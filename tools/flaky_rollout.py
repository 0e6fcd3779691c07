#!/usr/bin/env python3
"""Determinism probe: repeat the tiny local-attention rollout in roll / paged mode and count distinct outputs."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import test_hip_model as T
import wan_oracle as O
cfg = O.tiny_config(local_attn_size=6, sink_size=1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for mode in ("roll", "page"):
    hs = []
    for i in range(n):
        out, _ = T._run_rollout("rollout_tiny_local.npz", cfg, paging=cfg.frame_seqlen if mode == "page" else None)
        hs.append(hashlib.md5(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:8])
    print(mode, hs)
