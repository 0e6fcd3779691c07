#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/block_720p_full_size.npz: ONE CausalWanAttentionBlock of the reference's CausVid
model (inferix/models/causvid/causal_model.py:229-319, self-attention :128-179) at BASELINE config 3's size — 10800 tokens
(3 x 45 x 80), dim 1536, 12 heads, ffn 8960 — with the caller-named cache slots of the CausVid pipeline, over L = 10800 (first
block, slots [0, 10800)) and L = 75600 (seventh block, slots [64800, 75600)) keys, run on CPU from the reference import.
Inputs are seeded (oracle/block_720p_inputs.py); stored: 192 output rows, the K / V cache rows of those tokens, and — as the
noise-floor yardstick — the same rows with the block's self-attention evaluated exactly (fp64) for those queries.

This is the oracle row of the launches that only exist at this size: the 192-token split tile of the FFN down-projection
(10800 rows), the two-per-CU attention schedule over 75600 keys, the row kernels at 10800 x 1536.

usage (build container only; several minutes of CPU):  python oracle/gen_golden_block_720p.py
"""
from __future__ import annotations

import importlib
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstub  # noqa: E402
import block_720p_inputs as BI  # noqa: E402
import wan_oracle as O  # noqa: E402
from fixture_io import GOLDEN_DIR, save_npz, weights_checksum  # noqa: E402
from gen_golden import _pc, check  # noqa: E402

BF = torch.bfloat16
torch.set_grad_enabled(False)


def main():
    if not _refstub.available():
        raise SystemExit("reference tree not present — fixtures can only be generated in the build container")
    _refstub.import_hot_path()
    from inferix.kvcache_manager.kvcache_manager import KVCacheManager, KVCacheRequest
    cvm = importlib.import_module("inferix.models.causvid.causal_model")
    cfg = BI.config()
    W = O.init_weights(cfg, seed=3)
    m = cvm.CausalWanModel(model_type="t2v", patch_size=cfg.patch_size, text_len=cfg.text_len, in_dim=cfg.in_dim,
                           dim=cfg.dim, ffn_dim=cfg.ffn_dim, freq_dim=cfg.freq_dim, text_dim=cfg.text_dim,
                           out_dim=cfg.out_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers, qk_norm=True,
                           cross_attn_norm=True, eps=cfg.eps, enable_kv_offload=False, parallel_config=_pc()).eval()
    m.load_state_dict({k: v.float() for k, v in W.items()}, strict=True)
    m = m.to(BF)
    blk = m.blocks[0]
    fs = cfg.frame_seqlen
    n = BI.FRAMES * fs
    grid = (BI.FRAMES, cfg.latent_h // 2, cfg.latent_w // 2)
    freqs = O.rope_freqs(cfg.head_dim)
    sel = BI.SEL
    fx = dict(weights_checksum=torch.tensor(weights_checksum(W)), sel=sel)
    for case in (0, 1):
        d = BI.make(case)
        start = int(d["kv_start"])
        end = start + n
        cap = BI.BLOCKS * n
        kvm, req = KVCacheManager(device="cpu"), [KVCacheRequest("r")]
        blk.kv_cache_manager.allocate_kv_cache(kv_cache_manager=kvm, kv_cache_request=req[0], sequence_length=cap, dtype=BF)
        blk.kv_cache_manager.allocate_crossattn_cache(kv_cache_manager=kvm, kv_cache_request=req[0],
                                                      crossattn_length=cfg.text_len, dtype=BF)
        blk.is_cross_attn_init = False
        raw = kvm.get_raw(req[0], "layer_0")
        raw.zero_()
        state = O.CacheState.allocate(cfg, 1, BF, cache_tokens=cap)
        if start:
            raw[0, :start, 0], raw[1, :start, 0] = d["prefix_k"], d["prefix_v"]
            state.layers[0].k[0, :start], state.layers[0].v[0, :start] = d["prefix_k"], d["prefix_v"]
        t0 = time.time()
        ref = blk(d["x"], e=d["e0"], seq_lens=torch.tensor([n]), grid_sizes=torch.tensor([list(grid)]), freqs=m.freqs,
                  context=d["ctx"], context_lens=None, block_mask=None, kv_start=start, kv_end=end, current_start=start,
                  current_end=end, kv_cache_manager=kvm, kv_cache_requests=req)
        print(f"case {case}: reference CausVid block forward, L = {end}: {time.time() - t0:.1f} s", flush=True)
        t0 = time.time()
        mine = O.block_forward(d["x"], d["e0"], d["ctx"], W, 0, cfg, grid, freqs, state, start, explicit=(start, end))
        print(f"case {case}: oracle block forward {time.time() - t0:.1f} s", flush=True)
        check(f"720p full-size block #{case}", ref, mine)
        check(f"720p full-size cache K #{case}", raw[0, :end, 0], state.layers[0].k[0, :end])
        check(f"720p full-size cache V #{case}", raw[1, :end, 0], state.layers[0].v[0, :end])
        # noise floor: the same block with EXACT self-attention for the stored queries (everything after attention is token-local)
        state2 = O.CacheState.allocate(cfg, 1, BF, cache_tokens=cap)
        if start:
            state2.layers[0].k[0, :start], state2.layers[0].v[0, :start] = d["prefix_k"], d["prefix_v"]
        orig = O.attention

        def mixed(q, k, v, impl="sdpa"):
            out = orig(q, k, v, impl="sdpa")
            if q.shape[1] == n and k.shape[1] == end:                # the self-attention call
                out[:, sel] = orig(q[:, sel], k, v, impl="math").to(out.dtype)
            return out
        O.attention = mixed
        try:
            exact_rows = O.block_forward(d["x"], d["e0"], d["ctx"], W, 0, cfg, grid, freqs, state2, start, explicit=(start, end))[0, sel]
        finally:
            O.attention = orig
        floor = float((ref[0, sel].double() - exact_rows.double()).norm() / exact_rows.double().norm())
        print(f"case {case}: reference rows vs exact-attention rows rel-L2 {floor:.3e}", flush=True)
        fx.update({f"c{case}_out_rows": ref[0, sel], f"c{case}_exact_rows": exact_rows,
                   f"c{case}_k_rows": raw[0, start + sel, 0], f"c{case}_v_rows": raw[1, start + sel, 0],
                   f"c{case}_start": torch.tensor(start), f"c{case}_x_checksum": torch.tensor(BI.checksum(d["x"])),
                   f"c{case}_floor": torch.tensor(floor)})
        if start:
            fx[f"c{case}_prefix_checksum"] = torch.tensor(BI.checksum(d["prefix_k"]) ^ BI.checksum(d["prefix_v"]))
        kvm.free(req[0])
    path = os.path.join(GOLDEN_DIR, "block_720p_full_size.npz")
    save_npz(path, fx)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
