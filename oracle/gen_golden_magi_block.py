#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/magi_block_tiny.npz and magi_block_real.npz by running the REFERENCE's own
`TransformerLayer` (inferix/models/magi/dit/dit_module.py:1201-1319: FullyParallelAttention + gate + MLP) on CPU, one
rank, cp_strategy "none", with the reference's own `MagiKVCacheManager` / `KVCacheManager` behind it.  The third-party
calls the module makes (flash-attn rotary / attention, flashinfer) and the Triton range_mod kernel are stood in for as
documented in oracle/_refstub.py (`install_magi`, `import_magi_dit`).  Only data is stored: inputs, outputs, a few
intermediate tensors; weights are regenerated from a seed by `magi_block_oracle.init_layer_weights`.

  tiny : 2 stacked layers (hidden 256, 4 q-heads on 2 kv-heads, head_dim 128), four forwards that walk the cache rule —
         store two chunks / read a prefix + store under the nearly-clean rule / read-only with a key window / no cache —
         with two denoising ranges and two caption segments.
  real : ONE layer at MAGI-4.5B dimensions (hidden 3072, 24 q-heads on 8 kv-groups, ffn 12288), 2 x 48 tokens, two forwards.

After writing, the restatement (oracle/magi_block_oracle.py) is checked against every stored tensor, bit for bit.
usage (build container only; /root/reference must exist):  python oracle/gen_golden_magi_block.py
"""
from __future__ import annotations

import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstub  # noqa: E402
import magi_block_oracle as MB  # noqa: E402
from fixture_io import GOLDEN_DIR, load_npz, save_npz  # noqa: E402

BF = torch.bfloat16


def scenario(cfg: MB.MagiLayerConfig, clip: int, caps, seed: int):
    """The forwards of one fixture: (inputs, LayerMeta) per call."""
    g = torch.Generator().manual_seed(seed)
    hd = cfg.kv_channels
    calls = []

    def mk(n_ranges, meta_kw, k_ranges):
        s = n_ranges * clip
        y_tokens = sum(caps[:n_ranges])
        cu_kv = [0]
        for c in caps[:n_ranges]:
            cu_kv.append(cu_kv[-1] + c)
        inp = dict(x=torch.randn(s, 1, cfg.hidden_size, generator=g).to(BF),
                   condition=torch.randn(1, n_ranges, cfg.cond_size, generator=g).to(BF),
                   condition_map=(torch.arange(s, dtype=torch.int32) // clip).reshape(s, 1),
                   y=torch.randn(y_tokens, cfg.xattn_size, generator=g).to(BF),
                   rope=torch.cat([torch.sin(r := torch.rand(s, hd // 2, generator=g) * 6.0), torch.cos(r)], dim=-1))
        meta = MB.LayerMeta(q_ranges=[(i * clip, (i + 1) * clip) for i in range(n_ranges)], k_ranges=k_ranges,
                            cu_seqlens_q=[i * clip for i in range(n_ranges + 1)], cu_seqlens_kv=cu_kv, clip_token_nums=clip,
                            **meta_kw)
        calls.append((inp, meta))
    c = clip
    mk(2, dict(slice_point=0, update_kv_cache=True, use_cache=True), [(0, c), (0, 2 * c)])                    # fwd_extra_1st_chunk
    mk(2, dict(slice_point=1, update_kv_cache=True, use_cache=True, distill_nearly_clean_chunk=True), [(0, 2 * c), (0, 3 * c)])
    mk(1, dict(slice_point=2, update_kv_cache=False, use_cache=True), [(c, 3 * c)])                           # window: skip clip 0
    mk(1, dict(slice_point=0, update_kv_cache=False, use_cache=False), [(0, c)])
    return calls


def run_reference(cfg: MB.MagiLayerConfig, n_layers: int, calls, wseed: int, max_tokens: int, fp8: bool = False):
    import importlib
    dm = _refstub.import_magi_dit()
    cfgm = importlib.import_module("inferix.core.config")
    types = importlib.import_module("inferix.core.types.inference")
    kvm = importlib.import_module("inferix.kvcache_manager.kvcache_manager")
    mc = cfgm.ModelConfig(model_name="golden", num_layers=max(n_layers, 3), hidden_size=cfg.hidden_size,
                          ffn_hidden_size=cfg.ffn_hidden_size, num_attention_heads=cfg.num_attention_heads,
                          num_query_groups=cfg.num_query_groups, kv_channels=cfg.kv_channels,
                          layernorm_epsilon=cfg.layernorm_epsilon, apply_layernorm_1p=cfg.apply_layernorm_1p,
                          params_dtype=BF, cond_hidden_ratio=cfg.cond_hidden_ratio,
                          xattn_cond_hidden_ratio=cfg.xattn_cond_hidden_ratio, cond_gating_ratio=cfg.cond_gating_ratio,
                          gated_linear_unit=cfg.gated_linear_unit)
    ec = cfgm.EngineConfig(cp_size=1, cp_strategy="none", fp8_quant=fp8, kv_offload=False)
    layers = []
    for li in range(n_layers):
        layer = dm.TransformerLayer(mc, ec, layer_number=li)
        for name, sub in layer.named_modules():                      # _high_precision_promoter (dit_model.py:620-637)
            if "_xattn" in name:
                continue
            if any(t in name for t in ("q_layernorm", "k_layernorm", "self_attn_post_norm", "mlp_post_norm")):
                sub.float()
        W = MB.init_layer_weights(cfg, wseed + li, fp8=fp8 and MB.layer_is_fp8(li, max(n_layers, 3)))
        missing = layer.load_state_dict(W, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        for n, p in layer.named_parameters():
            assert p.dtype == W[n].dtype, (n, p.dtype, W[n].dtype)
        layers.append(layer.eval())
    ip = object.__new__(types.InferenceParams)
    ip.max_sequence_length, ip.max_batch_size, ip.sequence_len_offset = max_tokens, 1, 0
    ip.kv_cache_request = kvm.KVCacheRequest(request_id="magi")
    ip.kv_cache_manager = kvm.KVCacheManager(device="cpu")
    ip.key_value_memory_dict, ip.update_kv_cache = {}, False
    outs = []
    import numpy as np
    for inp, m in calls:
        qr = torch.tensor(m.q_ranges, dtype=torch.int32)
        kr = torch.tensor(m.k_ranges, dtype=torch.int32)
        core = types.PackedCoreAttnParams(q_range=qr, k_range=kr, np_q_range=qr.numpy(), np_k_range=kr.numpy(),
                                          max_seqlen_q=m.clip_token_nums, max_seqlen_k=int(kr[:, 1].max()))
        cross = types.PackedCrossAttnParams(q_ranges=None, kv_ranges=None,
                                            cu_seqlens_q=torch.tensor(m.cu_seqlens_q, dtype=torch.int32),
                                            cu_seqlens_kv=torch.tensor(m.cu_seqlens_kv, dtype=torch.int32),
                                            max_seqlen_q=m.clip_token_nums,
                                            max_seqlen_kv=int(np.diff(m.cu_seqlens_kv).max()))
        meta = types.ModelMetaArgs(H=1, W=1, cp_pad_size=0, cp_split_sizes=None, slice_point=m.slice_point,
                                   denoising_range_num=len(m.q_ranges), range_num=len(m.q_ranges) + m.slice_point,
                                   extract_prefix_video_feature=False,
                                   fwd_extra_1st_chunk=m.use_cache and m.slice_point == 0,
                                   distill_nearly_clean_chunk=m.distill_nearly_clean_chunk, clip_token_nums=m.clip_token_nums,
                                   enable_cuda_graph=False, core_attn_params=core, cross_attn_params=cross)
        ip.update_kv_cache = m.update_kv_cache
        x = inp["x"]
        per_layer = []
        with torch.no_grad():
            for layer in layers:
                x = layer(x, inp["condition"], inp["condition_map"], inp["y"], inp["rope"], ip, meta)
                per_layer.append(x)
        outs.append(per_layer)
    caches = [ip.kv_cache_manager.get_raw(ip.kv_cache_request, f"layer_{li}").clone() for li in range(n_layers)]
    return outs, caches


def run_oracle(cfg, n_layers, calls, wseed, max_tokens, want_taps=False, fp8=False):
    Ws = [MB.init_layer_weights(cfg, wseed + li, fp8=fp8 and MB.layer_is_fp8(li, max(n_layers, 3))) for li in range(n_layers)]
    caches = [MB.MagiLayerCache(max_tokens, cfg.num_query_groups, cfg.kv_channels) for _ in range(n_layers)]
    outs, taps_all = [], []
    for inp, m in calls:
        x = inp["x"]
        per_layer, taps_call = [], []
        for W, cache in zip(Ws, caches):
            taps = {} if want_taps else None
            x = MB.layer_forward(W, cfg, x, inp["condition"], inp["condition_map"], inp["y"], inp["rope"], m, cache, taps)
            per_layer.append(x)
            taps_call.append(taps)
        outs.append(per_layer)
        taps_all.append(taps_call)
    return outs, caches, taps_all


def meta_tensor(m: MB.LayerMeta) -> dict:
    return dict(q_ranges=torch.tensor(m.q_ranges), k_ranges=torch.tensor(m.k_ranges), cu_q=torch.tensor(m.cu_seqlens_q),
                cu_kv=torch.tensor(m.cu_seqlens_kv),
                flags=torch.tensor([m.clip_token_nums, m.slice_point, int(m.update_kv_cache), int(m.use_cache),
                                    int(m.distill_nearly_clean_chunk)]))


TAPS = ("q", "k", "v", "core", "xattn", "proj", "gate", "attn_res", "mlp")


def build(name: str, cfg: MB.MagiLayerConfig, n_layers: int, clip: int, caps, seed: int, wseed: int, n_calls: int,
          tap_names=TAPS, fp8: bool = False):
    calls = scenario(cfg, clip, caps, seed)[:n_calls]
    max_tokens = 4 * clip
    ref_outs, ref_caches = run_reference(cfg, n_layers, calls, wseed, max_tokens, fp8)
    orc_outs, orc_caches, taps = run_oracle(cfg, n_layers, calls, wseed, max_tokens, want_taps=True, fp8=fp8)
    fx = {"geom": torch.tensor([cfg.hidden_size, cfg.ffn_hidden_size, cfg.num_attention_heads, cfg.num_query_groups,
                                cfg.kv_channels, n_layers, clip, len(calls), wseed, max_tokens]),
          "fp8_quant": torch.tensor(int(fp8))}
    written = 0
    for ci, (inp, m) in enumerate(calls):
        for k, v in inp.items():
            fx[f"c{ci}_in_{k}"] = v
        for k, v in meta_tensor(m).items():
            fx[f"c{ci}_meta_{k}"] = v
        for li in range(n_layers):
            fx[f"c{ci}_out_l{li}"] = ref_outs[ci][li]
            assert torch.equal(ref_outs[ci][li], orc_outs[ci][li]), (name, "oracle != reference", ci, li)
        # intermediate tensors of layer 0 come from the restatement, which the line above has just pinned to the reference
        for t in (tap_names if ci == 0 or tap_names is TAPS else ()):
            fx[f"c{ci}_tap_{t}"] = taps[ci][0][t]
        if m.update_kv_cache:
            n = inp["x"].shape[0]
            written = max(written, m.slice_point * clip + (n - clip if m.distill_nearly_clean_chunk else n))
    for li in range(n_layers):
        c = ref_caches[li].clone()                          # (2, tokens, 1, hn, hd); allocated with torch.empty upstream:
        c[:, written:] = 0                                  # only the stored extent is defined
        fx[f"cache_l{li}"] = c
        assert torch.equal(c[0, :written, 0], orc_caches[li].k[:written]) and torch.equal(c[1, :written, 0], orc_caches[li].v[:written])
    fx["cache_written"] = torch.tensor(written)
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    save_npz(path, fx)
    print(f"wrote {path} ({os.path.getsize(path)} bytes); oracle == reference on every output and cache row (bit-exact)")


def main():
    if not _refstub.available():
        raise SystemExit("reference tree not available: fixtures can only be generated in the build container")
    warnings.filterwarnings("ignore")
    torch.manual_seed(0)
    build("magi_block_tiny", MB.tiny_config(), n_layers=2, clip=24, caps=(7, 5), seed=11, wseed=500, n_calls=4)
    # 2 x 48 tokens: not a multiple of the 64-row kernel tiles; two intermediate tensors only (bf16 noise does not compress)
    build("magi_block_real", MB.MagiLayerConfig(), n_layers=1, clip=48, caps=(40, 25), seed=12, wseed=600, n_calls=2,
          tap_names=("core", "attn_res"))
    # engine_config.fp8_quant (the 4.5B distill-quant config): three layers, the middle one on the static-scale FP8 linears
    build("magi_block_fp8_tiny", MB.tiny_config(), n_layers=3, clip=24, caps=(7, 5), seed=13, wseed=700, n_calls=3,
          tap_names=("proj", "mlp"), fp8=True)


if __name__ == "__main__":
    main()
