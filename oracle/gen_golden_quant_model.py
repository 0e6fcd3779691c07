"""TEST INFRASTRUCTURE — fixture of the QUANTISED model (BASELINE config 4): tests/golden/quant_model_tiny.npz.

The reference delegates the 8-bit linears to DAX (`quantize_dynamic(transformer, qconfig_dict)`,
example/quantization/run_self_forcing_quantized.py:47-64), which is not in /root/reference: no reference run can produce
these outputs here, so **parity with DAX stays unpinned**.  What this fixture pins is the WIRING: the restated model
(`oracle/wan_oracle.py`, bit-exact against the imported reference on every bf16 golden) with `oracle/quant_oracle.py`'s
linear put at every nn.Linear the reference's exclusion dict leaves quantised (`quant_oracle.model_hook`).  Stored:
inputs, the rollout with the reference's bf16 SDPA (`out_*`) and with exact fp64 attention (`out_*_exact`, the yardstick of
the floor rule), the per-call list of (linear name, format) the hook saw, and one real-geometry block (dim 1536, ffn 8960).
`tests/test_oracle_golden.py` re-runs the oracle against it on the CPU (regression pin); `tests/test_hip_quant.py` holds the
HIP model to it with the 1.25 x floor + 5e-4 rule.

    python oracle/gen_golden_quant_model.py
"""
from __future__ import annotations

import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "tests")]

import quant_oracle as Q  # noqa: E402
import wan_oracle as O  # noqa: E402
from fixture_io import golden, save_npz  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")
BF = torch.bfloat16
NAMES = {Q.FP8: "fp8", Q.INT8: "int8"}


def rollout_inputs():
    """The tiny rollout's own inputs (noise, prompt, drawn re-noise tensors, step list — produced by the reference run of
    gen_golden.py) on a model whose freq_dim is 128, so that every quantised linear has K % 128 == 0 (the HIP GEMM's constraint)."""
    fx = golden("rollout_tiny.npz")
    cfg = O.tiny_config(freq_dim=128)
    return fx, cfg, O.init_weights(cfg, seed=0)


def run_rollout(fx, cfg, W, fmt, attn_impl, log=None):
    renoise = [fx[f"renoise_{i}"] for i in range(int(fx["num_renoise"]))]
    with O.linear_override(Q.model_hook(Q.reference_qconfig_dict(fmt), log)):
        out, state = O.inference(W, cfg, fx["noise"], list(fx["prompt_embeds"]), fx["steps"].tolist(), renoise=renoise,
                                 shift=float(fx["shift"]), num_frame_per_block=3, attn_impl=attn_impl)
    return out, state


def block_inputs():
    fx = golden("block_real_dims.npz")
    cfg = O.WanConfig(num_layers=1, text_len=32, text_dim=64, freq_dim=64, latent_h=8, latent_w=12)
    return fx, cfg, O.init_weights(cfg, seed=3)


def run_block(fx, cfg, W, fmt, attn_impl):
    fs, nf = cfg.frame_seqlen, 3
    st = O.CacheState.allocate(cfg, 1, BF, cache_tokens=6 * fs)
    freqs = O.rope_freqs(cfg.head_dim)
    outs = []
    with O.linear_override(Q.model_hook(Q.reference_qconfig_dict(fmt))):
        for b in range(2):
            outs.append(O.block_forward(fx[f"x{b}"], fx[f"e0_{b}"], fx["context"], W, 0, cfg, (3, 4, 6), freqs, st, b * nf * fs,
                                        attn_impl=attn_impl))
    return outs, st


def main():
    fx, cfg, W = rollout_inputs()
    out = dict(steps=fx["steps"], shift=fx["shift"])
    for fmt, nm in NAMES.items():
        log = []
        q, st = run_rollout(fx, cfg, W, fmt, "sdpa", log)
        qx, stx = run_rollout(fx, cfg, W, fmt, "math")
        le = st.layers[0].local_end
        out[f"out_{nm}"], out[f"out_{nm}_exact"] = q, qx.to(BF)
        out[f"cache_k_layer0_{nm}"], out[f"cache_k_layer0_{nm}_exact"] = st.layers[0].k[0, :le], stx.layers[0].k[0, :le].to(BF)
        first = log[:log.index(("head.head", None)) + 1]          # the linears of ONE forward, in call order
        quantised = sorted({n for n, f in first if f is not None})
        kept = sorted({n for n, f in first if f is None})
        assert kept == ["head.head", "text_embedding.0", "text_embedding.2"], kept
        assert len(quantised) == cfg.num_layers * 10 + 3, quantised
        out[f"trace_{nm}"] = torch.tensor([[s.local_start, s.local_end, s.global_end] for s in st.trace])
        fl = float((q.double() - qx.double()).norm() / qx.double().norm())
        print(f"rollout {nm}: {len(quantised)} quantised linears per forward, kept {kept}; floor (sdpa vs exact attention) {fl:.3e}")
    out["quantised_names"] = torch.tensor([len(quantised)])
    bfx, bcfg, bW = block_inputs()
    for fmt, nm in NAMES.items():
        (o0, o1), st = run_block(bfx, bcfg, bW, fmt, "sdpa")
        (x0, x1), _ = run_block(bfx, bcfg, bW, fmt, "math")
        n = 3 * bcfg.frame_seqlen
        out[f"block_out0_{nm}"], out[f"block_out1_{nm}"] = o0, o1
        out[f"block_out0_{nm}_exact"], out[f"block_out1_{nm}_exact"] = x0.to(BF), x1.to(BF)
        out[f"block_cache_k_{nm}"], out[f"block_cache_v_{nm}"] = st.layers[0].k[0, :2 * n], st.layers[0].v[0, :2 * n]
        fl = [float((a.double() - b.double()).norm() / b.double().norm()) for a, b in ((o0, x0), (o1, x1))]
        print(f"real-dims block {nm}: floors {fl[0]:.3e} / {fl[1]:.3e}")
    save_npz(os.path.join(GOLDEN_DIR, "quant_model_tiny.npz"), out)


if __name__ == "__main__":
    main()
