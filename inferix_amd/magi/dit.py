"""MAGI transformer layer on MI355X — the denoise-step compute of BASELINE config 5 behind the reference's module names.

Mirrors `inferix/models/magi/dit/dit_module.py`:
  `TransformerLayer` (:1201-1319)            -> `HipMagiTransformerLayer`   (forward: same seven arguments)
  `FullyParallelAttention` (:833-1195)       -> `HipFullyParallelAttention` (`cp_strategy` "none", "cp_ulysses", "cp_shuffle_overlap")
  `TransformerBlock` (:1322-1390)            -> `HipMagiTransformerBlock`   (the layer stack + fp32 final LayerNorm)
State-dict keys are the reference's (`self_attention.linear_qkv.q.weight`, ...), loaded with `load_state_dict`.

Per layer, 17 kernel launches for one denoising range (every one an `ifx_*` entry of libinferix_hip.so, no torch compute):
  ifx_layernorm (affine)            linear_qkv.layer_norm                                               :415-416
  ifx_gemm_bf16                     q | qx | k | v in ONE GEMM over the concatenated weights             :418-431
  ifx_magi_head_prep                per-head LayerNorm (fp32 q/k, bf16 qx) + rotary + K/V written in place to the cache  :902-970
  ifx_attn_fwd_paged_ld x ranges    range attention, grouped-query heads, prefix read in place, output -> columns [0, Q) of
                                    the projection input                                                :972-1015
  ifx_gemm_bf16 + ifx_magi_head_prep   linear_kv_xattn + k_layernorm_xattn on the caption tokens         :959-970
  ifx_attn_fwd_paged_ld x segments  varlen cross-attention -> columns [Q, 2Q)                            :1047-1085
  ifx_gemm_bf16                     linear_proj; the "(n hn hd) -> (hn n hd)" interleave of :1287 is folded into a column
                                    permutation of the weight at load time, so the concatenation is never built
  ifx_act_rows, ifx_gemm_bf16, ifx_act_rows   SiLU -> AdaModulateLayer.proj -> softcap(tanh)  (tiny: one row per range) :196-198,:1300-1303
  ifx_magi_gate_norm_residual       range_mod gate x post-norm (fp32) + residual                        :295-313
  ifx_layernorm, ifx_gemm_bf16 (+ exact-GELU epilogue), ifx_gemm_bf16        CustomMLP                    :545-557
  ifx_magi_gate_norm_residual       the MLP half of the gate

Batch 1 only (as `core_attention`'s cached path upstream: 3-cfg folds ranges into the batch before it gets here).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from .. import _hip
from .. import hip_ops as ops
from ..quant import StaticFp8Linear
from . import context_parallel as cpl
from .attention import MagiKVCacheManager
from .types import InferenceParams, ModelMetaArgs

BF16 = torch.bfloat16
# self-attention queries leave ifx_magi_head_prep multiplied by softmax_scale * log2(e) (one rounding to bf16 either way) and the core
# attention is called with scale = ln 2: the same softmax(q k^T / sqrt(128)), on the exponent fast path of the attention kernel.  The
# cross-attention queries (qx) keep the reference's form.
Q_SCALE, ATTN_SCALE = ops.attn_q_prescale(128) if os.environ.get("IFX_MAGI_Q_PRESCALE", "1") != "0" else (0.0, 0.0)
import os as _os

_ULYSSES_FAST = _os.environ.get("IFX_MAGI_ULYSSES_FAST", "1") != "0"     # the cp_ulysses layer with its layout copies folded away (_ulysses)

FP32_PARAMS = ("self_attention.q_layernorm.", "self_attention.k_layernorm.", "self_attn_post_norm.", "mlp_post_norm.",
               "final_layernorm.")


def _cfg(model_config, name, default=None):
    return getattr(model_config, name, default)


class HipFullyParallelAttention:
    """The attention half of a layer; owns the per-layer `MagiKVCacheManager` like the reference's module."""

    def __init__(self, model_config, engine_config, layer_number: int, device):
        self.model_config, self.engine_config, self.layer_number = model_config, engine_config, layer_number
        self.device = torch.device(device)
        self.hd = model_config.kv_channels
        if self.hd != 128:
            raise NotImplementedError("the HIP attention kernels are built for head_dim 128 (MAGI-4.5B / 24B)")
        self.hq = model_config.num_attention_heads
        self.hk = model_config.num_query_groups
        cp = cpl.get_cp_world_size() if getattr(engine_config, "cp_strategy", "none") != "none" else 1
        # dit_module.py:796-800: kv heads per rank (replicated when there are fewer kv heads than ranks)
        self.hk_local = 1 if (cp > self.hk and cp % self.hk == 0) else self.hk // cp
        self.hq_local = self.hq // cp
        self.kv_cache_manager = MagiKVCacheManager(layer_number=layer_number, num_query_groups_per_partition=self.hk_local,
                                                   hidden_size_per_attention_head=self.hd, engine_config=engine_config)
        self.adapt_linear_quant = bool(getattr(engine_config, "fp8_quant", False)) and layer_number != 0 and \
            layer_number != model_config.num_layers - 1
        self.w: Dict[str, torch.Tensor] = {}
        self.fp8: Dict[str, StaticFp8Linear] = {}      # fp8_quant layers: q / qx / k / v / proj on the static-scale FP8 linears

    def load(self, W: Dict[str, torch.Tensor], prefix: str) -> None:
        dev, Q = self.device, self.hq * self.hd
        g = lambda k: W[prefix + k].to(dev)
        quant = prefix + "linear_qkv.q.weight_scale" in W
        if quant != self.adapt_linear_quant:
            raise ValueError(f"layer {self.layer_number}: engine_config.fp8_quant says {'FP8' if self.adapt_linear_quant else 'bf16'} "
                             f"linears, the checkpoint holds {'FP8' if quant else 'bf16'} ones")
        if quant:                                    # PerTensorQuantizedFp8Linear x 4 (dit_module.py:408-413): own divisor each
            for nm in ("q", "qx", "k", "v"):
                k = f"linear_qkv.{nm}."
                self.fp8[nm] = StaticFp8Linear(g(k + "weight"), g(k + "weight_scale"), g(k + "input_scale"), g(k + "input_scale"))
            # the four linears quantise the same LayerNorm row, each with its own input_scale vector: one pass produces one e4m3 copy per
            # DISTINCT vector (a checkpoint calibrated on the shared input may well carry the same vector four times: then one copy)
            h = self.fp8["q"].in_features
            uniq, self.qkv_slot = [], {}
            for nm in ("q", "qx", "k", "v"):
                d = self.fp8[nm].divisor.expand(h)
                hit = next((i for i, u in enumerate(uniq) if torch.equal(u, d)), None)
                if hit is None:
                    uniq.append(d)
                    hit = len(uniq) - 1
                self.qkv_slot[nm] = hit
            self.qkv_divisors = torch.stack(uniq).contiguous()
        else:
            self.w["qkv"] = torch.cat([g("linear_qkv.q.weight"), g("linear_qkv.qx.weight"), g("linear_qkv.k.weight"),
                                       g("linear_qkv.v.weight")], dim=0).to(BF16).contiguous()
        self.w["ln_w"], self.w["ln_b"] = g("linear_qkv.layer_norm.weight").to(BF16), g("linear_qkv.layer_norm.bias").to(BF16)
        self.w["kvx"] = g("linear_kv_xattn.weight").to(BF16).contiguous()
        # attn_linear_proj (:1287): the module multiplies linear_proj.weight with rearrange(cat([core, xattn]),
        # "(n hn hd) -> (hn n hd)", n=2, hn=8).  Column j = n*Q + g*c + r of the un-rearranged concatenation (c = Q/8) is
        # column g*2c + n*c + r of the rearranged one: permute the weight's columns once instead of the activations every call.
        c = Q // 8
        j = torch.arange(2 * Q)
        n, rem = j // Q, j % Q
        src = (rem // c) * (2 * c) + n * c + rem % c
        if quant:                                    # PerChannelQuantizedFp8Linear (:867): bytes and smooth_scale permuted alike
            pw = g("linear_proj.weight").view(torch.uint8).reshape(-1, 2 * Q)[:, src.to(dev)].contiguous()
            self.fp8["proj"] = StaticFp8Linear(pw, g("linear_proj.weight_scale"), g("linear_proj.input_scale"),
                                               g("linear_proj.smooth_scale").reshape(-1)[src.to(dev)])
        else:
            self.w["proj"] = g("linear_proj.weight").to(BF16)[:, src.to(dev)].contiguous()
        for nm in ("q_layernorm", "k_layernorm"):
            self.w[nm] = (g(nm + ".weight").float().contiguous(), g(nm + ".bias").float().contiguous())
        for nm in ("q_layernorm_xattn", "k_layernorm_xattn"):
            self.w[nm] = (g(nm + ".weight").to(BF16).contiguous(), g(nm + ".bias").to(BF16).contiguous())

    # -----------------------------------------------------------------------------------------------------------------
    def forward(self, hidden_states: torch.Tensor, key_value_states: torch.Tensor, inference_params: Optional[InferenceParams],
                rotary_pos_emb: torch.Tensor, meta_args: ModelMetaArgs, attn_cat: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> `[sq, 2Q]` bf16 = (core_attn_out | xattn_out) per token, the input of `linear_proj` (weights pre-permuted)."""
        mc, w = self.model_config, self.w
        s_len, bsz, h = hidden_states.shape
        if bsz != 1:
            raise NotImplementedError("MAGI runs the cached path with batch 1 (3-cfg converts ranges to batch upstream)")
        eps, one_p = mc.layernorm_epsilon, bool(mc.apply_layernorm_1p)
        Q = self.hq * self.hd
        x2 = hidden_states.view(s_len, h)
        if self.fp8:                                                                      # [s, q | qx | k | v]
            KV = self.hk * self.hd
            mixed = torch.empty(s_len, 2 * Q + 2 * KV, dtype=BF16, device=x2.device)
            # one pass: LayerNorm + the four linears' own static quantisers of its bf16 row (each has its own input_scale vector)
            hq = ops.layernorm_quant_static(x2, eps, self.qkv_divisors, gamma=w["ln_w"], beta=w["ln_b"])
            off = 0
            for nm, n in (("q", Q), ("qx", Q), ("k", KV), ("v", KV)):
                self.fp8[nm].matmul(hq[:, self.qkv_slot[nm]], out=mixed[:, off:off + n])
                off += n
        else:
            hln = ops.layernorm(x2, eps, gamma=w["ln_w"], beta=w["ln_b"])
            mixed = ops.linear(hln, w["qkv"], None)
        if attn_cat is None:
            attn_cat = torch.empty(s_len, 2 * Q, dtype=BF16, device=x2.device)
        q_buf = torch.empty(s_len, Q, dtype=BF16, device=x2.device)
        qx_buf = torch.empty(s_len, Q, dtype=BF16, device=x2.device)
        rope = rotary_pos_emb if rotary_pos_emb.dtype == torch.float32 else rotary_pos_emb.float()
        strategy = getattr(self.engine_config, "cp_strategy", "none")
        cp = cpl.get_cp_world_size() if strategy != "none" else 1
        cq = meta_args.core_attn_params.np_q_range
        ck = meta_args.core_attn_params.np_k_range
        if cp == 1:
            handle, (row0, split, row1) = self.kv_cache_manager.prepare_append(s_len, self.hk, self.hd, BF16, x2.device,
                                                                               inference_params, meta_args)
            ops.magi_head_prep(mixed, layout=0, q_heads=self.hq, kv_heads=self.hk, eps=eps, layernorm_1p=one_p,
                               k_out=handle.view.k, v_out=handle.view.v, kv_head_stride=self.hd, ld_kv=self.hk * self.hd,
                               row0=row0, split=split, row1=row1, rope=rope, qn=w["q_layernorm"], kn=w["k_layernorm"],
                               xn=w["q_layernorm_xattn"], q_out=q_buf, qx_out=qx_buf, q_scale=Q_SCALE)
            _range_attention(q_buf, handle, cq, ck, meta_args.denoising_range_num, attn_cat, self.hq)
        elif strategy == "cp_shuffle_overlap":
            self._cso(mixed, rope, q_buf, qx_buf, attn_cat, inference_params, meta_args, cp, eps, one_p)
        else:
            self._ulysses(mixed, rope, q_buf, qx_buf, attn_cat, inference_params, meta_args, cp, eps, one_p)
        # ---- cross-attention on the caption tokens (:954-970, :1047-1085); the local tokens attend the whole caption
        xp = meta_args.cross_attn_params
        kvx = ops.linear(key_value_states, w["kvx"], None)                                # [y, hk x (k | v)]
        yt = kvx.shape[0]
        kx = torch.empty(yt, self.hk, self.hd, dtype=BF16, device=x2.device)
        vx = torch.empty(yt, self.hk, self.hd, dtype=BF16, device=x2.device)
        ops.magi_head_prep(kvx, layout=1, q_heads=0, kv_heads=self.hk, eps=eps, layernorm_1p=one_p, k_out=kx, v_out=vx,
                           kv_head_stride=self.hd, ld_kv=self.hk * self.hd, xn=w["k_layernorm_xattn"])
        xview = ops.KvCacheView(kx, vx)
        for (qs, qe), (ks, ke) in _cross_segments(xp):
            ops.attention_ld(qx_buf[qs:qe], xview, ke, attn_cat[qs:qe, Q:], self.hq, kv_start=ks, tag="attn_magi_x")
        return attn_cat

    def _ulysses(self, mixed, rope, q_buf, qx_buf, attn_cat, inference_params, meta_args, cp, eps, one_p):
        """cp_ulysses (:1126-1158): local tokens x all heads -> all tokens x this rank's heads, attention, and back.  K and V of
        the local tokens are produced straight into the (K | V)-fused staging layout `all_to_all_input_split` sends."""
        w = self.w
        s_len = mixed.shape[0]
        sizes = [int(v) for v in meta_args.cp_split_sizes]
        cq = meta_args.core_attn_params.np_q_range
        ck = meta_args.core_attn_params.np_k_range
        overlap = getattr(self.engine_config, "ulysses_overlap_degree", 1)
        if _ULYSSES_FAST and overlap == 1 and self.hk == cp and self.hq % cp == 0:
            # Round 5: the same collectives with the layout copies folded into their neighbours.  The head -> rank all-to-alls send
            # [(cp seq), heads / cp, hd] (context_parallel.py `_heads_to_ranks`: a permute + contiguous of q and of k|v per layer); the
            # head-prep kernel writes both operands in that order directly (one kv head per rank: a head stride; q: ifx_magi_head_prep's
            # q_group), and the way back — all-to-all pieces [cp, seq, hn hd] -> rows [seq, (cp hn hd)] of the projection input — is ONE
            # strided copy instead of cat + permute + contiguous + copy.  Four ~12 us copies per layer and forward less; the bytes on
            # the wire, the cache rule and the attention calls are those of the scheduler below (tests: the 2-rank goldens).
            hpr = self.hq // cp
            kv_send = torch.empty(cp, s_len, 1, 2 * self.hd, dtype=BF16, device=mixed.device)
            q_send = torch.empty(cp, s_len, hpr * self.hd, dtype=BF16, device=mixed.device)
            ops.magi_head_prep(mixed, layout=0, q_heads=self.hq, kv_heads=self.hk, eps=eps, layernorm_1p=one_p,
                               k_out=kv_send, v_out=kv_send.view(-1)[self.hd:], kv_head_stride=s_len * 2 * self.hd, ld_kv=2 * self.hd,
                               rope=rope, qn=w["q_layernorm"], kn=w["k_layernorm"], xn=w["q_layernorm_xattn"], q_out=q_send,
                               qx_out=qx_buf, q_scale=Q_SCALE, q_group=hpr)
            total = sum(sizes)
            kv_all = torch.empty(total, 1, 2 * self.hd, dtype=BF16, device=mixed.device)
            q_all = torch.empty(total, hpr, self.hd, dtype=BF16, device=mixed.device)
            hkv = cpl._a2a(kv_all, kv_send.view(cp * s_len, 1, 2 * self.hd), out_sizes=sizes)
            hq_ = cpl._a2a(q_all, q_send.view(cp * s_len, hpr, self.hd), out_sizes=sizes)
            hkv.wait()
            key = self.kv_cache_manager.adjust_key_and_value_for_inference(kv_all, inference_params, meta_args)
            hq_.wait()
            out = torch.empty_like(q_all)
            _range_attention(q_all.view(total, -1), key, cq, ck, meta_args.denoising_range_num, out.view(total, -1), hpr)
            back, hb = cpl.all_to_all_output_split(out, sizes)                     # [(cp seq), hpr, hd], source-rank major
            hb.wait()
            Q = self.hq * self.hd
            attn_cat[:, :Q].view(s_len, cp, hpr * self.hd).copy_(back.view(cp, s_len, hpr * self.hd).transpose(0, 1))
            return
        kv_stage = torch.empty(s_len, self.hk, 2 * self.hd, dtype=BF16, device=mixed.device)
        ops.magi_head_prep(mixed, layout=0, q_heads=self.hq, kv_heads=self.hk, eps=eps, layernorm_1p=one_p,
                           k_out=kv_stage, v_out=kv_stage.view(-1)[self.hd:], kv_head_stride=2 * self.hd,
                           ld_kv=self.hk * 2 * self.hd, rope=rope, qn=w["q_layernorm"], kn=w["k_layernorm"],
                           xn=w["q_layernorm_xattn"], q_out=q_buf, qx_out=qx_buf, q_scale=Q_SCALE)

        def core(qc, key, value):
            out = torch.empty_like(qc)
            _range_attention(qc.view(qc.shape[0], -1), key, cq, ck, meta_args.denoising_range_num, out.view(qc.shape[0], -1), qc.shape[1])
            return out
        Q = self.hq * self.hd
        core_out, _ = cpl.UlyssesScheduler.get_attn_and_xattn_with_fused_kv_comm(
            lambda: q_buf.view(s_len, self.hq, self.hd), lambda: kv_stage,
            lambda kv: (self.kv_cache_manager.adjust_key_and_value_for_inference(kv, inference_params, meta_args),) * 2,
            core, lambda: None, getattr(self.engine_config, "ulysses_overlap_degree", 1), 1, cp, sizes)
        attn_cat[:, :Q].copy_(core_out.view(s_len, Q))


    def _cso(self, mixed, rope, q_buf, qx_buf, attn_cat, inference_params, meta_args, cp, eps, one_p):
        """cp_shuffle_overlap (:1156-1188): this rank holds a slice of every denoising chunk.  K/V of all chunks go to the head
        owners in one message; the queries travel chunk by chunk, each message overlapping the attention of the chunk before
        (`CSOHelper.overlap`), whose output rides back with the next queries."""
        w = self.w
        s_len = mixed.shape[0]
        sizes = [int(v) for v in meta_args.cp_split_sizes]
        dn = int(meta_args.denoising_range_num)
        kv_stage = torch.empty(s_len, self.hk, 2 * self.hd, dtype=BF16, device=mixed.device)
        ops.magi_head_prep(mixed, layout=0, q_heads=self.hq, kv_heads=self.hk, eps=eps, layernorm_1p=one_p,
                           k_out=kv_stage, v_out=kv_stage.view(-1)[self.hd:], kv_head_stride=2 * self.hd,
                           ld_kv=self.hk * 2 * self.hd, rope=rope, qn=w["q_layernorm"], kn=w["k_layernorm"],
                           xn=w["q_layernorm_xattn"], q_out=q_buf, qx_out=qx_buf, q_scale=Q_SCALE)
        kv, handle_kv = cpl.cso_communication(kv_stage, cp, sizes, "kv")               # [(cp dn m), hk_local, 2 hd]
        helper = cpl.CSOHelper(dn, cp, sizes)
        qs, handle_q = helper.split_query_for_overlap(q_buf.view(s_len, self.hq, self.hd))
        handle_kv.wait()
        m = s_len // dn
        # (cp dn m) -> dn (cp m), every chunk's padding rows dropped: the cache rule sees the chunks as a single device does
        kv = kv.view(cp, dn, m, kv.shape[1], kv.shape[2]).transpose(0, 1).reshape(dn, cp * m, kv.shape[1], kv.shape[2])
        kv = kv[:, :meta_args.clip_token_nums].flatten(0, 1).contiguous()
        handle = self.kv_cache_manager.adjust_key_and_value_for_inference(kv, inference_params, meta_args)
        handle_q.wait()
        ck = meta_args.core_attn_params.np_k_range

        def fattn(q, key, value, i):            # full_attention (:1017-1045): every (padded) query row of chunk i over its key range
            q = q.contiguous()
            out = torch.empty_like(q)
            rows, heads = q.shape[0], q.shape[1]
            ks, ke = int(ck[i, 0]), int(ck[i, 1])
            if ke > handle.kv_len:
                raise ValueError(f"k_range[{i}] = [{ks}, {ke}) exceeds the {handle.kv_len} available keys")
            ops.attention_ld(q.view(rows, -1), handle.view, ke, out.view(rows, -1), heads, kv_start=ks, scale=ATTN_SCALE, tag="attn_magi")
            return out
        outs, handle_attn = helper.overlap(fattn, qs, handle, handle)
        handle_attn.wait()
        Q = self.hq * self.hd
        cat = torch.concat(outs, dim=0)                                                # (dn cp m) hn hd -> (dn m) (cp hn hd)
        core = cat.view(dn, cp, m, cat.shape[1], self.hd).permute(0, 2, 1, 3, 4).reshape(s_len, Q)
        attn_cat[:, :Q].copy_(core)


def _range_attention(q2d: torch.Tensor, handle, q_range, k_range, n: int, out2d: torch.Tensor, heads: int) -> None:
    """core_attention (:972-1015): per denoising range, queries attend their key range of the in-place cache; launches that would
    leave the chip half empty (a rank's 3 query heads) split their key range (ifx_attn_split_plan)."""
    qr = [(int(q_range[i, 0]), int(q_range[i, 1])) for i in range(n)]
    kr = [(int(k_range[i, 0]), int(k_range[i, 1])) for i in range(n)]
    for i, (ks, ke) in enumerate(kr):
        if ke > handle.kv_len:
            raise ValueError(f"k_range[{i}] = [{ks}, {ke}) exceeds the {handle.kv_len} available keys")
    # (All ranges in ONE launch — ops.attention_ranges / ifx_attn_fwd_ranges — was measured on the rank shape of cp = 8, 4 x 12150
    #  queries x 3 heads over 2 .. 5 chunks of keys: 727 TFLOP/s against 780 for the four split-KV launches below.  Tiles of
    #  different ranges stream different key windows out of phase, so the L2 sharing between the query tiles of one launch is lost.
    #  The entry point stays for callers with many short ranges, where launch count is what costs.)
    for (qs, qe), (ks, ke) in zip(qr, kr):
        ops.attention_ld(q2d[qs:qe], handle.view, ke, out2d[qs:qe], heads, kv_start=ks, scale=ATTN_SCALE, tag="attn_magi")


def _cross_segments(xp):
    """(query range, key range) pairs of the packed cross-attention: the per-rank `q_ranges` / `kv_ranges` when context
    parallelism has clipped them (context_parallel.py:135-216), otherwise consecutive `cu_seqlens` segments."""
    if xp.q_ranges is not None:
        return list(zip(xp.q_ranges.tolist(), xp.kv_ranges.tolist()))
    cq, ck = xp.cu_seqlens_q.tolist(), xp.cu_seqlens_kv.tolist()
    return [((cq[i], cq[i + 1]), (ck[i], ck[i + 1])) for i in range(len(cq) - 1)]


class HipMagiTransformerLayer:
    def __init__(self, model_config, engine_config, layer_number: int = 1, device="cuda"):
        self.model_config, self.engine_config, self.layer_number = model_config, engine_config, layer_number
        self.device = torch.device(device)
        if _cfg(model_config, "gated_linear_unit", False):
            raise NotImplementedError("gated_linear_unit (flashinfer silu_and_mul) is not used by MAGI-4.5B and not built")
        self.self_attention = HipFullyParallelAttention(model_config, engine_config, layer_number, device)
        self.w: Dict[str, torch.Tensor] = {}
        self.fp8: Dict[str, StaticFp8Linear] = {}

    def load_state_dict(self, W: Dict[str, torch.Tensor], prefix: str = "") -> None:
        dev = self.device
        g = lambda k: W[prefix + k].to(dev)
        self.self_attention.load(W, prefix + "self_attention.")
        self.w["ada_w"], self.w["ada_b"] = g("ada_modulate_layer.proj.0.weight").to(BF16).contiguous(), g("ada_modulate_layer.proj.0.bias").to(BF16)
        for nm in ("self_attn_post_norm", "mlp_post_norm"):
            self.w[nm] = (g(nm + ".weight").float().contiguous(), g(nm + ".bias").float().contiguous())
        self.w["mlp_ln"] = (g("mlp.layer_norm.weight").to(BF16), g("mlp.layer_norm.bias").to(BF16))
        if prefix + "mlp.linear_fc1.weight_scale" in W:          # fp8_quant: fc1 per-tensor form, fc2 per-channel form (:526, :539)
            self.fp8["fc1"] = StaticFp8Linear(g("mlp.linear_fc1.weight"), g("mlp.linear_fc1.weight_scale"),
                                              g("mlp.linear_fc1.input_scale"), g("mlp.linear_fc1.input_scale"))
            self.fp8["fc2"] = StaticFp8Linear(g("mlp.linear_fc2.weight"), g("mlp.linear_fc2.weight_scale"),
                                              g("mlp.linear_fc2.input_scale"), g("mlp.linear_fc2.smooth_scale"))
        else:
            self.w["fc1"], self.w["fc2"] = g("mlp.linear_fc1.weight").to(BF16).contiguous(), g("mlp.linear_fc2.weight").to(BF16).contiguous()

    def gate(self, condition: torch.Tensor) -> torch.Tensor:
        """softcap(AdaModulateLayer(condition)) `[b * ranges, 2h]` (:196-198, :1300-1303)."""
        c2 = condition.reshape(-1, condition.shape[-1]).contiguous()
        g = ops.linear(ops.act_rows(c2, _hip.IFX_ACT_SILU), self.w["ada_w"], self.w["ada_b"])
        return ops.act_rows(g, _hip.IFX_ACT_TANH)

    def forward(self, hidden_states: torch.Tensor, condition: torch.Tensor, condition_map: torch.Tensor,
                y_xattn_flat: torch.Tensor, rotary_pos_emb: torch.Tensor, inference_params: Optional[InferenceParams],
                meta_args: ModelMetaArgs) -> torch.Tensor:
        mc, w = self.model_config, self.w
        s_len, bsz, h = hidden_states.shape
        eps, one_p = mc.layernorm_epsilon, bool(mc.apply_layernorm_1p)
        x2 = hidden_states.reshape(s_len * bsz, h)
        cmap = condition_map.reshape(-1)
        if cmap.dtype != torch.int32:
            cmap = cmap.to(torch.int32)
        attn_cat = self.self_attention.forward(hidden_states, y_xattn_flat, inference_params, rotary_pos_emb, meta_args)
        sa = self.self_attention
        proj = sa.fp8["proj"](attn_cat) if sa.fp8 else ops.linear(attn_cat, sa.w["proj"], None)
        gate = self.gate(condition)
        hs = ops.magi_gate_norm_residual(proj, x2, cmap, gate[:, :h], *w["self_attn_post_norm"], eps, one_p)
        if self.fp8:      # LayerNorm -> fc1's quantiser in one pass; fc1's GELU epilogue writes fc2's quantised input
            fc1, fc2 = self.fp8["fc1"], self.fp8["fc2"]
            mq = ops.layernorm_quant_static(hs, eps, fc1.divisor.expand(h).view(1, -1).contiguous(), gamma=w["mlp_ln"][0], beta=w["mlp_ln"][1])
            m = fc2.matmul(fc1.matmul_quant_out(mq[:, 0], fc2, _hip.IFX_EPI_GELU_ERF))
        else:
            m = ops.layernorm(hs, eps, gamma=w["mlp_ln"][0], beta=w["mlp_ln"][1])
            m = ops.linear(m, w["fc1"], None, epilogue=_hip.IFX_EPI_GELU_ERF)
            m = ops.linear(m, w["fc2"], None)
        out = ops.magi_gate_norm_residual(m, hs, cmap, gate[:, h:], *w["mlp_post_norm"], eps, one_p)
        return out.view(s_len, bsz, h)

    __call__ = forward


class HipMagiTransformerBlock:
    """`TransformerBlock` (:1322-1390): the layer stack of one pipeline stage.  The final LayerNorm runs on `.float()` hidden
    states upstream (fp32 module, dit_model.py:635-636) and feeds the fp32 `final_linear`; it is applied by the caller."""

    def __init__(self, model_config, engine_config, device="cuda", num_layers: Optional[int] = None):
        n = model_config.num_layers if num_layers is None else num_layers
        self.layers: List[HipMagiTransformerLayer] = [HipMagiTransformerLayer(model_config, engine_config, i, device) for i in range(n)]

    def load_state_dict(self, W: Dict[str, torch.Tensor], prefix: str = "layers.") -> None:
        for i, layer in enumerate(self.layers):
            layer.load_state_dict(W, f"{prefix}{i}.")

    def forward(self, hidden_states, condition, condition_map, y_xattn_flat, rotary_pos_emb, inference_params, meta_args):
        for layer in self.layers:
            hidden_states = layer(hidden_states, condition, condition_map, y_xattn_flat, rotary_pos_emb, inference_params, meta_args)
        return hidden_states

    __call__ = forward


def synthetic_layer_state_dict(model_config, seed: int = 0, device="cuda", fp8: bool = False) -> Dict[str, torch.Tensor]:
    """Random weights of ONE layer with the reference's state-dict keys and dtypes (bf16 parameters, fp32 q/k layer norms and post
    norms — dit_model.py:620-637), generated on the device: benchmarks and smoke runs (no checkpoint exists offline).  Matrices
    ~ N(0, 1/fan_in) so that activations keep unit scale through the stack."""
    mc = model_config
    g = torch.Generator(device=device).manual_seed(seed)
    h, hd = mc.hidden_size, mc.kv_channels
    q, kv, f = hd * mc.num_attention_heads, hd * mc.num_query_groups, mc.ffn_hidden_size
    cond = int(h * getattr(mc, "cond_hidden_ratio", 0.25))
    xat = int(h * getattr(mc, "xattn_cond_hidden_ratio", 1.0))
    mat = lambda o, i: (torch.randn(o, i, generator=g, device=device) * i ** -0.5).to(BF16)
    vec = lambda n, dt, mean=0.0: (mean + 0.1 * torch.randn(n, generator=g, device=device)).to(dt)
    sd = {"ada_modulate_layer.proj.0.weight": mat(2 * h, cond), "ada_modulate_layer.proj.0.bias": vec(2 * h, BF16),
          "self_attention.linear_qkv.layer_norm.weight": vec(h, BF16, 1.0), "self_attention.linear_qkv.layer_norm.bias": vec(h, BF16),
          "self_attention.linear_qkv.q.weight": mat(q, h), "self_attention.linear_qkv.qx.weight": mat(q, h),
          "self_attention.linear_qkv.k.weight": mat(kv, h), "self_attention.linear_qkv.v.weight": mat(kv, h),
          "self_attention.linear_kv_xattn.weight": mat(2 * kv, xat), "self_attention.linear_proj.weight": mat(h, 2 * q),
          "mlp.layer_norm.weight": vec(h, BF16, 1.0), "mlp.layer_norm.bias": vec(h, BF16),
          "mlp.linear_fc1.weight": mat(f, h), "mlp.linear_fc2.weight": mat(h, f)}
    for nm, n, dt in (("self_attention.q_layernorm", hd, torch.float32), ("self_attention.k_layernorm", hd, torch.float32),
                      ("self_attention.q_layernorm_xattn", hd, BF16), ("self_attention.k_layernorm_xattn", hd, BF16),
                      ("self_attn_post_norm", h, torch.float32), ("mlp_post_norm", h, torch.float32)):
        sd[nm + ".weight"], sd[nm + ".bias"] = vec(n, dt), vec(n, dt)
    if fp8:       # an inner layer of an fp8_quant checkpoint: e4m3 weights [1, out, in] + static scales (dit_module.py:434-490)
        for nm, per_channel in (("self_attention.linear_qkv.q", False), ("self_attention.linear_qkv.qx", False),
                                ("self_attention.linear_qkv.k", False), ("self_attention.linear_qkv.v", False),
                                ("mlp.linear_fc1", False), ("self_attention.linear_proj", True), ("mlp.linear_fc2", True)):
            wf = sd[nm + ".weight"].float()
            ws = (wf.abs().max() / 448.0).reshape(1)
            sd[nm + ".weight"] = (wf / ws).to(torch.float8_e4m3fn).unsqueeze(0)
            sd[nm + ".weight_scale"] = ws
            spread = 0.03 * (1.0 + 0.2 * (2 * torch.rand(wf.shape[1], generator=g, device=device) - 1))
            if per_channel:
                sd[nm + ".input_scale"] = torch.full((1,), 0.03, device=device)
                sd[nm + ".smooth_scale"] = spread.unsqueeze(0)
            else:
                sd[nm + ".input_scale"] = spread
    return sd
