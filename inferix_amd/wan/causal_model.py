"""HipCausalWanModel — the causal Wan DiT denoise-step forward on MI355X.

Drop-in for the inference path of the reference's `CausalWanModel`
(inferix/models/self_forcing/causal_model.py:518-1026): same constructor arguments,
same `state_dict` key names, same `forward(x, t, context, seq_len, kv_cache_meta=...,
crossattn_cache_meta=..., current_start=..., kv_cache_manager=..., kv_cache_requests=...)`
call and the same mutation contract on the KV cache and on `kv_cache_meta`
(`global_end_index` / `local_end_index`, causal_model.py:277-329).

Host code only orchestrates: every per-token op of the 30-layer stack is one of the
hand-written gfx950 kernels behind the C-ABI (`inferix_amd.hip_ops`):

    AdaLN-LN -> fused QKV GEMM -> fused RMSNorm+RoPE+KV-append -> paged flash attention
    -> O GEMM (+gate+residual) -> affine LN -> q GEMM -> RMSNorm -> cross attention
    -> O GEMM (+residual) -> AdaLN-LN -> FFN GEMM (+GELU) -> FFN GEMM (+gate+residual)

13 launches per layer, the activation is updated in place, the KV cache is read in
place through the manager's tensors (no per-layer fetch/stack/copy-back as in
causal_model.py:416-441).  PyTorch is used for allocation, streams and the O(frames)
glue: timestep embedding MLP (M = frames), modulation table add, patchify /
unpatchify index permutes.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .. import _hip, hip_ops as ops
from ..kvcache_manager import KVCacheManager, KVCacheRequest
from ..kvcache_manager.model import SelfForcingKVCacheManagerFactory
from . import components as C

BF16 = torch.bfloat16


class ParallelConfig:
    """Mirror of inferix/models/wan_base/utils/parallel_config.py:3-30.  The reference raises when no
    flash-attn wheel is importable; here the (only) backend is the in-tree HIP kernel."""

    def __init__(self, ulysses_size=1, ring_size=1, local_rank=0, rank=0, world_size=1,
                 ring_strategy="pass-kv", attn_backend=None):
        from ..attention import collect_supported_attn
        ulysses_size, ring_size, world_size = int(ulysses_size), int(ring_size), int(world_size)
        if ulysses_size < 1 or ring_size < 1:
            raise ValueError(f"ulysses_size={ulysses_size} / ring_size={ring_size}: degrees must be >= 1")
        if ulysses_size * ring_size not in (1, world_size):
            # The reference's launcher passes degrees whose product is the world size (example/self_forcing/self_forcing.sh:12-13,
            # run_self_forcing.py:58-67) and shards its cache tokens / ring x heads / ulysses (self_forcing_kv_cache_manager.py:45-57)
            # for CoreAttention's all-to-all + ring (attention/distributed.py:53-281).  Here BOTH degrees select the same thing:
            # the sequence-parallel exchange of inferix_amd/sequence_parallel.py at degree world_size (hw slice per frame, K/V
            # exchange, replicated cache — the same function of the inputs; 12 heads do not divide by 8).  A product that is not
            # the world size names a process-group layout no launcher of the reference produces.
            raise ValueError(f"ulysses_size={ulysses_size} x ring_size={ring_size} != world_size={world_size}: the sequence-parallel "
                             "degree is ulysses_size * ring_size and must equal the number of ranks")
        self.ulysses_size, self.ring_size = ulysses_size, ring_size
        self.local_rank, self.rank, self.world_size = local_rank, rank, world_size
        self.ring_strategy = ring_strategy
        supported = collect_supported_attn()
        if attn_backend is None:
            attn_backend = "HipPagedFA"
        if attn_backend not in supported:
            raise ValueError(f"Specified attention backend '{attn_backend}' is not available. "
                             f"Available backends: {list(supported)}")
        self.attn_backend = attn_backend


@dataclass
class KVIndexStep:
    local_start: int
    local_end: int
    global_end: int
    evicted: int
    rolled: int
    sink_tokens: int


def kv_index_update(global_end: int, local_end: int, current_start: int, num_new: int, cache_size: int,
                    local_attn_size: int, sink_tokens: int) -> KVIndexStep:
    """Integer slot arithmetic of one cache append (causal_model.py:277-300,328-329), host side, no
    device sync.  Rolling branch iff a NEW block (current_end > global_end) does not fit."""
    current_end = current_start + num_new
    evicted = rolled = 0
    if local_attn_size != -1 and current_end > global_end and num_new + local_end > cache_size:
        evicted = num_new + local_end - cache_size
        rolled = local_end - evicted - sink_tokens
    new_local_end = local_end + current_end - global_end - evicted
    return KVIndexStep(new_local_end - num_new, new_local_end, current_end, evicted, rolled, sink_tokens)


class _Block:
    """Per-layer handle: weights + the reference-named `kv_cache_manager` adapter attribute that the
    pipeline uses (`generator.model.blocks[i].kv_cache_manager`, CausalInferencePipeline.py:463,485,500)."""

    def __init__(self, layer_idx: int, num_heads: int, head_dim: int, enable_kv_offload: bool):
        self.layer_idx = layer_idx
        self.kv_cache_manager = SelfForcingKVCacheManagerFactory.create_manager(
            layer_idx, num_heads, head_dim, enable_kv_offload=enable_kv_offload)
        self.w: Dict[str, torch.Tensor] = {}
        self.cross_meta = {"is_init": False}       # CausVid keeps this flag on the block (causal_model.py:227)

    @property
    def is_cross_attn_init(self) -> bool:
        return self.cross_meta["is_init"]

    @is_cross_attn_init.setter
    def is_cross_attn_init(self, v: bool) -> None:
        self.cross_meta["is_init"] = bool(v)


class HipCausalWanModel(torch.nn.Module):
    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, local_attn_size=-1,
                 sink_size=0, qk_norm=True, cross_attn_norm=True, eps=1e-6, enable_kv_offload=False,
                 parallel_config: Optional[ParallelConfig] = None, device="cuda"):
        super().__init__()
        if model_type != "t2v":
            raise NotImplementedError("HipCausalWanModel: only the t2v cross-attention path is built")
        if not (qk_norm and cross_attn_norm):
            raise NotImplementedError("HipCausalWanModel: qk_norm and cross_attn_norm are always on (Wan2.1 config)")
        if dim % num_heads or dim // num_heads != 128:
            raise NotImplementedError("HipCausalWanModel: kernels are built for head_dim 128")
        _hip.load()                                   # fail loudly if the HIP library is missing
        self.model_type, self.patch_size, self.text_len = model_type, tuple(patch_size), text_len
        self.in_dim, self.dim, self.ffn_dim, self.freq_dim = in_dim, dim, ffn_dim, freq_dim
        self.text_dim, self.out_dim, self.num_heads, self.num_layers = text_dim, out_dim, num_heads, num_layers
        self.head_dim = dim // num_heads
        self.local_attn_size, self.sink_size, self.eps = local_attn_size, sink_size, eps
        self.qk_norm, self.cross_attn_norm = qk_norm, cross_attn_norm
        self.enable_kv_offload = enable_kv_offload
        self.parallel_config = parallel_config if parallel_config is not None else ParallelConfig()
        pc = self.parallel_config
        u, r = int(getattr(pc, "ulysses_size", 1)), int(getattr(pc, "ring_size", 1))
        if u * r not in (1, int(getattr(pc, "world_size", 1))):   # also catches duck-typed configs (SimpleNamespace)
            raise ValueError(f"CausalWanModel: ulysses_size={u} x ring_size={r} must equal world_size={getattr(pc, 'world_size', 1)} "
                             "(both degrees map onto the one sequence-parallel exchange; see ParallelConfig)")
        self.num_frame_per_block = 1
        self.independent_first_frame = False
        self.device_ = torch.device(device)
        self.blocks: List[_Block] = [_Block(i, num_heads, self.head_dim, enable_kv_offload) for i in range(num_layers)]
        self.g: Dict[str, torch.Tensor] = {}         # non-block weights
        self.mod_all: Optional[torch.Tensor] = None  # [L, 1, 6, dim]
        self.freqs = C.rope_table(self.head_dim).to(self.device_)     # [1024, 64, 2] fp64
        self._scratch: Dict[Tuple, torch.Tensor] = {}
        # cross-attention over zero-padded prompts: request id -> (keys kept, multiplicity of the last one); `cross_dedup = False`
        # attends all text_len keys as the reference does (same mathematics, different rounding of the padded keys' weight)
        self.cross_dedup = True
        self._cross_dedup: Dict[str, Tuple[int, int]] = {}
        self._roll_scratch: Optional[torch.Tensor] = None
        self.v_direct = os.environ.get("IFX_V_DIRECT", "1") != "0"   # the q|k|v projection stores V straight into the cache rows (_run_block)
        self._temb_cache: Dict[Tuple, Tuple] = {}      # timestep tensor identity -> (tensor, E, eh): see _prologue
        self.index_trace: Optional[list] = None       # tests set a list: layer 0's KV index state after every forward
        self._chain = 0                               # 0 except while forward_pair enqueues its second forward
        self._pair_stream: Optional[torch.cuda.Stream] = None
        self.cp = None                                # set by inferix_amd.sequence_parallel when world_size > 1
        self.q_prescale = self.head_dim == 128        # exponent fast path of the self-attention kernel (hip_ops.attn_q_prescale)

    # ------------------------------------------------------------------ weights
    def parameters(self, recurse: bool = True):       # `next(generator.parameters())` is used by the pipeline
        for t in self.g.values():
            yield t
        for b in self.blocks:
            for t in b.w.values():
                if isinstance(t, torch.Tensor):
                    yield t

    def state_dict_keys(self) -> List[str]:
        keys = ["patch_embedding.weight", "patch_embedding.bias"]
        for n in ("text_embedding.0", "text_embedding.2", "time_embedding.0", "time_embedding.2", "time_projection.1"):
            keys += [n + ".weight", n + ".bias"]
        for i in range(self.num_layers):
            p = f"blocks.{i}."
            keys += [p + "modulation", p + "norm3.weight", p + "norm3.bias"]
            for a in ("self_attn", "cross_attn"):
                for n in ("q", "k", "v", "o"):
                    keys += [p + f"{a}.{n}.weight", p + f"{a}.{n}.bias"]
                keys += [p + f"{a}.norm_q.weight", p + f"{a}.norm_k.weight"]
            keys += [p + "ffn.0.weight", p + "ffn.0.bias", p + "ffn.2.weight", p + "ffn.2.bias"]
        keys += ["head.modulation", "head.head.weight", "head.head.bias"]
        return keys

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True, assign: bool = False):
        """Takes the reference's CausalWanModel state_dict (same key names) and packs it for the kernels:
        bf16 on the GPU, q|k|v weights fused to one [3*dim, dim] matrix."""
        missing = [k for k in self.state_dict_keys() if k not in sd]
        if missing and strict:
            raise KeyError(f"missing keys in state_dict: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        dev = self.device_

        def t(name):
            return sd[name].detach().to(device=dev, dtype=BF16).contiguous()

        g = self.g
        g["patch_w"] = t("patch_embedding.weight").flatten(1).contiguous()       # [dim, C*pt*ph*pw]
        g["patch_b"] = t("patch_embedding.bias")
        for short, n in (("text0", "text_embedding.0"), ("text2", "text_embedding.2"), ("time0", "time_embedding.0"),
                         ("time2", "time_embedding.2"), ("tproj", "time_projection.1"), ("head", "head.head")):
            g[short + "_w"], g[short + "_b"] = t(n + ".weight"), t(n + ".bias")
        g["head_mod"] = t("head.modulation")                                      # [1, 2, dim]
        mods = []
        for i, blk in enumerate(self.blocks):
            p = f"blocks.{i}."
            w = blk.w
            w["qkv_w"] = torch.cat([t(p + f"self_attn.{n}.weight") for n in "qkv"], dim=0).contiguous()
            w["qkv_b"] = torch.cat([t(p + f"self_attn.{n}.bias") for n in "qkv"], dim=0).contiguous()
            w["nq"], w["nk"] = t(p + "self_attn.norm_q.weight"), t(p + "self_attn.norm_k.weight")
            w["o_w"], w["o_b"] = t(p + "self_attn.o.weight"), t(p + "self_attn.o.bias")
            w["n3_w"], w["n3_b"] = t(p + "norm3.weight"), t(p + "norm3.bias")
            for n in "qkvo":
                w[f"c{n}_w"], w[f"c{n}_b"] = t(p + f"cross_attn.{n}.weight"), t(p + f"cross_attn.{n}.bias")
            w["cnq"], w["cnk"] = t(p + "cross_attn.norm_q.weight"), t(p + "cross_attn.norm_k.weight")
            w["f0_w"], w["f0_b"] = t(p + "ffn.0.weight"), t(p + "ffn.0.bias")
            w["f2_w"], w["f2_b"] = t(p + "ffn.2.weight"), t(p + "ffn.2.bias")
            mods.append(t(p + "modulation"))
        self.mod_all = torch.stack(mods, dim=0).contiguous()                      # [L, 1, 6, dim]
        self._temb_cache.clear()
        return self

    # ------------------------------------------------------------------ helpers
    def _buf(self, name: str, *shape) -> torch.Tensor:
        key = (self._chain, name, shape)              # `_chain`: which of two layer-interleaved forwards is being enqueued (forward_pair)
        b = self._scratch.get(key)
        if b is None:
            b = torch.empty(*shape, dtype=BF16, device=self.device_)
            self._scratch[key] = b
        return b

    @staticmethod
    def _meta_int(v) -> int:
        return int(v.item()) if isinstance(v, torch.Tensor) else int(v)

    @staticmethod
    def _meta_set(meta: dict, key: str, value: int) -> None:
        cur = meta.get(key)
        if isinstance(cur, torch.Tensor):
            cur.fill_(value)
        else:
            meta[key] = value

    def _kv_view(self, mgr: KVCacheManager, req: KVCacheRequest, name: str) -> ops.KvCacheView:
        t = mgr.get_raw(req, name)
        pt = mgr.page_table(req, name) if hasattr(mgr, "page_table") else None
        # the zero-copy K / V views of a layer are kept with the request's cache entry (they die with `free` / `free_layer`) and are
        # rebuilt when the storage or its page table changed
        caches = getattr(mgr, "request_to_kv_caches", {}).get(getattr(req, "request_id", None))
        store = getattr(caches, "views", None)
        key = (t.data_ptr(), None if pt is None else (pt.device.data_ptr(), pt.page_size))
        if store is not None:
            hit = store.get(name)
            if hit is not None and hit[0] == key:
                return hit[1]
        if not t.is_cuda:
            raise _hip.HipKernelError(
                f"KV cache '{name}' lives on {t.device}: the HIP path reads pages in place from HBM "
                "(create the model with enable_kv_offload=False)")
        if pt is not None:
            v = ops.KvCacheView.from_manager_tensor(t, pt.device, pt.page_size)
        else:
            v = ops.KvCacheView.from_manager_tensor(t)
        if store is not None:
            store[name] = (key, v)
        return v

    def _evict(self, mgr, req, name: str, view: ops.KvCacheView, step: KVIndexStep) -> None:
        """Sink + rolling eviction (causal_model.py:287-292): page-table rotation when the spans are page
        aligned, otherwise the physical shift kernel."""
        pt = mgr.page_table(req, name) if hasattr(mgr, "page_table") else None
        if pt is not None and step.sink_tokens % pt.page_size == 0 and step.evicted % pt.page_size == 0 \
                and step.rolled % pt.page_size == 0:
            mgr.rotate_pages(req, name, step.sink_tokens // pt.page_size, step.evicted // pt.page_size,
                             step.rolled // pt.page_size)
            return
        need = step.rolled * self.dim
        if self._chain:                               # the second of two interleaved forwards shifts on its own stream: its own scratch
            ops.kv_roll(view, step.sink_tokens, step.evicted, step.rolled, self._buf("roll", need))
            return
        if self._roll_scratch is None or self._roll_scratch.numel() < need:
            self._roll_scratch = torch.empty(need, dtype=BF16, device=self.device_)
        ops.kv_roll(view, step.sink_tokens, step.evicted, step.rolled, self._roll_scratch)

    def _q8_scratch(self, rows: int, cols: int, device):
        c = self._chain
        xq = self._scratch.get((c, "xq", rows, cols))
        if xq is None:
            xq = torch.empty(rows, cols, dtype=torch.uint8, device=device)
            self._scratch[(c, "xq", rows, cols)] = xq
            self._scratch[(c, "xs", rows)] = torch.empty(rows, dtype=torch.float32, device=device)
        return xq, self._scratch[(c, "xs", rows)]

    def _lin(self, w: Dict[str, torch.Tensor], key: str, x: torch.Tensor, **kw) -> torch.Tensor:
        """Linear `key` of a block: bf16 MFMA GEMM, or (after inferix_amd.quant.quantize_dynamic) per-token
        activation quantisation + fp8 / int8 MFMA GEMM with the same epilogue."""
        if key + "_q" in w:
            fmt = w[key + "_fmt"]
            xq, xs = self._q8_scratch(x.shape[0], x.shape[1], x.device)
            if w.get(key + "_act") == "per_tensor":
                ops.quant_per_tensor(x, fmt, q=xq, scale=xs)
            else:
                ops.quant_per_token(x, fmt, q=xq, scale=xs)
            return ops.linear_q8(xq, xs, w[key + "_q"], w[key + "_s"], w[key + "_b"], fmt, **kw)
        return ops.linear(x, w[key + "_w"], w[key + "_b"], **kw)

    def _norm_lin(self, w: Dict[str, torch.Tensor], key: str, x: torch.Tensor, h: torch.Tensor, norm_kw: dict, **kw) -> torch.Tensor:
        """`linear_key(layernorm(x, ...))`.  A quantised linear takes its bytes and scales straight from the norm kernel
        (ifx_layernorm_quant: same bytes as quantising the norm's bf16 output, which then never goes to HBM)."""
        if key + "_q" in w:
            fmt = w[key + "_fmt"]
            xq, xs = self._q8_scratch(x.shape[0], x.shape[1], x.device)
            if w.get(key + "_act") == "per_tensor":          # a tensor-wide scale needs the whole norm output first
                ops.layernorm(x, self.eps, out=h, **norm_kw)
                ops.quant_per_tensor(h, fmt, q=xq, scale=xs)
            else:
                ops.layernorm_quant(x, self.eps, fmt, q=xq, scale=xs, **norm_kw)
            return ops.linear_q8(xq, xs, w[key + "_q"], w[key + "_s"], w[key + "_b"], fmt, **kw)
        ops.layernorm(x, self.eps, out=h, **norm_kw)
        return ops.linear(h, w[key + "_w"], w[key + "_b"], **kw)

    def _run_block(self, l: int, xact: torch.Tensor, El: torch.Tensor, st: dict, meta: dict, cmeta: dict,
                   kv_cache_manager, kv_cache_requests) -> None:
        """One CausalWanAttentionBlock (causal_model.py:384-484) on the in-place activation `xact` [B*N, dim];
        `El` = (modulation + e0) of this layer, [B*Ft, 6, dim]."""
        blk = self.blocks[l]
        w = blk.w
        d, H, hd = self.dim, self.num_heads, self.head_dim
        B, N, rows_per_group, rope = st["B"], st["N"], st["rows_per_group"], st["rope"]
        sink_tokens, current_start, ctx = st["sink_tokens"], st["current_start"], st["ctx"]
        h = self._buf("h", B * N, d)
        qkv = self._buf("qkv", B * N, 3 * d)
        qb = self._buf("q", B * N, d)
        ab = self._buf("a", B * N, d)
        ub = self._buf("u", B * N, self.ffn_dim)
        # ---------------- self attention ----------------
        # `samples`: what row block b = xact[b * N:(b + 1) * N] is — its request, RoPE grid, first token and (CausVid) named slots.  One per
        # request normally; under `lockstep` (forward_pair's batched form) the blocks are TWO forwards of the same request(s), and the
        # index state moves after each of them, as it would between the two calls.
        explicit0 = st.get("explicit_slots")         # CausVid: (kv_start, kv_end) given by the caller
        samples = st.get("samples")
        if samples is None:
            samples = [dict(req=req, rope=rope, current_start=current_start, explicit=explicit0, ctx_index=b)
                       for b, req in enumerate(kv_cache_requests)]
        lockstep = bool(st.get("lockstep"))
        explicit_any = any(smp["explicit"] is not None for smp in samples)
        if not explicit_any:
            g_end, l_end = self._meta_int(meta["global_end_index"]), self._meta_int(meta["local_end_index"])
        else:
            g_end = l_end = 0
        steps_done: List[KVIndexStep] = []

        def index_state(smp=None):                   # (global_end, local_end) a sample's slot arithmetic starts from
            # the samples of ONE forward (its batch rows: separate caches, one shared index state) all start from the state that forward
            # found; the second forward of a lockstep pair starts from what the first one left
            if lockstep and smp is not None and smp.get("fwd", 0) > 0:
                prev = [stp for stp, s0 in zip(steps_done, samples) if s0.get("fwd", 0) == smp["fwd"] - 1]
                if prev:
                    return prev[-1].global_end, prev[-1].local_end
            return g_end, l_end
        step = None
        # 8-bit linears stay on the fused (norm + quantise -> one qkv GEMM) path: quantize_dynamic keeps the bf16 "qkv_w" next to
        # "qkv_q" / "qkv_s", so the test is for the QUANTISED entry — a split projection on the bf16 weights would silently bypass it
        kv_first = self.cp is not None and self.cp.kv_first and "qkv_q" not in w
        if kv_first:
            # sequence parallel: the K/V projection first, so that the exchange of the new block's K/V (side stream) runs under the
            # q projection, its norm/RoPE and the attention over the old prefix (same GEMM rows as the fused projection)
            ops.layernorm(xact, self.eps, out=h, mod=El, shift_slot=0, scale_slot=1, rows_per_group=rows_per_group)
            kvp = self._buf("kvp", B * N, 2 * d)
            ops.linear(h, w["qkv_w"][d:], w["qkv_b"][d:], out=kvp)
            name = blk.kv_cache_manager.self_name
            pending = []
            for b, smp in enumerate(samples):
                ge, le = index_state(smp)
                pst = self.cp.begin(self, l, self._kv_view(kv_cache_manager, smp["req"], name), kvp[b * N:(b + 1) * N], w, smp["rope"],
                                    smp["current_start"], ge, le, sink_tokens, kv_cache_manager, smp["req"], name)
                pending.append(pst)
                steps_done.append(pst["step"])
            qraw = self._buf("qraw", B * N, d)
            ops.linear(h, w["qkv_w"][:d], w["qkv_b"][:d], out=qraw)
            for b, pst in enumerate(pending):
                ops.rmsnorm_rope_kv_append(qraw[b * N:(b + 1) * N], w["nq"], None, self.eps, samples[b]["rope"], None, 0, d,
                                           q_out=qb[b * N:(b + 1) * N])
                step = self.cp.finish(self, pst, qb[b * N:(b + 1) * N], ab[b * N:(b + 1) * N])
        planned: List[Tuple[KVIndexStep, ops.KvCacheView]] = []
        v_direct = None

        def plan(smp):
            """The slot arithmetic of one sample, its eviction shift enqueued: (step, view)."""
            name = blk.kv_cache_manager.self_name
            view = self._kv_view(kv_cache_manager, smp["req"], name)
            ge, le = index_state(smp)
            if smp["explicit"] is not None:
                step = KVIndexStep(smp["explicit"][0], smp["explicit"][1], smp["explicit"][1], 0, 0, 0)
            else:
                step = kv_index_update(ge, le, smp["current_start"], N, view.k.shape[0], self.local_attn_size, sink_tokens)
            if step.local_start < 0 or step.local_end > view.k.shape[0]:
                raise _hip.HipKernelError(f"KV cache overflow: slots [{step.local_start}, {step.local_end}) of "
                                          f"{view.k.shape[0]} (layer {l})")
            if step.evicted:
                self._evict(kv_cache_manager, smp["req"], name, view, step)
                view = self._kv_view(kv_cache_manager, smp["req"], name)
            steps_done.append(step)
            return step, view
        if not kv_first and self.cp is None and not lockstep:
            # the slot arithmetic (and the eviction shift, which must be on the stream ahead of anything that writes the cache) first:
            # the projection below may store its V columns straight into the new block's cache rows.  (Separate requests = separate
            # caches: the order between them does not matter.  The two forwards of a lockstep pair share ONE cache — the second one's
            # eviction must stay behind the first one's attention — so they are planned one at a time, below.)
            planned = [plan(smp) for smp in samples]
            # V straight into the cache (ifx_epilogue.y2): one sample, a contiguous cache (no page table / segment map), a launch the
            # ping-pong tiles serve, bf16 weights.  The rope / append kernel then moves q and K only: a third of its 86 MB less.
            step0, view0 = planned[0]
            # (ifx_gemm_bf16 serves y2 on the automatic selection and the ping-pong variants 22..25 only: under a lab variant — the tools'
            #  IFX_GEMM_VARIANT sweeps — the V columns take the copy path instead of failing the forward)
            if (self.v_direct and len(samples) == 1 and B * N >= 2048 and "qkv_q" not in w and view0.page_table is None
                    and not view0.seg_split and d % 256 == 0 and ops.get_option("gemm_variant") in (0, 22, 23, 24, 25)):
                v_direct = view0.v.view(view0.v.shape[0], d)[step0.local_start:step0.local_start + N]
        if not kv_first:
            extra = dict(out2=v_direct, split_col=2 * d) if v_direct is not None else {}
            self._norm_lin(w, "qkv", xact, h, dict(mod=El, shift_slot=0, scale_slot=1, rows_per_group=rows_per_group), out=qkv, **extra)
        for b, smp in enumerate(() if kv_first else samples):
            req, rope_b = smp["req"], smp["rope"]
            name = blk.kv_cache_manager.self_name
            if self.cp is not None:
                view = self._kv_view(kv_cache_manager, req, name)
                ge, le = index_state(smp)
                step = self.cp.self_attention(self, l, b, view, qkv[b * N:(b + 1) * N], qb[b * N:(b + 1) * N],
                                              ab[b * N:(b + 1) * N], w, rope_b, smp["current_start"], ge, le,
                                              sink_tokens, kv_cache_manager, req, name)
                steps_done.append(step)
                continue
            step, view = planned[b] if planned else plan(smp)
            ops.rmsnorm_rope_kv_append(qkv[b * N:(b + 1) * N], w["nq"], w["nk"], self.eps, rope_b, view,
                                       step.local_start, d, q_out=qb[b * N:(b + 1) * N], v_in_place=v_direct is not None)
            ops.attention(qb[b * N:(b + 1) * N].view(N, H, hd), view, step.local_end, scale=st.get("attn_scale", 0.0),
                          out=ab[b * N:(b + 1) * N].view(N, H, hd), tag="attn_self")
        if not explicit_any:
            self._meta_set(meta, "global_end_index", step.global_end)
            self._meta_set(meta, "local_end_index", step.local_end)
        if self.index_trace is not None and l == 0:   # tests: (current_start, global_end, local_end) of every FORWARD, in cache order
            for smp, stp in zip(samples, steps_done if steps_done else [step] * len(samples)):
                if smp["ctx_index"] == 0:
                    self.index_trace.append((int(smp["current_start"]), int(stp.global_end), int(stp.local_end)))
        self._lin(w, "o", ab, epilogue=_hip.IFX_EPI_GATE_RES, residual=xact, mod=El, gate_slot=2,
                  rows_per_group=rows_per_group, out=xact)
        # ---------------- cross attention ----------------
        self._norm_lin(w, "cq", xact, h, dict(gamma=w["n3_w"], beta=w["n3_b"]), out=qb)
        ops.rmsnorm(qb, w["cnq"], self.eps, out=qb)
        b = 0
        while b < len(samples):
            req = samples[b]["req"]
            nb = 1                                   # consecutive row blocks of the SAME request (lockstep pair): one launch over all of them
            while b + nb < len(samples) and samples[b + nb]["req"] is req:
                nb += 1
            cview = self._kv_view(kv_cache_manager, req, blk.kv_cache_manager.cross_name)
            if not cmeta["is_init"]:
                ci = samples[b]["ctx_index"]
                cb = ctx[ci * self.text_len:(ci + 1) * self.text_len]
                kx = self._lin(w, "ck", cb)
                ops.rmsnorm(kx, w["cnk"], self.eps, out=cview.k.view(self.text_len, d))
                self._lin(w, "cv", cb, out=cview.v.view(self.text_len, d))
            nkeys, mult = self._cross_dedup.get(req.request_id, (self.text_len, 1)) if self.cross_dedup else (self.text_len, 1)
            qv, av = qb[b * N:(b + nb) * N].view(nb * N, H, hd), ab[b * N:(b + nb) * N].view(nb * N, H, hd)
            if mult > 1:      # the zero-padded context rows are ONE key with a multiplicity (ifx_attn_fwd_dedup)
                ops.attention_dedup(qv, cview, nkeys, mult, out=av, tag="attn_cross")
            else:
                ops.attention(qv, cview, self.text_len, out=av, tag="attn_cross")
            b += nb
        cmeta["is_init"] = True
        self._lin(w, "co", ab, epilogue=_hip.IFX_EPI_RESIDUAL, residual=xact, out=xact)
        # ---------------- feed forward ----------------
        self._norm_lin(w, "f0", xact, h, dict(mod=El, shift_slot=3, scale_slot=4, rows_per_group=rows_per_group),
                       epilogue=_hip.IFX_EPI_GELU_TANH, out=ub)
        self._lin(w, "f2", ub, epilogue=_hip.IFX_EPI_GATE_RES, residual=xact, mod=El, gate_slot=5,
                  rows_per_group=rows_per_group, out=xact)


    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x, t, context, seq_len=None, clip_fea=None, y=None, kv_cache_meta=None,
                crossattn_cache_meta=None, current_start: int = 0, cache_start=None,
                kv_cache_manager: Optional[KVCacheManager] = None,
                kv_cache_requests: Optional[Sequence[KVCacheRequest]] = None,
                kv_start: Optional[int] = None, kv_end: Optional[int] = None,
                current_end: Optional[int] = None) -> torch.Tensor:
        """x: [B, C, F, H, W] tensor (or list of [C, F, H, W]); t: [B, F]; context: [B, L, text_dim] tensor or
        list of [L_i, text_dim].  Returns the flow prediction [B, C_out, F, H, W] (bf16)."""
        fw = self._prologue(x, t, context, kv_cache_meta, crossattn_cache_meta, current_start, kv_cache_manager, kv_cache_requests,
                            kv_start, kv_end)
        with self._small_split_scope():               # shard-sized launches of a sequence-parallel rank only (attach_sequence_parallel)
            for l in range(self.num_layers):
                self._run_block(l, fw["xact"], fw["E"][l], fw["st"], fw["kv_meta"][l], fw["cross_meta"][l], kv_cache_manager,
                                kv_cache_requests)
        return self._epilogue(fw)

    def _small_split_scope(self):
        import contextlib
        return ops.option_scope("gemm_small_split", 1) if getattr(self.cp, "gemm_small_split", False) else contextlib.nullcontext()

    @torch.no_grad()
    def forward_pair(self, first: dict, second: dict):
        """TWO forwards enqueued layer by layer on two streams — `first` on the current stream, `second` on a side stream whose layer l
        starts when `first`'s layer l has been enqueued and finished (an event per layer) — for the one pair of generator calls of the
        block loop that depend on each other ONLY through the cache: the clean-context re-run of block b (it writes block b's K / V
        rows, layer by layer) and the first denoising step of block b + 1 (whose layer l attends to those rows of layer l and writes
        the rows of block b + 1).  Same launches, same arguments, same order per cache as the two calls one after the other: the
        results are bit-identical to them (tests/test_hip_model.py); what changes is that the second chain's launches fill the CUs and
        the dependent-launch gaps the first one leaves — a sequence-parallel rank's launches (585 rows at P = 8) occupy a fraction of
        the chip each.  `first` / `second`: the keyword arguments of `forward`.  Returns both flow predictions."""
        if self._pair_mode() == "lockstep":
            return self._forward_pair_lockstep(first, second)
        dev = self.device_
        main = torch.cuda.current_stream(dev)
        if self._pair_stream is None:
            self._pair_stream = torch.cuda.Stream(device=dev)
        side = self._pair_stream
        side.wait_stream(main)                        # the second forward's inputs were produced on the caller's stream
        kvm, reqs = first["kv_cache_manager"], first["kv_cache_requests"]
        assert second["kv_cache_manager"] is kvm, "forward_pair: both forwards must address the same cache"

        def pro(kw):
            return self._prologue(kw["x"], kw["t"], kw["context"], kw.get("kv_cache_meta"), kw.get("crossattn_cache_meta"),
                                  kw.get("current_start", 0), kw["kv_cache_manager"], kw["kv_cache_requests"], kw.get("kv_start"),
                                  kw.get("kv_end"))
        try:
            fa = pro(first)
            self._chain = 1
            with torch.cuda.stream(side):
                fb = pro(second)
            with self._small_split_scope():
                for l in range(self.num_layers):
                    self._chain = 0
                    self._run_block(l, fa["xact"], fa["E"][l], fa["st"], fa["kv_meta"][l], fa["cross_meta"][l], kvm, reqs)
                    ev = torch.cuda.Event()
                    ev.record(main)
                    self._chain = 1
                    with torch.cuda.stream(side):
                        side.wait_event(ev)           # layer l of the cache holds `first`'s rows before `second`'s layer l reads / evicts
                        self._run_block(l, fb["xact"], fb["E"][l], fb["st"], fb["kv_meta"][l], fb["cross_meta"][l], kvm,
                                        second["kv_cache_requests"])
            self._chain = 0
            ya = self._epilogue(fa)
            self._chain = 1
            with torch.cuda.stream(side):
                yb = self._epilogue(fb)
        finally:
            self._chain = 0
        main.wait_stream(side)
        yb.record_stream(main)                        # allocated on the side stream, consumed (and later freed) on the caller's
        return ya, yb

    def _pair_mode(self) -> str:
        """How forward_pair runs its two forwards.  "streams" (default): two launch chains on two streams, layer-interleaved —
        bit-identical to the sequential calls; the second chain fills the tails and the dependent-launch gaps of the first: - 2.2 % /
        - 3 % of a clip on one GPU (two boxes), - 4 % on an emulated rank of 8.  "lockstep": ONE chain over both forwards' rows — every
        row-local kernel (LayerNorms, the six GEMMs, RMSNorms, the cross-attention) runs once on 2 N rows, the self-attention part once
        per forward in call order.  Built for the sequence-parallel rank (585 rows at P = 8: in the lab a layer's GEMMs take 110 us
        for one forward and 168 us for two) and MEASURED behind the two-stream form there (283 vs 279 ms per clip at P = 8, 408 vs 407
        at P = 4, 626 vs 620 at P = 2: inside the clip the 1170-row launches cost what two 585-row launches overlapped on two streams
        cost), so it is not the default anywhere; kept, tested bit for bit against the sequential calls, selectable through the
        `pair_mode` attribute / env IFX_PAIR_MODE."""
        import os
        m = getattr(self, "pair_mode", None) or os.environ.get("IFX_PAIR_MODE", "")
        return m if m in ("streams", "lockstep") else "streams"

    def _forward_pair_lockstep(self, first: dict, second: dict):
        """forward_pair as one launch chain over the rows of both forwards (see _pair_mode).  Row block b < B is `first`'s sample b, row
        block B + b `second`'s; the cache index state moves after each block's self-attention, exactly as between the two calls.  The
        arithmetic of a row is the sequential call's; the GEMM tile (hence, where a tile splits K, the fp32 summation order) follows the
        launch's row count as it does everywhere else: bit-identical to the sequential calls where the tile choice is row-invariant
        (one GPU), within summation-order noise on a sequence-parallel rank (gemm_small_split)."""
        kvm, reqs = first["kv_cache_manager"], list(first["kv_cache_requests"])
        assert second["kv_cache_manager"] is kvm and list(second["kv_cache_requests"]) == reqs, \
            "forward_pair: both forwards must address the same requests"
        d = self.dim

        def pro(kw, out):
            return self._prologue(kw["x"], kw["t"], kw["context"], kw.get("kv_cache_meta"), kw.get("crossattn_cache_meta"),
                                  kw.get("current_start", 0), kvm, reqs, kw.get("kv_start"), kw.get("kv_end"), xact_out=out)
        lat = first["x"] if isinstance(first["x"], torch.Tensor) else torch.stack(list(first["x"]))
        B = lat.shape[0]
        pt_, ph, pw = self.patch_size
        rows = B * (lat.shape[2] // pt_) * (lat.shape[3] // ph) * (lat.shape[4] // pw) // self.parallel_config.world_size
        x2 = self._buf("x2", 2 * rows, d)
        fa = pro(first, x2[:rows])
        self._chain = 1                               # the second prologue's scratch (its `h`) must not alias the first's
        try:
            fb = pro(second, x2[rows:])
        finally:
            self._chain = 0
        if fa["N"] != fb["N"] or fa["rows_per_group"] != fb["rows_per_group"] or fa["grid"] != fb["grid"]:
            raise ValueError("forward_pair (lockstep): the two forwards must have the same latent geometry and timestep layout")
        N = fa["N"]
        E2 = torch.cat([fa["E"], fb["E"]], dim=1).contiguous()              # [L, 2 B Ft, 6, d]
        eh2 = torch.cat([fa["eh"], fb["eh"]], dim=0).contiguous()
        sa, sb = fa["st"], fb["st"]
        samples = [dict(req=r, rope=s_["rope"], current_start=s_["current_start"], explicit=s_.get("explicit_slots"), ctx_index=b, fwd=f)
                   for f, s_ in enumerate((sa, sb)) for b, r in enumerate(reqs)]
        st = dict(sa, B=2 * B, samples=samples, lockstep=True, ctx=sa["ctx"] if sa["ctx"] is not None else sb["ctx"])
        if sa["ctx"] is None and sb["ctx"] is not None or any(not m["is_init"] for m in fa["cross_meta"]):
            raise RuntimeError("forward_pair: the cross-attention cache must be initialised before two forwards are paired")
        with self._small_split_scope():
            for l in range(self.num_layers):
                self._run_block(l, x2, E2[l], st, fa["kv_meta"][l], fa["cross_meta"][l], kvm, reqs)
        h2 = self._buf("h", 2 * B * N, d)
        ops.layernorm(x2, self.eps, mod=eh2, shift_slot=0, scale_slot=1, rows_per_group=fa["rows_per_group"], out=h2)
        yv = self._lin(self.g, "head", h2)
        if self.cp is not None:
            yv = self.cp.gather_head(yv, 2 * B, fa["F_"])
        y = C.unpatchify(yv, 2 * B, fa["grid"], self.patch_size, self.out_dim)
        return y[:B], y[B:]

    def _prologue(self, x, t, context, kv_cache_meta, crossattn_cache_meta, current_start, kv_cache_manager, kv_cache_requests,
                  kv_start=None, kv_end=None, xact_out: Optional[torch.Tensor] = None) -> dict:
        """Everything of a forward in front of the layers: embeddings, modulation tables, the text context (first call), the per-forward
        state the layers share.  Enqueues on the current stream, takes its scratch from chain `self._chain`."""
        explicit = None
        if kv_start is not None and kv_end is not None:
            # CausVid addressing (models/causvid/causal_model.py:128-179,258-277): the caller names the cache
            # slots of this block; the cross-attention init flag lives on the block (`is_cross_attn_init`)
            explicit = (int(kv_start), int(kv_end))
            kv_cache_meta = [dict() for _ in range(self.num_layers)]
            crossattn_cache_meta = [b.cross_meta for b in self.blocks]
        if kv_cache_meta is None:
            raise NotImplementedError("HipCausalWanModel implements the KV-cached inference path only")
        assert kv_cache_manager is not None and kv_cache_requests is not None
        if self.mod_all is None:
            raise RuntimeError("HipCausalWanModel: weights not loaded (call load_state_dict first)")
        dev, d, L = self.device_, self.dim, self.num_layers
        lat = torch.stack(list(x)) if not isinstance(x, torch.Tensor) else x
        lat = lat.to(device=dev, dtype=BF16)
        B, _, nf, hh, ww = lat.shape
        pt_, ph, pw = self.patch_size
        grid = (nf // pt_, hh // ph, ww // pw)
        F_, fs = grid[0], grid[1] * grid[2]
        cp_ws = self.parallel_config.world_size
        cp_rank = self.parallel_config.rank
        if fs % cp_ws != 0:
            raise ValueError(f"sequence parallelism splits every frame's {fs} tokens over {cp_ws} ranks: not divisible")
        hw_local = fs // cp_ws
        fs_l = hw_local                               # tokens per frame on this rank
        N = F_ * fs_l                                 # tokens per sample on this rank
        assert len(kv_cache_requests) == B

        # ---- embeddings ------------------------------------------------------------------
        patches = C.patchify(lat, self.patch_size)                               # [B*F*fs, 64]
        if cp_ws > 1:                                                            # keep hw-slice `rank` of every frame
            patches = patches.view(B * F_, fs, -1)[:, cp_rank * hw_local:(cp_rank + 1) * hw_local].reshape(B * N, -1)
        xact = ops.linear(patches.contiguous(), self.g["patch_w"], self.g["patch_b"],
                          out=self._buf("x", B * N, d) if xact_out is None else xact_out)
        Ft = t.shape[1]                                                          # frames carrying a timestep (F or 1)
        rows_per_group = (F_ // Ft) * fs_l
        # The modulation tables are a function of the timestep VALUES alone (and of the weights): a clip has five distinct ones (four
        # denoising steps + the clean-context re-run) over 35 forwards.  Memoised for the pipelines' own constant timestep tensors only
        # (schedulers.const_timestep: every entry = one scalar the host knows), keyed on that scalar + shape — never on a storage
        # address or a version counter, which a raw-pointer write would not move and inference tensors do not have.  Any other
        # timestep tensor is embedded every call.  ~16 glue launches (sinusoid, three small GEMMs, SiLUs, the two table adds) per
        # forward less; the same bits.
        from ..schedulers import carry_tag, const_tag
        t_in = t
        t = carry_tag(t_in, t.to(dev))
        tc = const_tag(t)
        tkey = None if tc is None else (tc, tuple(t.shape), t.dtype, self.g["time0_w"].data_ptr(), self.g.get("time0_fmt"))
        hit = self._temb_cache.get(tkey) if tkey is not None else None
        stream_now = ops._stream()
        if hit is not None:
            E, eh = hit[1], hit[2]
            if hit[3] != stream_now:                  # made on the other chain's stream: order this stream behind the kernels that wrote
                cur = torch.cuda.current_stream(dev)  # them (record_stream guards the allocation, not the data) and tell the
                if hit[0] is not None:                # allocator this stream reads them too
                    cur.wait_event(hit[0])
                E.record_stream(cur)
                eh.record_stream(cur)
        else:
            emb = C.sinusoidal_embedding_1d(self.freq_dim, t.flatten()).to(BF16)      # [B*F, freq_dim]
            # time_embedding / time_projection (a handful of rows): the same MFMA GEMM as the block linears, SiLU is elementwise glue
            e = self._lin(self.g, "time2", F.silu(self._lin(self.g, "time0", emb.contiguous())))
            e0 = self._lin(self.g, "tproj", F.silu(e)).unflatten(1, (6, d))                         # [B*Ft, 6, d]
            E = (self.mod_all + e0.unsqueeze(0)).contiguous()                        # [L, B*Ft, 6, d] bf16
            eh = (self.g["head_mod"] + e.unsqueeze(1)).contiguous()                  # [B*Ft, 2, d]
            if tkey is not None:
                if len(self._temb_cache) >= 16:
                    self._temb_cache.pop(next(iter(self._temb_cache)))
                ready = torch.cuda.Event() if t.is_cuda else None
                if ready is not None:
                    ready.record(torch.cuda.current_stream(dev))
                self._temb_cache[tkey] = (ready, E, eh, stream_now)

        need_ctx = any(not m["is_init"] for m in crossattn_cache_meta)
        ctx = None
        if need_ctx:
            if isinstance(context, torch.Tensor):
                context = list(context)
            padded = torch.stack([torch.cat([u.to(dev, BF16), torch.zeros(self.text_len - u.size(0), u.size(1),
                                                                       device=dev, dtype=BF16)]) for u in context])
            if self.cross_dedup:
                # rows behind the prompt's own tokens are the zero padding: identical inputs -> identical text_embedding rows ->
                # identical K / V rows in every layer.  Count the trailing run once per prompt (one host read per request and
                # cache initialisation); cross-attention then runs over the distinct rows + one key with that multiplicity.
                same = (padded == padded[:, -1:]).all(-1)                        # [B, text_len]
                idx = torch.arange(self.text_len, device=dev).expand(B, -1)
                first = (torch.where(same, -1, idx).amax(1) + 1).tolist()         # first row of the trailing identical run
                live = getattr(kv_cache_manager, "request_to_kv_caches", None)
                if live is not None:                                             # entries of requests the manager has freed since
                    for rid in [r for r in self._cross_dedup if r not in live]:
                        del self._cross_dedup[rid]
                for b, req in enumerate(kv_cache_requests or []):
                    j0 = min(int(first[b]), self.text_len - 1)
                    self._cross_dedup[req.request_id] = (j0 + 1, self.text_len - j0)
            c0 = self._lin(self.g, "text0", padded.view(B * self.text_len, -1), epilogue=_hip.IFX_EPI_GELU_TANH)
            ctx = self._lin(self.g, "text2", c0)                                 # [B*text_len, d]

        # ---- scratch ---------------------------------------------------------------------
        h = self._buf("h", B * N, d)
        start_frame = current_start // fs
        # q leaves the norm / RoPE kernel already multiplied by scale * log2(e) (one rounding to bf16 either way) and the
        # attention is called with scale = ln 2 — the same softmax(q k^T / sqrt(d)), without a scale-FMA per score in the attention loop
        q_scale, attn_scale = ops.attn_q_prescale(self.head_dim) if self.q_prescale else (0.0, 0.0)
        self._attn_scale = attn_scale                 # the sequence-parallel attention calls (sequence_parallel.finish) read it here
        rope = ops.RopeGridSpec(self.freqs, start_frame, grid[1], grid[2], cp_rank * hw_local, hw_local, q_scale)
        sink_tokens = self.sink_size * fs

        st = dict(B=B, N=N, F_=F_, fs=fs, rows_per_group=rows_per_group, rope=rope, sink_tokens=sink_tokens,
                  current_start=current_start, ctx=ctx, explicit_slots=explicit, attn_scale=attn_scale)
        return dict(xact=xact, E=E, eh=eh, h=h, st=st, kv_meta=kv_cache_meta, cross_meta=crossattn_cache_meta, grid=grid, B=B, F_=F_,
                    rows_per_group=rows_per_group, N=N, requests=list(kv_cache_requests))

    def _epilogue(self, fw: dict) -> torch.Tensor:
        """The head behind the layers (modulated LayerNorm, projection, gather of the sequence-parallel shards, unpatchify)."""
        ops.layernorm(fw["xact"], self.eps, mod=fw["eh"], shift_slot=0, scale_slot=1, rows_per_group=fw["rows_per_group"], out=fw["h"])
        yv = self._lin(self.g, "head", fw["h"])                                   # [B*N, out*prod(patch)]
        if self.cp is not None:
            yv = self.cp.gather_head(yv, fw["B"], fw["F_"])
        return C.unpatchify(yv, fw["B"], fw["grid"], self.patch_size, self.out_dim)
