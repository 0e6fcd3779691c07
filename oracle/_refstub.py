"""TEST INFRASTRUCTURE — not product code.

Import shim that lets the *reference's own* Python hot path
(/root/reference/inferix/...) be imported on CPU inside the build container,
where most of its third-party dependencies (diffusers, yunchang, xfuser,
omegaconf, easydict, torchvision, ...) are not installed.

It is used ONLY by `oracle/gen_golden.py` (fixture generation) and by the
optional `tests/test_oracle_vs_reference_live.py` (skipped when
/root/reference is absent, i.e. always on the GPU box).  Nothing of the
reference is copied: missing third-party modules are replaced by empty
stand-in modules whose attributes are inert dummies, which is enough because
the hot path never calls into them on a single CPU rank.

Recipe follows SURVEY.md §8c / Appendix A.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("INFERIX_REFERENCE_ROOT", "/root/reference")

# third-party top-level packages the reference imports but this image lacks.
# flash_attn / flash_attn_interface / flashinfer / magi_attention are NOT
# stubbed on purpose: upstream guards them with try/except and stubbing them
# would flip HAS_FLASH_ATTN and break the SDPA fallback the oracle relies on.
_STUB_TOPLEVEL = {
    "diffusers", "yunchang", "xfuser", "easydict", "torchvision", "ftfy",
    "imageio", "dashscope", "torchdiffeq", "lmdb", "decord", "av",
    "omegaconf", "gradio", "cv2", "timm", "pynvml", "aiortc", "ffmpeg",
    "bs4", "dax", "moviepy", "PIL", "html", "loguru",
}


class _AttrDict(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


class _Meta(type):
    # enum-like access at class-definition time (e.g. AttnType.FA)
    def __getattr__(cls, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return k


class _Dummy(metaclass=_Meta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self


_SPECIAL = {
    "ConfigMixin": type("ConfigMixin", (), {}),
    "ModelMixin": type("ModelMixin", (torch.nn.Module,), {}),
    "register_to_config": (lambda f: f),
    "EasyDict": _AttrDict,
    "KarrasDiffusionSchedulers": [],
    "SchedulerMixin": type("SchedulerMixin", (), {}),
    "fix_text": (lambda s: s),
}


class _StubModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        if k in _SPECIAL:
            return _SPECIAL[k]
        v = type(k, (_Dummy,), {})
        setattr(self, k, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        top = name.split(".")[0]
        if top in _STUB_TOPLEVEL and top not in _REAL:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_REAL: set = set()
_installed = False


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "inferix"))


def install() -> None:
    """Make `import inferix...` resolve to the reference tree on CPU."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    # things that are really installed must win over the stubs
    for top in list(_STUB_TOPLEVEL):
        try:
            if importlib.util.find_spec(top) is not None:
                _REAL.add(top)
        except (ImportError, ValueError):
            pass
    import transformers  # noqa: F401  (probes torchvision via find_spec: do it before stubbing)
    from transformers import AutoTokenizer  # noqa: F401
    # reference evaluates torch.cuda.current_device() as a default argument at
    # class-definition time (wan_base/text_encoder/t5.py)
    torch.cuda.current_device = lambda: 0
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def import_hot_path():
    """Return the reference modules of the hot path, patched for CPU execution."""
    install()
    import importlib

    import torch.nn.functional as F

    cm = importlib.import_module("inferix.models.self_forcing.causal_model")
    attn_pkg = importlib.import_module("inferix.models.attention")

    # cross-attention calls flash_attention() directly, which asserts CUDA;
    # route it to the same SDPA the reference's own `attention()` falls back to.
    def _sdpa(q, k, v, q_lens=None, k_lens=None, **kw):
        out = F.scaled_dot_product_attention(
            q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
        return out.transpose(1, 2).contiguous()

    attn_pkg.flash_attention = _sdpa
    return cm


# ---------------------------------------------------------------------------------------------------------------------
# MAGI: `inferix/models/magi/dit/dit_module.py` imports flash-attn, flashinfer and triton UNCONDITIONALLY (lines 19-30).
# triton is installed here (its kernels cannot run without a GPU); flash-attn and flashinfer are not.  The stand-ins
# below make the import succeed and give the module's own code something to call on CPU.  They restate the PUBLISHED
# definition of each third-party function (both packages are unpinned, requirements-torch.txt:8 / Installation.md:27):
#   flash_attn.layers.rotary.apply_rotary_emb(x, cos, sin)   non-interleaved rotary: with x = [x1 | x2] over the first
#       2*cos.shape[-1] channels, out = [x1*cos - x2*sin | x1*sin + x2*cos], remaining channels untouched, fp32 math
#   flash_attn_func / flash_attn_varlen_func                 softmax(q k^T / sqrt(d)) v, no mask, grouped-query heads
#   flashinfer.activation.silu_and_mul(x)                    silu(x[..., :d]) * x[..., d:]
#   flashinfer.gemm.bmm_fp8(A, B, A_scale, B_scale, dtype)   (A.float() @ B.float()) * A_scale * B_scale -> dtype
# Everything else a MAGI fixture pins — layer norms, projections, head layouts, the gate, post-norms, the cache rule,
# the range bookkeeping, div_clamp_to — is the reference's own code, executed as is.
# ---------------------------------------------------------------------------------------------------------------------
_magi_installed = False


def install_magi():
    """`install()` + stand-ins for the third-party imports of the MAGI DiT module.  Call in a process that does NOT also
    generate the Wan fixtures: stubbing flash_attn flips the try/except guards of inferix/models/attention/backends.py."""
    global _magi_installed
    install()
    if _magi_installed:
        return
    import math

    import torch.nn.functional as F

    def apply_rotary_emb(x, cos, sin, interleaved=False, inplace=False, seqlen_offsets=0, cu_seqlens=None, max_seqlen=None):
        assert not interleaved and cu_seqlens is None and seqlen_offsets == 0
        ro = cos.shape[-1] * 2
        seqlen = x.shape[1]
        c = cos[:seqlen].float()[None, :, None, :]
        s = sin[:seqlen].float()[None, :, None, :]
        x1, x2 = x[..., : ro // 2].float(), x[..., ro // 2: ro].float()
        out = torch.cat([x1 * c - x2 * s, x1 * s + x2 * c, x[..., ro:].float()], dim=-1)
        return out.to(x.dtype)

    def _sdpa(q, k, v):                      # [b, s, h, d]; grouped-query heads repeated
        rep = q.shape[2] // k.shape[2]
        if rep > 1:
            k, v = k.repeat_interleave(rep, dim=2), v.repeat_interleave(rep, dim=2)
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                           scale=1.0 / math.sqrt(q.shape[-1]))
        return o.transpose(1, 2).contiguous()

    def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, deterministic=False, **kw):
        assert not causal and softmax_scale is None
        return _sdpa(q, k, v)

    def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                               softmax_scale=None, causal=False, deterministic=False, **kw):
        assert not causal and softmax_scale is None
        cq, ck = cu_seqlens_q.tolist(), cu_seqlens_k.tolist()
        out = torch.empty_like(q)
        for i in range(len(cq) - 1):
            out[cq[i]:cq[i + 1]] = _sdpa(q[None, cq[i]:cq[i + 1]], k[None, ck[i]:ck[i + 1]], v[None, ck[i]:ck[i + 1]])[0]
        return out

    def silu_and_mul(x):
        d = x.shape[-1] // 2
        return F.silu(x[..., :d]) * x[..., d:]

    def bmm_fp8(A, B, A_scale, B_scale, dtype, out=None, backend="cublas"):
        return ((A.float() @ B.float()) * A_scale.flatten()[0] * B_scale.flatten()[0]).to(dtype)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    fa = mod("flash_attn", flash_attn_func=flash_attn_func, flash_attn_varlen_func=flash_attn_varlen_func)
    fa.flash_attn_interface = mod("flash_attn.flash_attn_interface", flash_attn_func=flash_attn_func,
                                  flash_attn_varlen_func=flash_attn_varlen_func)
    fa.layers = mod("flash_attn.layers")
    fa.layers.rotary = mod("flash_attn.layers.rotary", apply_rotary_emb=apply_rotary_emb)
    fi = mod("flashinfer")
    fi.activation = mod("flashinfer.activation", silu_and_mul=silu_and_mul)
    fi.gemm = mod("flashinfer.gemm", bmm_fp8=bmm_fp8)
    torch.cuda.get_device_capability = lambda *a, **k: (8, 0)      # dit_module.py:982: selects the flash_attn_func branch
    _magi_installed = True


def import_magi_dit(cp_world: int = 1, cp_rank: int = 0):
    """The reference's MAGI DiT module, patched for one CPU rank: parallel_state answers (cp_world, cp_rank) and a single
    pipeline stage; `range_mod_triton` (a Triton kernel: needs a GPU) is replaced by the indexing it performs —
    y[row] = x[row] * gatings[map[row]] (dit_module.py:204-292)."""
    install_magi()
    import importlib
    dm = importlib.import_module("inferix.models.magi.dit.dit_module")
    ps = dm.parallel_state
    ps.get_tp_world_size = lambda with_context_parallel=False: cp_world
    ps.get_cp_world_size = lambda: cp_world
    ps.get_cp_rank = lambda: cp_rank
    ps.get_pp_rank = lambda: 0
    ps.get_pp_world_size = lambda: 1

    def range_mod(x, c_mapping, gatings):
        s, b, h = x.shape
        xf = x.transpose(0, 1).flatten(0, 1)
        mp = c_mapping.transpose(0, 1).flatten(0, 1).long()
        g = gatings.flatten(0, 1)
        return (xf * g[mp]).reshape(b, s, h).transpose(0, 1)

    dm.range_mod_triton = range_mod
    return dm
