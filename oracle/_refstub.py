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
