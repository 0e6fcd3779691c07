"""ctypes binding of libinferix_hip.so (C-ABI declared in include/inferix_hip.h).

The HIP library is the product: there is NO fallback.  Importing this module
without the built library, or calling an op on a non-GPU tensor, raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# torch bundles its own libamdhip64: it MUST be in the process before libinferix_hip.so is dlopen'ed so
# that both share one HIP runtime (loading /opt/rocm's copy first leaves torch without a device).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IFX_HIP_LIB", os.path.join(_HERE, "libinferix_hip.so"))   # override: kernel studies only

IFX_LN_PLAIN, IFX_LN_AFFINE, IFX_LN_MODULATE = 0, 1, 2
IFX_EPI_BIAS, IFX_EPI_GELU_TANH, IFX_EPI_RESIDUAL, IFX_EPI_GATE_RES, IFX_EPI_GELU_ERF = 0, 1, 2, 3, 4
IFX_ACT_SILU, IFX_ACT_TANH = 0, 1
IFX_Q_FP8_E4M3, IFX_Q_INT8 = 0, 1


class HipLibraryMissing(RuntimeError):
    pass


class HipKernelError(RuntimeError):
    pass


class KvView(C.Structure):
    """ifx_kv_view"""
    _fields_ = [("k", C.c_void_p), ("v", C.c_void_p), ("page_table", C.c_void_p),
                ("page_size", C.c_int32), ("num_slots", C.c_int32), ("kv_heads", C.c_int32),
                ("head_dim", C.c_int32), ("seg_split", C.c_int32), ("seg_delta", C.c_int32)]


class RopeGrid(C.Structure):
    """ifx_rope_grid"""
    _fields_ = [("freqs", C.c_void_p), ("max_pos", C.c_int32), ("start_frame", C.c_int32),
                ("height", C.c_int32), ("width", C.c_int32), ("hw_offset", C.c_int32),
                ("hw_local", C.c_int32), ("q_scale", C.c_float), ("flags", C.c_int32)]


class Conv3dDesc(C.Structure):
    """ifx_conv3d_desc"""
    _fields_ = [("x", C.c_void_p), ("in_frame_stride", C.c_int64), ("in_slots", C.POINTER(C.c_int32)),
                ("hs", C.c_int32), ("ws", C.c_int32), ("cin", C.c_int32), ("upsample", C.c_int32),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("kt", C.c_int32), ("ks", C.c_int32),
                ("y", C.c_void_p), ("out_frame_stride", C.c_int64), ("out_slots", C.POINTER(C.c_int32)),
                ("cout", C.c_int32), ("t_out", C.c_int32), ("residual", C.c_void_p), ("zero_page", C.c_void_p),
                ("in_planar", C.c_int32)]


class Epilogue(C.Structure):
    """ifx_epilogue"""
    _fields_ = [("epilogue", C.c_int32), ("residual", C.c_void_p), ("ld_res", C.c_int32),
                ("mod", C.c_void_p), ("mod_slots", C.c_int32), ("gate_slot", C.c_int32),
                ("rows_per_group", C.c_int32), ("y2", C.c_void_p), ("ldy2", C.c_int32), ("split_col", C.c_int32)]


class MagiHeadPrepDesc(C.Structure):
    """ifx_magi_head_prep_desc"""
    _fields_ = [("inp", C.c_void_p), ("ld_in", C.c_int32), ("rows", C.c_int32), ("layout", C.c_int32),
                ("q_heads", C.c_int32), ("kv_heads", C.c_int32), ("head_dim", C.c_int32), ("rope", C.c_void_p),
                ("qn_w", C.c_void_p), ("qn_b", C.c_void_p), ("kn_w", C.c_void_p), ("kn_b", C.c_void_p),
                ("xn_w", C.c_void_p), ("xn_b", C.c_void_p), ("eps", C.c_float), ("layernorm_1p", C.c_int32),
                ("q_out", C.c_void_p), ("ld_q", C.c_int32), ("qx_out", C.c_void_p), ("ld_qx", C.c_int32),
                ("k_out", C.c_void_p), ("v_out", C.c_void_p), ("ld_kv", C.c_int32), ("kv_head_stride", C.c_int32),
                ("row0", C.c_int32), ("split", C.c_int32), ("row1", C.c_int32), ("q_scale", C.c_float), ("rope_half", C.c_int32),
                ("q_group", C.c_int32), ("q_group_stride", C.c_int64)]


IFX_MAX_PEERS, IFX_PEER_HANDLE_BYTES = 8, 64


class PeerCaches(C.Structure):
    """ifx_peer_caches"""
    _fields_ = [("count", C.c_int32), ("k", C.c_void_p * IFX_MAX_PEERS), ("v", C.c_void_p * IFX_MAX_PEERS)]


class PeerFlags(C.Structure):
    """ifx_peer_flags"""
    _fields_ = [("count", C.c_int32), ("flags", C.c_void_p * IFX_MAX_PEERS)]


# name -> (restype, argtypes); the complete export list of include/inferix_hip.h
_vp, _i32, _f32 = C.c_void_p, C.c_int32, C.c_float
SIGNATURES = {
    "ifx_version": (C.c_int, []),
    "ifx_last_error": (C.c_char_p, []),
    "ifx_arch": (C.c_char_p, []),
    "ifx_set_option": (C.c_int, [C.c_char_p, _i32]),
    "ifx_get_option": (C.c_int, [C.c_char_p, C.POINTER(_i32)]),
    "ifx_device_error": (_i32, [_i32]),
    "ifx_attn_fwd_paged": (C.c_int, [_vp, _vp, _vp, C.POINTER(KvView), _i32, _i32, _i32, _i32, _f32, _vp]),
    "ifx_attn_split_plan": (_i32, [_i32, _i32, _i32, _i32, C.POINTER(C.c_int64)]),
    "ifx_attn_fwd_paged_split": (C.c_int, [_vp, _vp, _vp, C.POINTER(KvView), _i32, _i32, _i32, _i32, _f32, _i32, _vp,
                                           C.c_int64, _vp]),
    "ifx_conv3d_cl": (C.c_int, [C.POINTER(Conv3dDesc), _vp]),
    "ifx_rmsnorm_cl": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.POINTER(_i32), _i32, _i32, _i32, _i32, _vp]),
    "ifx_softmax_rows": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "ifx_t5_attention": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "ifx_t5_gated_gelu": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "ifx_kv_scatter_shards": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(KvView), _vp]),
    "ifx_peer_alloc": (C.c_int, [C.c_int64, _i32, C.POINTER(C.c_void_p)]),
    "ifx_peer_free": (C.c_int, [_vp]),
    "ifx_peer_export": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "ifx_peer_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "ifx_peer_close": (C.c_int, [_vp]),
    "ifx_rmsnorm_rope_kv_push": (C.c_int, [_vp, _i32, _vp, C.POINTER(RopeGrid), C.POINTER(PeerCaches), C.POINTER(KvView), _i32, _i32,
                                           _i32, _i32, _i32, _i32, _f32, _vp]),
    "ifx_peer_signal": (C.c_int, [C.POINTER(PeerFlags), _i32, _i32, _vp]),
    "ifx_peer_wait": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "ifx_attn_fwd_partial": (C.c_int, [_vp, C.POINTER(KvView), _i32, _i32, _i32, _i32, _f32, _i32, _vp, C.c_int64, _i32,
                                       _i32, C.POINTER(_i32), _vp]),
    "ifx_attn_merge_partials": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "ifx_lse_merge": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "ifx_rmsnorm_rope_kv_append": (C.c_int, [_vp, _i32, _vp, _vp, _vp, C.POINTER(RopeGrid), C.POINTER(KvView),
                                             _i32, _i32, _i32, _f32, _vp]),
    "ifx_rmsnorm": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _f32, _vp]),
    "ifx_layernorm": (C.c_int, [_vp, _vp, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ifx_gemm_bf16": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.POINTER(Epilogue), _vp]),
    "ifx_kv_roll": (C.c_int, [C.POINTER(KvView), _i32, _i32, _i32, _vp, _vp]),
    "ifx_quant_per_token": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    "ifx_layernorm_quant": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ifx_gemm_q8": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(Epilogue), _vp]),
    "ifx_gemm_q8_ws": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(Epilogue), _vp, C.c_int64, _vp]),
    "ifx_gemm_q8_workspace_bytes": (C.c_int64, [_i32, _i32, _i32]),
    "ifx_attn_fwd_paged_ld": (C.c_int, [_vp, _i32, _vp, _i32, _vp, C.POINTER(KvView), _i32, _i32, _i32, _i32, _f32, _i32, _vp,
                                        C.c_int64, _vp]),
    "ifx_gemm_q8_quant_out": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(Epilogue), _vp, _i32, _vp]),
    "ifx_layernorm_quant_static": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _i32, _vp]),
    "ifx_quant_static": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ifx_quant_per_tensor": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "ifx_magi_head_prep": (C.c_int, [C.POINTER(MagiHeadPrepDesc), _vp]),
    "ifx_magi_gate_norm_residual": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _f32,
                                              _vp]),
    "ifx_act_rows": (C.c_int, [_vp, _vp, C.c_int64, _i32, _vp]),
    "ifx_kv_split_rows": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ifx_attn_fwd_dedup": (C.c_int, [_vp, _vp, C.POINTER(KvView), _i32, _i32, _i32, _i32, _f32, _vp]),
    "ifx_attn_fwd_ranges": (C.c_int, [_vp, _i32, _vp, _i32, C.POINTER(KvView), _i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32), _f32, _vp]),
    "ifx_gemm_workspace_bytes": (C.c_int64, [_i32, _i32, _i32]),
    "ifx_gemm_bf16_ws": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.POINTER(Epilogue), _vp, C.c_int64, _vp]),
}

_lib: Optional[C.CDLL] = None
ABI_MINOR = 7      # = IFX_ABI_MINOR of include/inferix_hip.h (checked against the header in tests/test_cabi_and_host.py)


def load() -> C.CDLL:
    """dlopen the in-tree library (built by `__graft_entry__.build()` / `make -C inferix_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  inferix_amd has no CPU or eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = (lib.ifx_version() >> 8) & 255
    if got != ABI_MINOR:       # a stale .so next to newer bindings (or the reverse): the argument structs would not line up
        raise HipLibraryMissing(f"{LIB_PATH} has ABI minor {got}, these bindings are for {ABI_MINOR} (include/inferix_hip.h "
                                "IFX_ABI_MINOR): rebuild with __graft_entry__.build()")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ifx_last_error().decode("utf-8", "replace")
        raise HipKernelError(f"{what} failed (code {rc}): {msg}")


def version() -> str:
    v = load().ifx_version()
    return f"{v >> 16}.{(v >> 8) & 255}.{v & 255}"
