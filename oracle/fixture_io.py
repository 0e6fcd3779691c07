"""TEST INFRASTRUCTURE — .npz fixture I/O with bf16 support (numpy has no bfloat16:
bf16 tensors are stored as their uint16 bit patterns under the key `<name>::bf16`,
complex128 as-is, everything else as its numpy dtype)."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def save_npz(path: str, tensors: Dict[str, object]) -> None:
    out = {}
    for k, v in tensors.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().contiguous()
            if v.dtype == torch.bfloat16:
                out[k + "::bf16"] = v.view(torch.int16).numpy().view(np.uint16)
            else:
                out[k] = v.numpy()
        else:
            out[k] = np.asarray(v)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **out)


def load_npz(path: str) -> Dict[str, torch.Tensor]:
    out = {}
    with np.load(path, allow_pickle=False) as z:
        for k in z.files:
            a = z[k]
            if k.endswith("::bf16"):
                out[k[:-6]] = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
            elif a.dtype.kind in "US":
                out[k] = a
            else:
                out[k] = torch.from_numpy(a.copy())
    return out


def golden(name: str) -> Dict[str, torch.Tensor]:
    return load_npz(os.path.join(GOLDEN_DIR, name))


def weights_checksum(W: Dict[str, torch.Tensor]) -> int:
    """Order-independent integer checksum of a bf16 weight dict (detects RNG drift)."""
    s = 0
    for k in sorted(W):
        s = (s * 1000003 + int(W[k].contiguous().view(torch.int16).to(torch.int64).sum().item())) % (1 << 61)
    return s
