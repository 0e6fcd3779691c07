"""GPU: size-independent properties at BASELINE's full sizes (Self-Forcing 480p: 4680 queries x 12 heads against the
32760-key prefix of the last block; the 4680 x 8960 x 1536 FFN GEMMs), where the CPU oracle would take minutes:
 * attention is a convex combination of the value rows (every output channel inside [min V, max V] of its head);
 * split-KV identity: attention over [0, L) == LSE-merge of [0, L/2) and [L/2, L), and == the split-KV launch;
 * permuting the cached keys (with their values) does not change the result beyond summation order;
 * the linear layers are linear: (x1 + x2) W^T + b == x1 W^T + x2 W^T + b within bf16 rounding;
 * re-running a block at the same `current_start` overwrites its KV slots with the same bytes (idempotence)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from util import rel_l2  # noqa: E402

BF = torch.bfloat16
N, H, D, L = 4680, 12, 128, 32760


@pytest.fixture(scope="module")
def ops():
    from inferix_amd import hip_ops
    return hip_ops


@pytest.fixture(scope="module")
def qkv():
    g = torch.Generator(device="cuda").manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g, device="cuda").to(BF)
    return r(N, H, D), r(L, H, D), r(L, H, D)


def test_attention_full_prefix_is_a_convex_combination(ops, qkv):
    q, k, v = qkv
    out, lse = ops.attention(q, ops.KvCacheView(k, v), L, return_lse=True)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    vmin, vmax = v.float().amin(0), v.float().amax(0)                    # [H, D]
    o = out.float()
    eps = 1e-2                                                           # one bf16 ulp at |v| ~ 4
    assert (o >= vmin[None] - eps).all() and (o <= vmax[None] + eps).all()
    # with ~33k random keys the softmax is diffuse: outputs are far inside the hull
    assert o.abs().max() < 1.0


def test_attention_full_prefix_split_identities(ops, qkv):
    q, k, v = qkv
    view = ops.KvCacheView(k, v)
    full, lse = ops.attention(q, view, L, return_lse=True, splits=1)
    half = (L // 2 // 64) * 64 + 17                                       # an unaligned cut
    o1, l1 = ops.attention(q, view, half, return_lse=True, splits=1)
    o2, l2 = ops.attention(q, view, L, return_lse=True, splits=1, kv_start=half)
    ops.lse_merge(o1, l1, o2, l2)
    assert rel_l2(o1.cpu(), full.cpu()) < 4e-3 and (l1 - lse).abs().max() < 1e-3
    for s in (2, 5):
        sp, lsp = ops.attention(q, view, L, return_lse=True, splits=s)
        assert rel_l2(sp.cpu(), full.cpu()) < 3e-3 and (lsp - lse).abs().max() < 1e-3


def test_attention_full_prefix_key_permutation(ops, qkv):
    q, k, v = qkv
    full = ops.attention(q, ops.KvCacheView(k, v), L)
    perm = torch.randperm(L, generator=torch.Generator().manual_seed(3)).cuda()
    permuted = ops.attention(q, ops.KvCacheView(k[perm].contiguous(), v[perm].contiguous()), L)
    # a different key order changes which bf16 roundings of P meet in which tile: with a diffuse softmax over 32760
    # random keys the outputs are small averages (|o| ~ 0.01) and two orderings differ by the rounding noise itself
    assert rel_l2(permuted.cpu(), full.cpu()) < 1e-2
    assert (permuted.float() - full.float()).abs().max() < 2e-3


def test_ffn_gemms_are_linear_at_full_size(ops):
    from inferix_amd import _hip
    g = torch.Generator(device="cuda").manual_seed(12)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device="cuda") * sc).to(BF)
    d, f = 1536, 8960
    x1, x2 = r(N, d), r(N, d)
    xs = (x1.float() + x2.float()).to(BF)
    w1, b1 = r(f, d, sc=d ** -0.5), r(f, sc=0.1)
    y1, y2, ys = ops.linear(x1, w1, b1), ops.linear(x2, w1, b1), ops.linear(xs, w1, b1)
    ref = y1.float() + y2.float() - b1.float()
    assert rel_l2(ys.cpu(), ref.cpu()) < 6e-3                            # three bf16 roundings of O(1) values
    u1, u2 = r(N, f), r(N, f)
    us = (u1.float() + u2.float()).to(BF)
    w2, b2 = r(d, f, sc=f ** -0.5), r(d, sc=0.1)
    z1, z2, zs = ops.linear(u1, w2, b2), ops.linear(u2, w2, b2), ops.linear(us, w2, b2)
    assert rel_l2(zs.cpu(), (z1.float() + z2.float() - b2.float()).cpu()) < 6e-3
    # GELU epilogue == GELU of the plain epilogue's bf16 output
    h0 = ops.linear(x1, w1, b1)
    h1 = ops.linear(x1, w1, b1, epilogue=_hip.IFX_EPI_GELU_TANH)
    ref = torch.nn.functional.gelu(h0.float(), approximate="tanh").to(BF)
    diff = (h1.float() - ref.float()).abs()
    assert (diff <= 2.0 ** -7 * ref.float().abs().clamp_min(2.0 ** -6)).all()        # at most one bf16 ulp apart
    assert (h1 != ref).float().mean() < 0.02


def test_kv_append_is_idempotent_at_full_size(ops):
    from inferix_amd.wan import components as C
    g = torch.Generator(device="cuda").manual_seed(13)
    d = H * D
    qkv_rows = torch.randn(N, 3 * d, generator=g, device="cuda").to(BF)
    w = torch.randn(d, generator=g, device="cuda").to(BF)
    kc = torch.zeros(L, H, D, dtype=BF, device="cuda")
    vc = torch.zeros(L, H, D, dtype=BF, device="cuda")
    rope = ops.RopeGridSpec(C.rope_table(D).cuda(), 18, 30, 52, 0, N // 3)      # frames 18..20 = the last block
    start = L - N
    q1 = ops.rmsnorm_rope_kv_append(qkv_rows, w, w, 1e-6, rope, ops.KvCacheView(kc, vc), start, d)
    k1, v1 = kc.clone(), vc.clone()
    q2 = ops.rmsnorm_rope_kv_append(qkv_rows, w, w, 1e-6, rope, ops.KvCacheView(kc, vc), start, d)
    assert torch.equal(q1, q2) and torch.equal(kc, k1) and torch.equal(vc, v1)
    assert not kc[:start].any() and not vc[:start].any()                 # nothing outside the block's slots was touched
    assert torch.equal(vc[start:].reshape(N, d), qkv_rows[:, 2 * d:])    # V is stored raw
