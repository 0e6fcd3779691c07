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


def test_streaming_eviction_at_full_size_page_rotation_equals_shift():
    """The product use of the path at its real size (round-5 verdict, item 5): a stream with `local_attn_size = 21` frames (a
    32760-token cache) and `sink_size = 3` long enough that the cache saturates and blocks 8, 9 and 10 each evict one block's rows
    behind the sink (causal_model.py:278-300) — 4680 tokens per block at the real channel geometry (dim 1536, 12 heads, ffn 8960),
    two layers, one denoise step + the clean-context re-run per block.  The two eviction forms must be the same function:
      * page-table rotation (one-frame pages: no row moves; the attention reads through the table) and
      * the shift kernel (`ifx_kv_roll` on a contiguous cache)
    give BIT-IDENTICAL latents, and the logical cache of every layer — the paged one read through its table — is bit-identical to
    the shifted one, at all 32760 keys.  (Tiny-size rollouts pin both forms to the reference's integer trace and cache contents:
    tests/test_hip_model.py; this is the 32760-key form of that statement.)"""
    from types import SimpleNamespace
    import wan_oracle as O
    from inferix_amd.core import DecodeMode
    from inferix_amd.kvcache_manager import KVCacheManager, KVCacheRequest
    from inferix_amd.pipeline import CausalInferencePipeline
    from inferix_amd.wan import HipCausalWanModel, HipWanDiffusionWrapper
    cfg = O.WanConfig(num_layers=2, local_attn_size=21, sink_size=3)
    W = O.init_weights(cfg, seed=4)
    blocks, fs = 10, cfg.frame_seqlen
    g = torch.Generator().manual_seed(21)
    noise = torch.randn(1, 3 * blocks, 16, cfg.latent_h, cfg.latent_w, generator=g).to(BF).cuda()
    pe = torch.zeros(1, cfg.text_len, cfg.text_dim)
    pe[:, :20] = torch.randn(1, 20, cfg.text_dim, generator=g)
    pe = pe.to(BF).cuda()
    results = {}
    for form in ("paged", "shift"):
        m = HipCausalWanModel(patch_size=cfg.patch_size, text_len=cfg.text_len, in_dim=cfg.in_dim, dim=cfg.dim, ffn_dim=cfg.ffn_dim,
                              freq_dim=cfg.freq_dim, text_dim=cfg.text_dim, out_dim=cfg.out_dim, num_heads=cfg.num_heads,
                              num_layers=cfg.num_layers, eps=cfg.eps, local_attn_size=21, sink_size=3, device="cuda")
        m.load_state_dict(W)
        gen = HipWanDiffusionWrapper(model=m, timestep_shift=5.0)
        args = SimpleNamespace(denoising_step_list=[1000], warp_denoising_step=True, num_frame_per_block=3, independent_first_frame=False,
                               context_noise=0, model_kwargs={}, frame_seq_length=fs, kv_cache_tokens=None)
        pipe = CausalInferencePipeline(args, "cuda", generator=gen, text_encoder=lambda text_prompts: {"prompt_embeds": pe}, vae=None)
        kvm, reqs = KVCacheManager("cuda"), [KVCacheRequest("s")]
        pipe._initialize_kv_cache(kvm, reqs, BF)
        trace = []
        m.index_trace = trace
        if form == "paged":
            for l in range(cfg.num_layers):
                kvm.enable_paging(reqs[0], f"layer_{l}", fs)
        out = pipe.inference(noise=noise, text_prompts=["x"], kv_cache_manager=kvm, kv_cache_requests=reqs,
                             decode_mode=DecodeMode.NO_DECODE, free_cache_before_vae=False)
        torch.cuda.synchronize()
        caches = []
        for l in range(cfg.num_layers):
            raw = kvm.get_raw(reqs[0], f"layer_{l}")
            k_log, v_log = raw[0, :, 0], raw[1, :, 0]
            pt = kvm.page_table(reqs[0], f"layer_{l}")
            if pt is not None:
                t = torch.arange(21 * fs)
                slot = (pt.host[t // pt.page_size].long() * pt.page_size + t % pt.page_size).cuda()
                assert sorted(pt.host.tolist()) == list(range(21)), "the table must stay a permutation of the physical pages"
                assert pt.host.tolist() != list(range(21)), "three evictions must have rotated the table"
                k_log, v_log = k_log[slot], v_log[slot]
            caches.append((k_log.clone(), v_log.clone()))
        results[form] = (out.clone(), caches, list(trace))
        kvm.free(reqs[0])
        del m, gen, pipe, kvm
        torch.cuda.empty_cache()
    (o_p, c_p, t_p), (o_s, c_s, t_s) = results["paged"], results["shift"]
    assert t_p == t_s and len(t_p) == 2 * blocks, "the integer index trace does not depend on the eviction form"
    # (global_end, local_end) of layer 0 after every forward: saturated at 32760 from block 7 on, the stream position keeps counting
    assert t_p[-1][2] == 21 * fs and t_p[-1][1] == 3 * blocks * fs, t_p[-1]
    assert torch.isfinite(o_p.float()).all()
    assert torch.equal(o_p.view(torch.int16), o_s.view(torch.int16)), \
        f"latents differ between page-table rotation and the shift kernel: rel L2 {rel_l2(o_p.cpu(), o_s.cpu()):.3e}"
    for l, ((kp, vp), (ks, vs)) in enumerate(zip(c_p, c_s)):
        assert torch.equal(kp.view(torch.int16), ks.view(torch.int16)), f"layer {l}: logical K through the page table != shifted cache"
        assert torch.equal(vp.view(torch.int16), vs.view(torch.int16)), f"layer {l}: logical V through the page table != shifted cache"
