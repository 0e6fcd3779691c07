"""GPU parity of the umT5 text-encoder path: `ifx_t5_attention`, `ifx_t5_gated_gelu` per op and
`HipWanTextEncoder` end to end against the CPU oracle and the reference-generated golden context."""
import pytest
import torch

import t5_oracle as T
from fixture_io import golden, weights_checksum
from util import assert_bf16_parity, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from inferix_amd import hip_ops
    return hip_ops


def rnd(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(BF)


@pytest.mark.parametrize("B,L,heads,lens", [(1, 64, 2, [64]), (2, 192, 4, [150, 37]), (1, 512, 3, [301]), (3, 96, 1, [1, 96, 50])])
def test_t5_attention_against_oracle_math(ops, B, L, heads, lens):
    from inferix_amd.t5 import relative_position_table
    g = torch.Generator().manual_seed(L + heads)
    da = heads * 64
    qkv = rnd(g, B * L, 3 * da, scale=0.6)
    emb = rnd(g, 32, heads, scale=0.5)
    mask = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    # oracle: the middle of t5_oracle.attention (between the q/k/v and o linears)
    q, k, v = [t.view(B, L, heads, 64) for t in qkv.split(da, dim=1)]
    bias = q.new_zeros(B, heads, L, L)
    bias += T.position_bias(emb, L, L, 32, 128)
    bias.masked_fill_(mask.view(B, 1, 1, -1) == 0, torch.finfo(BF).min)
    attn = torch.einsum("binc,bjnc->bnij", q, k) + bias
    attn = torch.softmax(attn.float(), dim=-1).type_as(attn)
    ref = torch.einsum("bnij,bjnc->binc", attn, v).reshape(B * L, da)
    qkv_d = qkv.cuda()
    got = ops.t5_attention(qkv_d[:, :da], qkv_d[:, da:2 * da], qkv_d[:, 2 * da:], relative_position_table(emb, L, 32).cuda(),
                           mask.sum(1).to(torch.int32).cuda(), B, heads)
    # Scores are rounded to bf16 BEFORE the softmax on both sides (s = bf16(bf16(q.k) + bias)): where the fp32 accumulation
    # order flips that rounding, s moves by one bf16 ulp (3 % of exp(s) at |s| ~ 4), and a row whose softmax is dominated by
    # that key moves with it.  So: almost every element identical (<= 2 % may differ at all), tight tensor error, and a loose
    # per-element cap for the rare flipped rows.
    assert_bf16_parity(got.cpu(), ref, max_ulp=8, max_mismatch_frac=0.02, rel=2e-3, floor=1.0, what=f"t5 attention L{L}")


def test_t5_gated_gelu_is_the_reference_op_chain(ops):
    g = torch.Generator().manual_seed(1)
    gf = rnd(g, 70, 2 * 256, scale=2.0)
    ref = gf[:, 256:] * T.gelu(gf[:, :256])
    got = ops.t5_gated_gelu(gf.cuda())
    # 1 + tanh(.) cancels in the negative tail: a one-ulp difference between the device tanhf and the host one is carried at the
    # scale of the operands (util.assert_bf16_parity: ops with a cancellation stage are compared with floor=1)
    assert_bf16_parity(got.cpu(), ref, max_ulp=2, max_mismatch_frac=0.02, floor=1.0, what="gated gelu")


T5_MINE, T5_REF = 9.974e-2, 2.201e-2       # rel-L2 of the tiny golden encoder measured on MI355X in round 3 (gpurun_out/r3_par.log): HIP vs fp32, HIP vs reference


@pytest.fixture(scope="module")
def tiny():
    g = golden("t5_encoder.npz")
    v, d, da, f, h, n = [int(i) for i in g["cfg"]]
    cfg = T.T5Config(vocab_size=v, dim=d, dim_attn=da, dim_ffn=f, num_heads=h, num_layers=n)
    W = T.make_params(cfg, int(g["seed"]))
    assert weights_checksum(W) == int(g["weights_checksum"])
    from inferix_amd.t5 import HipWanTextEncoder
    enc = HipWanTextEncoder(W, tokenizer=lambda texts, return_mask=True, add_special_tokens=True: (g["ids"], g["mask"]),
                            dim=d, dim_attn=da, dim_ffn=f, num_heads=h, num_layers=n)
    return g, cfg, W, enc


def test_encoder_matches_reference_golden(tiny):
    g, cfg, W, enc = tiny
    got = enc(["prompt a", "prompt b"])["prompt_embeds"].cpu()
    ref = g["context"]
    assert got.shape == ref.shape and got.dtype == BF
    lens = g["mask"].sum(1).tolist()
    assert all(float(got[b, n:].abs().max()) == 0.0 for b, n in enumerate(lens))        # padding rows zeroed (wrapper.py:54-55)
    # exact answer = the fp32 evaluation of the same encoder; the reference's bf16 run sits `floor` away from it
    W32 = {k: v.float() for k, v in W.items()}
    exact = T.text_encoder_forward(cfg, W32, g["ids"], g["mask"])
    floor, mine, r = rel_l2(ref, exact), rel_l2(got, exact), rel_l2(got, ref)
    print(f"t5: bf16 noise floor (reference vs fp32) {floor:.3e}; HIP vs fp32 {mine:.3e}; HIP vs reference {r:.3e}")
    # `floor` is large here (0.099: every bf16 activation rounding of a random-weight encoder shows up against the fp32 evaluation) and
    # the HIP encoder sits AT it (0.0997) — but only 0.022 from the reference, whose roundings it mostly shares.  floor x 2 for that
    # second distance would let a 5x regression through (round-2 verdict), so both distances are also held to 1.5 x what this build
    # measured on MI355X in round 3 (T5_MINE, T5_REF above).
    assert mine <= min(1.25 * floor, 1.5 * T5_MINE) and r <= min(2.0 * floor, 1.5 * T5_REF), (floor, mine, r)


def test_encoder_is_deterministic_and_batch_invariant(tiny):
    g, cfg, W, enc = tiny
    a = enc.encode_ids(g["ids"], g["mask"])["prompt_embeds"]
    b = enc.encode_ids(g["ids"], g["mask"])["prompt_embeds"]
    assert torch.equal(a, b)
    one = enc.encode_ids(g["ids"][1:], g["mask"][1:])["prompt_embeds"]
    assert torch.equal(one[0], a[1]), "a prompt's embedding must not depend on what else is in the batch"


def test_encoder_errors(tiny):
    g, cfg, W, enc = tiny
    from inferix_amd.t5 import HipT5Encoder, HipWanTextEncoder
    with pytest.raises(ValueError, match="multiple of 32"):
        enc.encode_ids(g["ids"][:, :100], g["mask"][:, :100])
    with pytest.raises(NotImplementedError, match="head_dim 64"):
        HipT5Encoder(W, dim=cfg.dim, dim_attn=cfg.dim_attn, dim_ffn=cfg.dim_ffn, num_heads=8, num_layers=cfg.num_layers)
    with pytest.raises(RuntimeError, match="tokenizer"):
        HipWanTextEncoder(W, None, dim=cfg.dim, dim_attn=cfg.dim_attn, dim_ffn=cfg.dim_ffn, num_heads=cfg.num_heads,
                          num_layers=cfg.num_layers)(["x"])


def test_pipeline_takes_the_hip_text_encoder(tiny, tmp_path):
    """The pipelines' `text_encoder=` seam with the real encoder: `text_encoder(text_prompts=[...])["prompt_embeds"]` feeds the
    generator's cross-attention (the tiny DiT here reads 256-wide text features of 192 tokens)."""
    import yaml
    import wan_oracle as O
    from inferix_amd.core import DecodeMode
    from inferix_amd.pipeline import SelfForcingPipeline
    g, tcfg, TW, enc = tiny
    cfg = O.tiny_config(text_len=192, text_dim=tcfg.dim)
    conf = dict(denoising_step_list=[1000, 500], warp_denoising_step=True, num_frame_per_block=3, independent_first_frame=False,
                context_noise=0, timestep_shift=5.0, kv_cache_tokens=21 * cfg.frame_seqlen,
                latent_shape=[cfg.in_dim, cfg.latent_h, cfg.latent_w],
                model_kwargs=dict(patch_size=list(cfg.patch_size), text_len=cfg.text_len, in_dim=cfg.in_dim, dim=cfg.dim,
                                  ffn_dim=cfg.ffn_dim, freq_dim=cfg.freq_dim, text_dim=cfg.text_dim, out_dim=cfg.out_dim,
                                  num_heads=cfg.num_heads, num_layers=cfg.num_layers, eps=cfg.eps))
    path = tmp_path / "sf.yaml"
    path.write_text(yaml.safe_dump(conf))
    one = type(enc)(TW, tokenizer=lambda texts, return_mask=True, add_special_tokens=True: (g["ids"][:len(texts)], g["mask"][:len(texts)]),
                    dim=tcfg.dim, dim_attn=tcfg.dim_attn, dim_ffn=tcfg.dim_ffn, num_heads=tcfg.num_heads, num_layers=tcfg.num_layers)
    pipe = SelfForcingPipeline(str(path), text_encoder=one, vae=None)
    W = O.init_weights(cfg, seed=0)
    ck = tmp_path / "ckpt.pt"
    torch.save({"generator": {"model." + k: v for k, v in W.items()}}, ck)
    pipe.load_checkpoint(str(ck), use_ema=False)
    pipe.setup_devices(low_memory=False, verbose=False)
    torch.manual_seed(3)
    _, lat = pipe._run_inference(["a prompt"], 3, 1, decode_mode=DecodeMode.NO_DECODE, return_latents=True)
    assert lat.shape == (1, 3, cfg.in_dim, cfg.latent_h, cfg.latent_w) and torch.isfinite(lat.float()).all()
    # the same latents when the oracle's context is injected instead
    ctx = T.text_encoder_forward(tcfg, TW, g["ids"][:1], g["mask"][:1])
    pipe2 = SelfForcingPipeline(str(path), text_encoder=lambda text_prompts: {"prompt_embeds": ctx.cuda()}, vae=None)
    pipe2.load_checkpoint(str(ck), use_ema=False)
    pipe2.setup_devices(low_memory=False, verbose=False)
    torch.manual_seed(3)
    _, lat2 = pipe2._run_inference(["a prompt"], 3, 1, decode_mode=DecodeMode.NO_DECODE, return_latents=True)
    assert rel_l2(lat.float().cpu(), lat2.float().cpu()) < 2e-2


def test_full_size_umt5_properties():
    """umT5-XXL geometry (24 layers, dim 4096, 64 heads, ffn 10240; weights generated on the device): the CPU oracle is out of
    reach here, so size-independent properties — finite, padding rows zeroed, what sits in the PADDED positions of `ids` cannot
    change the valid rows (key-padding mask), a prompt does not depend on its batch neighbours, run-to-run bit determinism."""
    from inferix_amd.t5 import HipWanTextEncoder, synthetic_t5_state_dict
    enc = HipWanTextEncoder(synthetic_t5_state_dict(device="cuda", seed=5), None)
    g = torch.Generator().manual_seed(2)
    L, lens = 512, [77, 300]
    ids = torch.randint(1, 256384, (2, L), generator=g)
    mask = torch.zeros(2, L, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        ids[b, n:] = 0
    a = enc.encode_ids(ids, mask)["prompt_embeds"]
    assert a.shape == (2, L, 4096) and a.dtype == BF and torch.isfinite(a.float()).all()
    assert all(float(a[b, n:].abs().max()) == 0.0 for b, n in enumerate(lens))
    assert float(a[0, :77].float().std()) > 0.1
    ids2 = ids.clone()
    ids2[0, 77:] = torch.randint(1, 256384, (L - 77,), generator=g)          # garbage behind the mask
    b2 = enc.encode_ids(ids2, mask)["prompt_embeds"]
    assert torch.equal(a[0, :77], b2[0, :77]) and torch.equal(a[1], b2[1])
    one = enc.encode_ids(ids[1:], mask[1:])["prompt_embeds"]
    assert torch.equal(one[0], a[1])
    assert torch.equal(a, enc.encode_ids(ids, mask)["prompt_embeds"])
    del enc
    torch.cuda.empty_cache()
