"""Comparison helpers shared by the parity tests."""
import torch


def bf16_ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 ULPs (monotone integer mapping of the bit patterns)."""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(i >= 0x8000, 0x8000 - i, i)
    return (key(a.to(torch.bfloat16)) - key(b.to(torch.bfloat16))).abs()


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def ulp_report(got: torch.Tensor, ref: torch.Tensor) -> str:
    """Mismatch fraction and histogram of the bf16 ULP distances between two tensors (raw bit-pattern distance, no magnitude floor):
    'differ 0.3121 | ulp 0: 68.79% 1: 30.02% 2: 1.13% 3: 0.05% 4: 0.01% >4: 0.00% | max 5' — printed by the full-size parity tests
    on success and carried by every assert_bf16_parity failure, so that drift inside a generous bound is visible."""
    d = bf16_ulp_diff(got.detach().cpu(), ref.detach().cpu()).flatten()
    n = max(1, d.numel())
    counts = [int((d == k).sum()) for k in range(5)] + [int((d > 4).sum())]
    hist = " ".join(f"{'>4' if k == 5 else k}: {100.0 * c / n:.2f}%" for k, c in enumerate(counts))
    return f"differ {1.0 - counts[0] / n:.4f} | ulp {hist} | max {int(d.max()) if d.numel() else 0}"


def assert_bf16_parity(got: torch.Tensor, ref: torch.Tensor, *, max_ulp=1, max_mismatch_frac=0.02, rel=1e-3,
                       floor=0.05, scale=None, what="", report=False):
    """The parity bar for one fused op on identical inputs (the stated bf16 tolerance of north_star):
      * every element within `max_ulp` bf16 ULPs of the reference, where the ULP is taken at
        max(|ref|, floor * tensor RMS) — fp32 reduction-order noise can flip one bf16 rounding of an
        intermediate, and results of cancellations (x - mean, a*cos - b*sin, x + y*g, GELU tails) carry
        that flip at the scale of their operands, so ops with such a stage are compared with floor=1
        (one ULP of a typical element) and pure normalisations / dot products with floor=0.05;
        `scale` (optional, same shape) raises the per-element magnitude at which the ULP is taken to that of
        the op's OPERANDS (e.g. the modulus of the rotated pair for RoPE: a rotation preserves it, so a one-ULP
        flip of an operand shows up at that magnitude whatever the size of the individual output component);
      * at most `max_mismatch_frac` of the elements differ at all;
      * tensor-level relative L2 error <= `rel` (1e-3)."""
    got, ref = got.detach().cpu(), ref.detach().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got.float()).all(), f"{what}: non-finite output"
    g, r0 = got.double(), ref.double()
    rms = float(r0.pow(2).mean().sqrt())
    mag = r0.abs() if scale is None else torch.maximum(r0.abs(), scale.detach().cpu().double())
    tol = max_ulp * 2.0 ** -7 * torch.maximum(mag, torch.tensor(floor * rms, dtype=torch.float64))
    diff = (g - r0).abs()
    frac = float((diff > 0).double().mean())
    r = rel_l2(got, ref)
    worst = float((diff / tol).max())
    if report:
        print(f"{what}: rel L2 {r:.3e} (bound {rel:.3e}), worst element {worst:.2f}x the {max_ulp}-ulp bound; {ulp_report(got, ref)}")
    assert worst <= 1.0, f"{what}: element error {worst:.2f}x the {max_ulp}-ulp bound (frac {frac:.4f}, rel {r:.2e}); {ulp_report(got, ref)}"
    assert frac <= max_mismatch_frac, f"{what}: {frac:.4f} of elements differ (> {max_mismatch_frac}); {ulp_report(got, ref)}"
    assert r <= rel, f"{what}: rel L2 {r:.3e} > {rel}; {ulp_report(got, ref)}"
    return frac, r


def pair_modulus(t: torch.Tensor) -> torch.Tensor:
    """|(t[2i], t[2i+1])| broadcast back to both components (RoPE operand scale)."""
    p = t.double().unflatten(-1, (-1, 2))
    return p.norm(dim=-1, keepdim=True).expand_as(p).flatten(-2)
