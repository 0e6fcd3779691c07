"""CPU: oracle/magi_block_oracle.py reproduces the reference's MAGI TransformerLayer fixtures bit for bit
(tests/golden/magi_block_{tiny,real}.npz, written by oracle/gen_golden_magi_block.py from the reference itself), and the
quantise-clamp-cast step reproduces the reference's `div_clamp_to` fixture."""
import pytest
import torch

import magi_block_oracle as MB
from fixture_io import golden


@pytest.mark.parametrize("name", ["magi_block_tiny", "magi_block_real"])
def test_oracle_reproduces_reference_layer(name):
    fx = golden(name + ".npz")
    cfg, n_layers, clip, n_calls, wseed, max_tokens = MB.fixture_geometry(fx)
    Ws = [MB.init_layer_weights(cfg, wseed + li) for li in range(n_layers)]
    caches = [MB.MagiLayerCache(max_tokens, cfg.num_query_groups, cfg.kv_channels) for _ in range(n_layers)]
    for ci in range(n_calls):
        inp, meta = MB.fixture_call(fx, ci)
        x = inp["x"]
        for li in range(n_layers):
            taps = {}
            x = MB.layer_forward(Ws[li], cfg, x, inp["condition"], inp["condition_map"], inp["y"], inp["rope"], meta,
                                 caches[li], taps)
            assert torch.equal(x, fx[f"c{ci}_out_l{li}"]), (name, ci, li)
            if li == 0:
                for t, v in taps.items():
                    if f"c{ci}_tap_{t}" in fx:
                        assert torch.equal(v, fx[f"c{ci}_tap_{t}"]), (name, ci, t)
    written = int(fx["cache_written"])
    for li in range(n_layers):
        assert torch.equal(caches[li].k[:written], fx[f"cache_l{li}"][0, :written, 0])
        assert torch.equal(caches[li].v[:written], fx[f"cache_l{li}"][1, :written, 0])


def test_exact_layer_is_close_to_the_bf16_layer():
    """The float64 evaluation (the yardstick of the GPU parity tests) agrees with the reference's bf16 output to bf16 noise."""
    fx = golden("magi_block_tiny.npz")
    cfg, n_layers, clip, n_calls, wseed, max_tokens = MB.fixture_geometry(fx)
    W = MB.init_layer_weights(cfg, wseed)
    inp, meta = MB.fixture_call(fx, 0)
    ex = MB.exact_layer_forward(W, cfg, inp["x"], inp["condition"], inp["condition_map"], inp["y"], inp["rope"], meta,
                                MB.MagiLayerCache(max_tokens, cfg.num_query_groups, cfg.kv_channels))
    ref = fx["c0_out_l0"].double()
    rel = float((ref - ex).norm() / ex.norm())
    assert rel < 1e-2, rel


def test_oracle_reproduces_reference_div_clamp_to_and_static_fp8_linears():
    fx = golden("quant_fp8.npz")
    x = fx["x"]
    K = x.shape[1]
    for name in ("vec", "one"):
        assert torch.equal(MB.div_clamp_to(x, fx[f"div_{name}"]).view(torch.uint8), fx[f"q_{name}"]), name
    assert int(fx["double_rounding_diffs"]) > 0        # the fixture does exercise the bf16 intermediate
    wq = fx["wq"].view(torch.float8_e4m3fn)
    xin = x.reshape(4, 24, K)
    ins = fx["in_scale"]
    assert torch.equal(MB.fp8_static_linear(xin, wq, fx["w_scale"], ins.expand(K), ins.expand(K)), fx["y_per_tensor"])
    assert torch.equal(MB.fp8_static_linear(xin, wq, fx["w_scale"], ins, fx["div_vec"].reshape(1, K)), fx["y_per_channel"])
