"""Import-path shim for the two DAX names the reference's quantization examples import
(example/quantization/run_self_forcing_quantized.py:19-23): `dax.quant.quantization.quantize_dynamic` and the qconfig
factories.  DAX itself (github.com/RiseAI-Sys/DAX) is not vendored by the reference and not available here: these names resolve to
this build's own dynamic 8-bit linears (inferix_amd/quant.py, parity with DAX unpinned)."""
