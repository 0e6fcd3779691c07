"""Import-path shim: `inferix.*` -> the MI355X-native implementation in `inferix_amd`.

The drop-in boundary of this build is the reference's own Python plugin / operator API.  With this directory on
`sys.path` in place of the reference's package, the reference's example scripts (example/self_forcing/run_self_forcing.py,
example/causvid/run_causvid.py, example/quantization/run_self_forcing_quantized.py, example/streaming/*) import the HIP path
under the names they already use.  Only the modules of the hot path and of its callers exist here; every module is a
re-export (no logic), INTEGRATION.md lists the mapping.  Nothing here imports the reference."""
__version__ = "0.2.0+mi355x"
