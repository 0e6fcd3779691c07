from inferix_amd.quant import quantize_dynamic  # noqa: F401
