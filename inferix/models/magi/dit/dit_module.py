"""inferix/models/magi/dit/dit_module.py: TransformerLayer :1201, TransformerBlock :1322, FullyParallelAttention :833,
the static-scale FP8 linears :434-490"""
from inferix_amd.magi.dit import HipFullyParallelAttention as FullyParallelAttention  # noqa: F401
from inferix_amd.magi.dit import HipMagiTransformerBlock as TransformerBlock  # noqa: F401
from inferix_amd.magi.dit import HipMagiTransformerLayer as TransformerLayer  # noqa: F401
from inferix_amd.quant import StaticFp8Linear  # noqa: F401
