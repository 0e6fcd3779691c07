#!/bin/bash
# Rasterisation-group sweep of the LDS-DMA GEMM kernels (compile-time IFX_GEMM_GM) into tools/bin/gm/; `build` locally, timing on the GPU box.
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/tools/bin/gm
SRC="ifx_core ifx_norm ifx_gemm ifx_gemm_glds ifx_attn ifx_attn_pp ifx_quant ifx_conv ifx_t5"
if [ "$1" = "build" ]; then
  for g in ${GM_LIST:-2 4 8 16}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DIFX_GEMM_GM=$g $(for s in $SRC; do echo $R/inferix_amd/csrc/$s.hip; done) -o $R/tools/bin/gm/lib_gm$g.so &
  done; wait; ls $R/tools/bin/gm
else
  for g in ${GM_LIST:-2 4 8 16}; do
    echo "GM=$g"; IFX_HIP_LIB=$R/tools/bin/gm/lib_gm$g.so python $R/tools/bench_gemm_tiles.py 4680 0,3,5 2>&1 | grep "^M=" | cut -c1-110
  done
fi
