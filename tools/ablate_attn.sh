#!/bin/bash
# Build ablated variants of the ping-pong attention kernel (compile-time PP_ABLATE) into tools/bin/abl/ and time them.
# Run the build part locally (hipcc cross-compiles), the timing part on the GPU box.
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/tools/bin/abl
if [ "$1" = "build" ]; then
  for a in ${ABL_LIST:-0 1 2 4 8 3 6 10 12 14 15}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ${ABL_FLAGS:-} -DPP_ABLATE=$a $R/inferix_amd/csrc/ifx_core.hip $R/inferix_amd/csrc/ifx_norm.hip \
      $R/inferix_amd/csrc/ifx_gemm.hip $R/inferix_amd/csrc/ifx_gemm_glds.hip $R/inferix_amd/csrc/ifx_attn.hip $R/inferix_amd/csrc/ifx_attn_pp.hip \
      $R/inferix_amd/csrc/ifx_quant.hip $R/inferix_amd/csrc/ifx_conv.hip $R/inferix_amd/csrc/ifx_t5.hip -o $R/tools/bin/abl/lib${ABL_TAG:-}_$a.so &
  done; wait; ls $R/tools/bin/abl
else
  for a in ${ABL_LIST:-0 1 2 4 8 3 6 10 12 14 15}; do
    echo -n "PP_ABLATE=$a  "; ATTN_VARIANT=${ATTN_VARIANT:-2} IFX_HIP_LIB=$R/tools/bin/abl/lib${ABL_TAG:-}_$a.so python $R/tools/bench_kernels.py attn 2>&1 | grep "L= 32760"
  done
fi
