#!/bin/bash
# Build ablated variants of the ping-pong attention kernel (compile-time PP_ABLATE) into tools/bin/abl/ and time them.
# Run the build part locally (hipcc cross-compiles), the timing part on the GPU box.
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/tools/bin/abl
if [ "$1" = "build" ]; then
  make -C $R/inferix_amd/csrc -j8 >/dev/null      # the other objects are linked as built
  for a in ${ABL_LIST:-0 1 2 4 8 3 6 10 12 14 15}; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${ABL_FLAGS:-} -DPP_ABLATE=$a -c $R/inferix_amd/csrc/ifx_attn_pp.hip -o $R/tools/bin/abl/pp${ABL_TAG:-}_$a.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $R/inferix_amd/csrc/build/*.o | grep -v ifx_attn_pp.o) $R/tools/bin/abl/pp${ABL_TAG:-}_$a.o -o $R/tools/bin/abl/lib${ABL_TAG:-}_$a.so ) &
  done; wait; ls $R/tools/bin/abl
else
  for a in ${ABL_LIST:-0 1 2 4 8 3 6 10 12 14 15}; do
    echo -n "PP_ABLATE=$a  "; ATTN_VARIANT=${ATTN_VARIANT:-2} IFX_HIP_LIB=$R/tools/bin/abl/lib${ABL_TAG:-}_$a.so python $R/tools/bench_kernels.py attn 2>&1 | grep "L= 32760"
  done
fi
