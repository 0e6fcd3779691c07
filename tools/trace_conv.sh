#!/bin/bash
# s_memtime step trace of conv_pp_kernel + the ablations DESIGN 13 quotes -> gpurun_out/r6_conv_step_trace.log
# (builds lab libraries next to the product one; removes them afterwards).   usage (GPU box): bash tools/trace_conv.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/inferix_amd/csrc
OBJS=$(ls build/*.o | grep -v "ifx_conv")
for v in ${TRACE_VARIANTS:-"" "NORES" "NOSTORE" "NOFRAG" "NOFRAG -DIFX_CONVPP_NOMFMA"}; do
  n=$(echo "trace$v" | tr -d ' -' | sed 's/DIFX_CONVPP_//g')
  flags="-DIFX_CONVPP_TRACE=1"; [ -n "$v" ] && flags="$flags -DIFX_CONVPP_$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c ifx_conv.hip -o /tmp/ifx_conv_$n.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/ifx_conv_$n.o -o /tmp/libinferix_hip_$n.so
done
cd $R
for n in trace traceNORES traceNOSTORE traceNOFRAG traceNOFRAGNOMFMA; do
  echo "== $n  (96 -> 96 @480x832 x12 frames, planar ring, residual; second launch of two)"
  IFX_HIP_LIB=/tmp/libinferix_hip_$n.so python tools/scratch/conv_one.py 480 832 12 0 1 2>&1 | grep -v amdgpu.ids | tail -3
done
echo "== trace, 192 -> 192 @240x416 x12"
C=192 IFX_HIP_LIB=/tmp/libinferix_hip_trace.so python tools/scratch/conv_one.py 240 416 12 0 1 2>&1 | grep -v amdgpu.ids | tail -3
echo "== trace, 384 -> 384 @120x208 x6"
C=384 IFX_HIP_LIB=/tmp/libinferix_hip_trace.so python tools/scratch/conv_one.py 120 208 6 0 1 2>&1 | grep -v amdgpu.ids | tail -3
