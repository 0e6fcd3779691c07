#!/bin/bash
# The tools/gemm_lab invocations behind profiles/r3_gemm_pp.md (run on the GPU box from the repository root; build the lab first:
# g++ -O2 -o tools/bin/gemm_lab tools/gemm_lab.cpp -I/opt/rocm/include -L/opt/rocm/lib -lamdhip64 -ldl -D__HIP_PLATFORM_AMD__).
# usage: tools/gemm_lab_runs.sh block|long|shard|streamk|trace
case "${1:-block}" in
  block)   # the six block GEMMs at 4680 and 9360 rows: r2 auto tiles (19 = four-wave, 5 = 256x256) against the ping-pong tiles
    timeout 300 tools/bin/gemm_lab -r 7 0,22,23,24 4680,4608,1536,0 4680,1536,1536,3 4680,1536,1536,0 4680,1536,1536,2 4680,8960,1536,1 4680,1536,8960,3
    timeout 300 tools/bin/gemm_lab -r 5 0,22,23,24 9360,4608,1536,0 9360,8960,1536,1 9360,1536,8960,3 9360,1536,1536,3 ;;
  long)    # MAGI's long-K shapes and the 2340-row shard shapes
    timeout 300 tools/bin/gemm_lab -r 5 19,22,23,5 6075,8192,3072,0 6075,12288,3072,1 6075,3072,12288,0 2340,4608,1536,0 2340,8960,1536,1 2340,1536,8960,3 2340,1536,1536,3 ;;
  shard)   # 585 / 1170 rows with the in-workgroup split tiles (what a sequence-parallel rank runs)
    IFX_GEMM_SMALL_SPLIT=1 timeout 300 tools/bin/gemm_lab -r 7 0 585,4608,1536,0 585,1536,1536,3 585,8960,1536,1 585,1536,8960,3 1170,4608,1536,0 1170,8960,1536,1 ;;
  streamk) # stream-K (variant 26) against the auto choice at shard sizes
    IFX_GEMM_SMALL_SPLIT=1 timeout 300 tools/bin/gemm_lab -r 7 0,26 585,4608,1536,0 585,1536,1536,3 585,1536,1536,0 585,8960,1536,1 585,1536,8960,3 1170,4608,1536,0 1170,1536,8960,3 ;;
  trace)   # segment sums of workgroup 0 (make -C inferix_amd/csrc trace first)
    for s in 4680,1536,8960,3 4680,1536,1536,3 6075,8192,3072,0; do
      timeout 120 tools/bin/gemm_lab -l inferix_amd/libinferix_hip_trace.so -t -r 3 0 $s; done ;;
esac
