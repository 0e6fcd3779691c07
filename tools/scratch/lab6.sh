export IFX_GEMM_SMALL_SPLIT=1
timeout 120 tools/bin/gemm_lab -l inferix_amd/libinferix_hip_trace.so -t -r 3 0 585,4608,1536,0
timeout 120 tools/bin/gemm_lab -l inferix_amd/libinferix_hip_trace.so -t -r 3 0 585,1536,8960,3
