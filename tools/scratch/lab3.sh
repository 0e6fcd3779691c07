L=inferix_amd/libinferix_hip_trace.so
for d in 0 1 2 3 4 8; do echo "=== IFX_PP_DEBUG=$d"; IFX_PP_DEBUG=$d timeout 100 tools/bin/gemm_lab -l $L -t -r 5 22 4680,8960,1536,0; done
echo "=== gelu"; timeout 100 tools/bin/gemm_lab -l $L -t -r 5 22 4680,8960,1536,1
echo "=== down"; timeout 100 tools/bin/gemm_lab -l $L -t -r 5 22,24 4680,1536,8960,0
