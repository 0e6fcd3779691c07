L=inferix_amd/libinferix_hip_trace.so
timeout 300 tools/bin/gemm_lab -r 5 0,22,23,24 block
for sh in 4680,8960,1536,0 4680,1536,1536,3 4680,1536,8960,3; do echo "=== trace v22 $sh"; timeout 100 tools/bin/gemm_lab -l $L -t -r 3 22 $sh; done
echo "=== trace v24 4680,1536,1536,3"; timeout 100 tools/bin/gemm_lab -l $L -t -r 3 24 4680,1536,1536,3
