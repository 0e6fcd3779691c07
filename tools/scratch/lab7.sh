for i in 1 2; do
for L in inferix_amd/libinferix_hip.so inferix_amd/libinferix_hip_stag.so; do
echo "== $L"
timeout 120 tools/bin/gemm_lab -l $L -r 7 22,23,24 4680,4608,1536,0 6075,8192,3072,0 4680,1536,1536,0 4680,8960,1536,1 | grep -v "^ *$"
done
done
