L=inferix_amd/libinferix_hip_trace.so
for d in 0 8; do for s in "24 4680,1536,1536,0" "22 6075,8192,3072,0"; do set -- $s; echo "=== IFX_PP_DEBUG=$d v$1 $2"; IFX_PP_DEBUG=$d timeout 100 tools/bin/gemm_lab -l $L -t -r 5 $1 $2 | grep -v "^ *$"; done; done
