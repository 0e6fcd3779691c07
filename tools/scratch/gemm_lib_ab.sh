R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
  for lib in libinferix_hip_prev.so libinferix_hip.so; do
    echo "== $lib"
    IFX_HIP_LIB=$R/inferix_amd/$lib python tools/scratch/gemm_block_time.py 2>&1 | grep -v amdgpu.ids
  done
done
