# lab: the same attention timings with two builds of the library, alternating processes on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
  for lib in libinferix_hip_prev.so libinferix_hip.so; do
    echo "== $lib"
    IFX_HIP_LIB=$R/inferix_amd/$lib python tools/scratch/attn_paged_ab.py 2>&1 | grep -E "contiguous:|  1560:"
  done
done
