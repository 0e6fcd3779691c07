# lab: tools/bench_conv.py with two builds of the library, alternating processes on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
  for lib in libinferix_hip_prev.so libinferix_hip.so; do
    echo "== $lib"
    IFX_HIP_LIB=$R/inferix_amd/$lib python tools/bench_conv.py --shapes main --variants 0 2>&1 | grep -E "^k"
  done
done
