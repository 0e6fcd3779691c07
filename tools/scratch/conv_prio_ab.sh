set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/inferix_amd/csrc
OBJS=$(ls build/*.o | grep -v "ifx_conv")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DIFX_CONVPP_PRIO -c ifx_conv.hip -o /tmp/ifx_conv_prio.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/ifx_conv_prio.o -o /tmp/libinferix_hip_prio.so
cd $R
for i in 1 2; do
echo "== default"; python tools/bench_conv.py --variants 0 2>&1 | grep -v amdgpu.ids | tail -6
echo "== prio";    IFX_HIP_LIB=/tmp/libinferix_hip_prio.so python tools/bench_conv.py --variants 0 2>&1 | grep -v amdgpu.ids | tail -6
done
