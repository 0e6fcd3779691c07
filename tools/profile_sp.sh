# Kernel traces of ONE emulated sequence-parallel rank (bench.py --emulate-sp P --sp-exchange peer) on the GPU box: gpurun_out/r5_sp<P>_emulated_kernel_stats.md
# usage (via gpurun): SP_DEGREES="2 4 8" bash tools/profile_sp.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for P in ${SP_DEGREES:-2 4 8}; do
  rocprofv3 --kernel-trace --stats -d $OUT/sp${P}_trace -o bench -- python $R/bench.py --emulate-sp $P --sp-exchange peer --steps 2 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs > $OUT/sp${P}_bench.json 2> $OUT/sp${P}.err
  DB=$(ls $OUT/sp${P}_trace/*/*.db $OUT/sp${P}_trace/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $OUT/r5_sp${P}_emulated_kernel_stats.md
  rm -rf $OUT/sp${P}_trace
done
