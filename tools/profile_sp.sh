set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for P in 2 4; do
  rocprofv3 --kernel-trace --stats -d $OUT/sp${P}_trace -o bench -- python $R/bench.py --emulate-sp $P --sp-exchange peer --steps 2 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs > $OUT/sp${P}_bench.json 2> $OUT/sp${P}.err
  DB=$(ls $OUT/sp${P}_trace/*/*.db $OUT/sp${P}_trace/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $OUT/r4_sp${P}_kernel_stats.md
  rm -rf $OUT/sp${P}_trace
done
