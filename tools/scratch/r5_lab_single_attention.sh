set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs --sp-exchange peer"
for P in 8 4; do
 for rep in 1 2; do
  for single in 0 1; do
    echo "sp$P single=$single" >> $OUT/r5h_single.log
    IFX_SP_SINGLE_ATTN=$single $B --emulate-sp $P 2>> $OUT/r5h.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms', d['ms_per_step'])" >> $OUT/r5h_single.log
  done
 done
done
cat $OUT/r5h_single.log
