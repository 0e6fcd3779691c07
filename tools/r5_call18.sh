set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs"
for rep in 1 2; do
  for P in 8 4 2; do
    echo "sp$P" >> $OUT/r5l.log
    $B --emulate-sp $P --sp-exchange peer 2>> $OUT/r5l.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms', d['ms_per_step'])" >> $OUT/r5l.log
  done
  echo "n1" >> $OUT/r5l.log
  $B 2>> $OUT/r5l.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms', d['ms_per_step'])" >> $OUT/r5l.log
  echo "n1 pair off" >> $OUT/r5l.log
  $B --pair off 2>> $OUT/r5l.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms', d['ms_per_step'])" >> $OUT/r5l.log
done
cat $OUT/r5l.log
