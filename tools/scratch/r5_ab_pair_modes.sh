set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
python -m pytest tests/test_hip_model.py -q -m gpu -x -s -k "paired or rollout" > $OUT/r5d_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r5d_tests.log
python -m pytest tests/test_hip_sequence_parallel.py -q -m gpu -x -s > $OUT/r5d_tests_sp.log 2>&1
echo "sp tests rc=$?" >> $OUT/r5d_tests_sp.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs --sp-exchange peer --pair on"
for P in 8 4 2; do
  for mode in streams lockstep streams lockstep; do
    echo "sp$P $mode" >> $OUT/r5d_modes.log
    $B --emulate-sp $P --pair-mode $mode 2>> $OUT/r5d.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms', d['ms_per_step'], 'mode', d['config']['pair_mode'])" >> $OUT/r5d_modes.log
  done
done
for mode in streams lockstep; do
  echo "n1 $mode" >> $OUT/r5d_modes.log
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs --pair on --pair-mode $mode 2>> $OUT/r5d.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms', d['ms_per_step'], 'mode', d['config']['pair_mode'])" >> $OUT/r5d_modes.log
done
tail -n 5 $OUT/r5d_tests.log $OUT/r5d_tests_sp.log
cat $OUT/r5d_modes.log
tail -5 $OUT/r5d.err
