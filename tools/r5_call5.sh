set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
python -m pytest tests/test_hip_model.py tests/test_hip_plugin_api.py tests/test_pipeline_host_api.py -q -m gpu -x > $OUT/r5e_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r5e_tests.log
python -m pytest tests/test_hip_sequence_parallel.py tests/test_hip_quant.py -q -m gpu -x > $OUT/r5e_tests_sp.log 2>&1
echo "sp tests rc=$?" >> $OUT/r5e_tests_sp.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs --sp-exchange peer"
for rep in 1 2; do
  for P in 8 4 2; do
    echo "sp$P streams" >> $OUT/r5e_modes.log
    $B --emulate-sp $P --pair on --pair-mode streams 2>> $OUT/r5e.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms', d['ms_per_step'], 'mode', d['config']['pair_mode'])" >> $OUT/r5e_modes.log
  done
  echo "n1 streams" >> $OUT/r5e_modes.log
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs --pair on --pair-mode streams 2>> $OUT/r5e.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms', d['ms_per_step'], 'mode', d['config']['pair_mode'])" >> $OUT/r5e_modes.log
  echo "n1 off" >> $OUT/r5e_modes.log
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs --pair off 2>> $OUT/r5e.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms', d['ms_per_step'], 'mode', d['config']['pair_mode'])" >> $OUT/r5e_modes.log
done
tail -n 5 $OUT/r5e_tests.log $OUT/r5e_tests_sp.log
cat $OUT/r5e_modes.log
