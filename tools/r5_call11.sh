set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -q -m gpu -x -k "second_destination or block_full or 720p_full or rollout_tiny or rope or append or other_resolutions" > $OUT/r5i_tests.log 2>&1
echo "rc=$?" >> $OUT/r5i_tests.log
tail -n 6 $OUT/r5i_tests.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs"
for rep in 1 2; do
 for v in 0 1; do
  echo "n1 v_direct=$v" >> $OUT/r5i_vd.log
  IFX_V_DIRECT=$v $B 2>> $OUT/r5i.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms', d['ms_per_step'])" >> $OUT/r5i_vd.log
 done
done
cat $OUT/r5i_vd.log
