set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
python -m pytest tests/test_hip_magi.py tests/test_hip_magi_block.py tests/test_hip_magi_model.py -q -m gpu -x > $OUT/r5o_magi_tests.log 2>&1; echo "rc=$?" >> $OUT/r5o_magi_tests.log
tail -n 5 $OUT/r5o_magi_tests.log
for fast in 0 1 0 1; do
  IFX_MAGI_ULYSSES_FAST=$fast python bench.py --magi-leg fp8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fast=$fast ms_clip_rank', d['ms_clip_rank'])"
done
