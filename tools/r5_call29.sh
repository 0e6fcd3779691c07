set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_hip_magi.py tests/test_hip_magi_block.py tests/test_hip_magi_model.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2 3; do
python bench.py --magi-leg fp8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('magi ms_clip_rank', d['ms_clip_rank'])"
done
