set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_hip_kernels.py tests/test_hip_magi.py tests/test_hip_magi_block.py tests/test_hip_full_size_properties.py -q -m gpu -x -k "attention or split or merge or range or magi" 2>&1 | tail -3
for rep in 1 2; do
python bench.py --magi-leg fp8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('magi ms_clip_rank', d['ms_clip_rank'])"
done
for P in 8 4 2; do
  python bench.py --emulate-sp $P --sp-exchange peer --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sp$P ms', d['ms_per_step'])"
done
