set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for pen in 0.01 0.02 0.03 0.05 0.005; do
  IFX_ATTN_SPLIT_PENALTY=$pen python bench.py --magi-leg fp8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pen=$pen magi ms_clip_rank', d['ms_clip_rank'])"
done
for pen in 0.01 0.03 0.05; do
  for P in 8 4; do
  IFX_ATTN_SPLIT_PENALTY=$pen python bench.py --emulate-sp $P --sp-exchange peer --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pen=$pen sp$P ms', d['ms_per_step'])"
  done
done
