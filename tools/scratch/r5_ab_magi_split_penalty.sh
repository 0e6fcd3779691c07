set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2 3; do
  for base in 0.00192 0.01; do
    IFX_ATTN_SPLIT_PENALTY=$base python bench.py --magi-leg fp8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base=$base magi ms_clip_rank', d['ms_clip_rank'])"
  done
done
