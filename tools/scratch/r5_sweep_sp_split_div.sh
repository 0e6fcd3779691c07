set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs --sp-exchange peer --pair on"
for P in 8 4; do
  for div in 1,1 2,2 2,4 1,2 4,4 1,8; do
    echo "sp$P div $div" >> $OUT/r5c_split_div.log
    IFX_SP_SPLIT_DIV=$div $B --emulate-sp $P 2>> $OUT/r5c.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms', d['ms_per_step'], 'attn frac', d['roofline']['frac'], 'avg attn ms', d['roofline']['avg_launch_ms'])" >> $OUT/r5c_split_div.log
  done
done
cat $OUT/r5c_split_div.log
