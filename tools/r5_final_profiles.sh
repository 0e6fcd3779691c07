# round 5: the profile set of the final tree — headline kernel trace + PMC passes, emulated sp2 / sp4 / sp8 rank traces, bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
bash tools/profile_bench.sh r5 > $OUT/r5_profile_bench.log 2>&1
SP_DEGREES="2 4 8" bash tools/profile_sp.sh > $OUT/r5_profile_sp.log 2>&1
cd $R
python bench.py > $OUT/r5_bench_line.json 2> $OUT/r5_bench_line.err
for P in 2 4 8; do
  python bench.py --emulate-sp $P --sp-exchange peer --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs > $OUT/r5_bench_sp$P.json 2>> $OUT/r5_bench_sp.err
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs --pair off > $OUT/r5_bench_pair_off.json 2>> $OUT/r5_bench_sp.err
ls -la $OUT | grep r5_ | tail -30
