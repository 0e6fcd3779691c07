# round 5, call 2: the sequence-parallel GPU tests again, the K split over 2 / 4 / 8 workgroups on the 128-token ping-pong tile at the
# shard sizes (gemm_variant 27 / 28 / 29 against the auto choice under gemm_small_split), and a kernel trace of the paired sp8 rank
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
python -m pytest tests/test_hip_sequence_parallel.py -q -m gpu -x -s > $OUT/r5b_tests_sp.log 2>&1
echo "sp tests rc=$?" >> $OUT/r5b_tests_sp.log
for M in 585 1170 2340; do
  IFX_SMALL_SPLIT=1 python tools/bench_gemm_tiles.py $M 0,27,28,29 >> $OUT/r5b_gemm_split.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/r5b_sp8_trace -o bench -- python $R/bench.py --emulate-sp 8 --sp-exchange peer --pair on --steps 2 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs > $OUT/r5b_sp8_bench.json 2> $OUT/r5b_sp8.err
DB=$(ls $OUT/r5b_sp8_trace/*/*.db $OUT/r5b_sp8_trace/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB > $OUT/r5b_sp8_pair_kernel_stats.md
rm -rf $OUT/r5b_sp8_trace
tail -n 4 $OUT/r5b_tests_sp.log
cat $OUT/r5b_gemm_split.log | cut -c1-200
head -25 $OUT/r5b_sp8_pair_kernel_stats.md | cut -c1-200
