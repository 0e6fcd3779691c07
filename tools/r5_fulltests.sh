set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q -x > $OUT/r5_gputests.log 2>&1
echo "rc=$?" >> $OUT/r5_gputests.log
tail -n 15 $OUT/r5_gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r5_smoke.log 2>&1; tail -3 $OUT/r5_smoke.log
