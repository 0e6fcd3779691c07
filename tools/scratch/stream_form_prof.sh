# rocprofv3 kernel trace of the two eviction forms of the streaming leg -> gpurun_out/stream_<form>_kernel_stats.md
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in page shift; do
  python $R/tools/scratch/stream_form.py $f 10 > $OUT/stream_$f.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/stream_${f}_trace -o s -- python $R/tools/scratch/stream_form.py $f 10 > $OUT/stream_${f}_prof.log 2>&1
  DB=$(ls $OUT/stream_${f}_trace/*/*.db $OUT/stream_${f}_trace/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $OUT/stream_${f}_kernel_stats.md
  rm -rf $OUT/stream_${f}_trace
done
for f in page shift; do tail -n 2 $OUT/stream_$f.log; done
