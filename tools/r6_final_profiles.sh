# round 6: the profile set of the tree — headline kernel trace + PMC passes (attention / GEMM stamp), VAE decoder kernel trace, conv PMC
# passes (both kernels), the bench line.   usage (GPU box): bash tools/r6_final_profiles.sh [what...]   (what: bench conv vae line; default all)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
WHAT=${*:-bench conv vae line}
cd $R
for w in $WHAT; do
  case $w in
    bench) bash tools/profile_bench.sh r6 > $OUT/r6_profile_bench.log 2>&1 ;;
    conv)  bash tools/profile_conv_pmc.sh r6 > $OUT/r6_profile_conv.log 2>&1
           IFX_CONV_VARIANT=1 bash tools/profile_conv_pmc.sh r6lockstep > $OUT/r6_profile_conv_lockstep.log 2>&1 ;;
    vae)   (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/r6_vae_trace -o vae -- python $R/tools/bench_vae.py > $OUT/r6_vae_bench.json 2> $OUT/r6_vae_trace.err;
            DB=$(ls $OUT/r6_vae_trace/*/*.db $OUT/r6_vae_trace/*.db 2>/dev/null | head -1); python $R/tools/rocpd_summary.py $DB > $OUT/r6_vae_decode_kernel_stats.md; rm -rf $OUT/r6_vae_trace)
           python tools/bench_vae.py --detail > $OUT/r6_vae_detail.log 2>&1
           python tools/bench_conv.py --shapes all > $OUT/r6_bench_conv.log 2>&1 ;;
    line)  python bench.py > $OUT/r6_bench_line.json 2> $OUT/r6_bench_line.err ;;
  esac
done
ls -la $OUT | grep r6_ | tail -30
