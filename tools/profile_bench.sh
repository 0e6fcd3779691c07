#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of bench.py + PMC passes over tools/pmc_micro.py (the full bench under
# --pmc crashes rocprofv3 on this image, so counters are collected on the dominant kernels at the mean-prefix shapes).
# Markdown summaries land in gpurun_out/<tag>_*.md; copy the ones to keep into profiles/.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config-legs $*"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o bench -- $BENCH > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
DB=$(ls $OUT/${TAG}_trace/*/*.db $OUT/${TAG}_trace/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB > $OUT/${TAG}_kernel_stats.md
MICRO="python $R/tools/pmc_micro.py attn,gemm 18720 3"   # (the MAGI long-K launches, "w4", have their own table in profiles/r3_pmc_attn_gemm.md)
RX="attn_fwd|gemm_"
: > $OUT/${TAG}_pmc.md
pass() {  # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$RX" -d $OUT/${TAG}_pmc_$name -o pmc -- $MICRO > /dev/null 2> $OUT/${TAG}_pmc_$name.err
  local db=$(ls $OUT/${TAG}_pmc_$name/*/*.db $OUT/${TAG}_pmc_$name/*.db 2>/dev/null | head -1)
  echo -e "\n## pass: $name\n" >> $OUT/${TAG}_pmc.md
  python $R/tools/rocpd_pmc.py $db >> $OUT/${TAG}_pmc.md 2>/dev/null
}
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
python $R/tools/update_pmc_traffic.py $OUT/${TAG}_pmc.md $OUT/${TAG}_pmc_traffic.json ${TAG}
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_mfma $OUT/${TAG}_pmc_lds 2>/dev/null
ls -la $OUT | tail -12
