#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of bench.py, outputs under gpurun_out/<tag>_*
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o bench -- $BENCH > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
SMALL="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $*"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "attn_fwd_kernel|gemm_bf16" -d $OUT/${TAG}_pmc_$C -o pmc -- $SMALL > /dev/null 2> $OUT/${TAG}_pmc_$C.err
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-include-regex "attn_fwd_kernel|gemm_bf16" -d $OUT/${TAG}_pmc_mfma -o pmc -- $SMALL > /dev/null 2> $OUT/${TAG}_pmc_mfma.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "attn_fwd_kernel|gemm_bf16" -d $OUT/${TAG}_pmc_lds -o pmc -- $SMALL > /dev/null 2> $OUT/${TAG}_pmc_lds.err
ls -la $OUT | tail -20
