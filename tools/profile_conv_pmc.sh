#!/bin/bash
# PMC passes over ifx_conv3d_cl at the three dominant decoder shapes -> gpurun_out/<tag>_pmc_conv.md
set -u
TAG=${1:-r1g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MICRO="python $R/tools/pmc_micro.py conv 0 2"
: > $OUT/${TAG}_pmc_conv.md
pass() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "conv_(cl|pp)_kernel" -d $OUT/${TAG}_pmcc_$name -o pmc -- $MICRO > /dev/null 2> $OUT/${TAG}_pmcc_$name.err
  local db=$(ls $OUT/${TAG}_pmcc_$name/*/*.db $OUT/${TAG}_pmcc_$name/*.db 2>/dev/null | head -1)
  echo -e "\n## pass: $name\n" >> $OUT/${TAG}_pmc_conv.md
  python $R/tools/rocpd_pmc.py $db >> $OUT/${TAG}_pmc_conv.md 2>/dev/null
  rm -rf $OUT/${TAG}_pmcc_$name
}
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
