#!/bin/bash
# Runs on the GPU box: kernel trace of bench.py's MAGI leg alone (one rank of cp = 8, emulated) -> gpurun_out/<tag>_magi_kernel_stats.md
# usage: tools/profile_magi_leg.sh <tag> [fp8|bf16]
TAG=${1:-r2}; MODE=${2:-fp8}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/magi_leg.py <<PY
import sys, json, torch
sys.path.insert(0, "$R")
import bench
r = bench.magi_cp8_emulated_leg(torch.device("cuda", 0), fp8_quant=("$MODE" == "fp8"))
print(json.dumps({k: v for k, v in r.items() if k != "workload"}))
PY
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_magi -o magi -- python /tmp/magi_leg.py > $OUT/${TAG}_magi_leg.json 2> $OUT/${TAG}_magi.err
DB=$(ls $OUT/${TAG}_magi/*/*.db $OUT/${TAG}_magi/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB > $OUT/${TAG}_magi_kernel_stats.md
rm -rf $OUT/${TAG}_magi
tail -1 $OUT/${TAG}_magi_leg.json | cut -c1-400
head -28 $OUT/${TAG}_magi_kernel_stats.md | cut -c1-200
