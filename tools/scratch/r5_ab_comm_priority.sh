#!/bin/bash
# A/B: exchange streams of an emulated sequence-parallel rank at default vs high stream priority
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { python bench.py --steps 2 --warmup 1 --no-config-legs --no-cpu-baseline --no-decode-leg "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; }
for rep in 1 2; do
  for pr in 0 -1; do
    export IFX_SP_COMM_PRIORITY=$pr
    echo "priority=$pr sp8 $(run --emulate-sp 8 --sp-exchange peer) sp4 $(run --emulate-sp 4 --sp-exchange peer) sp2 $(run --emulate-sp 2 --sp-exchange peer)"
  done
done 2>&1 | tee gpurun_out/r5_ab_comm_priority.log
