#!/bin/bash
# A/B: GPU_MAX_HW_QUEUES unset (4 hardware queues: torch's pool streams alias, tools/probe_stream_queues.py) vs 8, on the paired clip
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { python bench.py --steps 2 --warmup 1 --no-config-legs --no-cpu-baseline --no-decode-leg "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; }
for rep in 1 2; do
  for q in 4 8 16; do
    export GPU_MAX_HW_QUEUES=$q
    echo "queues=$q n1 $(run) sp8 $(run --emulate-sp 8 --sp-exchange peer) sp4 $(run --emulate-sp 4 --sp-exchange peer)"
  done
done 2>&1 | tee gpurun_out/r5_ab_hw_queues.log
