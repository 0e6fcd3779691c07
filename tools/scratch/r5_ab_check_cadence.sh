#!/bin/bash
# What a host synchronisation per forward costs a sequence-parallel rank: the emulated rank with the status word read (IFX_SP_LAB_CHECK=1)
# every 1 / 5 / 1000 forwards (a real rank's check also runs a small collective on top)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { python bench.py --steps 3 --warmup 1 --no-config-legs --no-cpu-baseline --no-decode-leg "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; }
export IFX_SP_LAB_CHECK=1
for rep in 1 2 3; do
  for ev in 1 5 1000; do
    export IFX_SP_CHECK_EVERY=$ev
    echo "check_every=$ev sp8 $(run --emulate-sp 8 --sp-exchange peer) sp4 $(run --emulate-sp 4 --sp-exchange peer)"
  done
done 2>&1 | tee gpurun_out/r5_ab_check_cadence.log
