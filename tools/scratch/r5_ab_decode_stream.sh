#!/bin/bash
# A/B: per-block VAE decode on the main stream vs on a second stream under the next block's denoising (bench.py per_block_decode leg)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2; do
  for m in 0 1; do
    IFX_BENCH_DECODE_STREAM=$m python bench.py --steps 1 --warmup 1 --no-config-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('decode_stream=$m', r['ms_per_step'], r['per_block_decode']['ms_per_clip'])"
  done
done 2>&1 | tee gpurun_out/r5_ab_decode_stream.log
