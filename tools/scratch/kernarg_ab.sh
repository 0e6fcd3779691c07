# lab: does the location of the kernel-argument buffers (HIP_FORCE_DEV_KERNARG) move the dependent-launch gaps of a clip?
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config-legs --no-decode-leg"
for rep in 1 2; do
  for v in default 1 0; do
    if [ $v = default ]; then $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['timed_clip_ms_per_block']['sum'])"
    else HIP_FORCE_DEV_KERNARG=$v $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HIP_FORCE_DEV_KERNARG=$v', d['ms_per_step'], d['timed_clip_ms_per_block']['sum'])"
    fi
  done
done | tee $OUT/kernarg_ab.log
