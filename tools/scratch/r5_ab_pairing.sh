# round 5, first GPU call: the new parity tests, then the headline clip and the emulated sequence-parallel ranks with and without
# the layer-interleaved forward pairs (bench.py --pair).  usage (via gpurun): bash tools/r5_call1.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -m pytest tests/test_hip_model.py tests/test_hip_kernels.py -q -m gpu -x -s \
  -k "720p or ffn_down or bounded or paired or rollout or block_full or attention_480p or attention_720p or split_k" > $OUT/r5a_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r5a_tests.log
python -m pytest tests/test_hip_quant.py tests/test_hip_sequence_parallel.py -q -m gpu -x -s > $OUT/r5a_tests2.log 2>&1
echo "tests2 rc=$?" >> $OUT/r5a_tests2.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs"
for pair in off on off on; do
  $B --pair $pair >> $OUT/r5a_bench_n1_pair_$pair.json 2>> $OUT/r5a_bench_n1.err
done
for P in 8 4 2; do
  for pair in off on off on; do
    $B --emulate-sp $P --sp-exchange peer --pair $pair >> $OUT/r5a_bench_sp${P}_pair_$pair.json 2>> $OUT/r5a_bench_sp.err
  done
done
tail -3 $OUT/r5a_tests.log $OUT/r5a_tests2.log
for f in $OUT/r5a_bench_*.json; do echo $f; python - "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print("   value", d["value"], "ms_per_step", d["ms_per_step"], "pair", d["config"].get("pair_forwards"), "attn frac", d["roofline"]["frac"])
PY
done
