set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
true
true
true
python bench.py > $OUT/r5g_bench.json 2> $OUT/r5g_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5g_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roof", d["roofline"]["frac"], "gemm", d.get("roofline_gemm", {}).get("frac"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("runs_s"))
for k in ("quant_fp8", "quant_int8"):
    print(k, d[k]["ms_per_clip"], d[k]["speedup_vs_bf16"], d[k]["roofline"]["frac"])
print("causvid", d["causvid_720p"]["ms_total"], d["causvid_720p"]["roofline"]["frac"])
print("magi", d["magi_cp8_emulated"]["ms_clip_rank"])
print("per_block", d["per_block_decode"])
PY
