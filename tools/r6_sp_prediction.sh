# round 6: one emulated rank of 2 / 4 / 8 (peer exchange replaced by local copies: zero wire cost) + the 1-GPU clip on the SAME box ->
# gpurun_out/sp_emulated_prediction.json (copy to profiles/: bench.py --gpus N reads it and prints measured / predicted).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs > $OUT/r6_sp1.json 2> $OUT/r6_sp.err
for P in 2 4 8; do
  python bench.py --emulate-sp $P --sp-exchange peer --steps 3 --warmup 1 --no-cpu-baseline --no-decode-leg --no-config-legs > $OUT/r6_sp$P.json 2>> $OUT/r6_sp.err
done
python - <<'PY'
import json, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
ms = {}
for p in (1, 2, 4, 8):
    ms[f"sp{p}"] = json.loads(open(os.path.join(out, f"r6_sp{p}.json")).read().strip().splitlines()[-1])["ms_per_step"]
doc = {"ms_per_clip": ms, "ratio_to_one_gpu": {k: round(ms["sp1"] / v, 3) for k, v in ms.items()},
       "source": "tools/r6_sp_prediction.sh: bench.py --emulate-sp P --sp-exchange peer (ONE rank of P emulated on one MI355X, the exchange replaced "
                 "by local copies = zero wire cost; sp1 = the 1-GPU clip on the same box, same session). INVALID as a multi-GPU measurement: an upper "
                 "bound on what a real sp-P rank can reach."}
json.dump(doc, open(os.path.join(out, "sp_emulated_prediction.json"), "w"), indent=1)
print(json.dumps(doc))
PY
