#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a PMC summary written by tools/profile_bench.sh (`tools/pmc_micro.py attn,gemm 18720 3`):
  * `attn_self`: the self-attention kernel's FETCH_SIZE / WRITE_SIZE per launch and its matrix-pipe occupancy;
  * `gemm_block`: the six GEMM launches of ONE transformer block at 4680 rows (QKV, O + gate, cross q, cross o, FFN up + GELU,
    FFN down + gate) summed: fetched / written KiB, MFMA busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (4 x sum SQ_BUSY_CU_CYCLES), per kernel rows.
Each section is stamped with the sha256 of the kernel sources it was measured on; bench.py reports `traffic` / `mfma_busy` as null
when a source has changed since (tests/test_cabi_and_host.py::test_pmc_traffic_stamp_matches_the_kernel_sources fails then).
usage: tools/update_pmc_traffic.py <tag>_pmc.md out.json <tag>"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
md, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
ATTN_SOURCES = ["inferix_amd/csrc/ifx_attn_pp.hip"]
GEMM_SOURCES = ["inferix_amd/csrc/ifx_gemm_pp.hip", "inferix_amd/csrc/ifx_gemm.hip"]      # the tile and its picker


def sha(paths):
    h = hashlib.sha256()
    for p in paths:
        h.update(open(os.path.join(ROOT, p), "rb").read())
    return h.hexdigest()


rows = {}          # kernel -> counter -> (dispatches, avg)
for line in open(md):
    m = re.match(r"\| `void ifx::(\w+<[^>]*>).*` \| (\w+) \| (\d+) \| ([0-9.e+]+) \|", line)
    if m:
        rows.setdefault(m.group(1), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)))
attn = [k for k in rows if re.match(r"attn_fwd_pp_kernel<(0|false), false, 2,", k)]
if len(attn) != 1 or not {"FETCH_SIZE", "WRITE_SIZE"} <= set(rows[attn[0]]):
    raise SystemExit(f"attention rows not found in {md}: {sorted(rows)}")
a = rows[attn[0]]
doc = {"attn_self": {
    "kernel": f"ifx::{attn[0]} (software-pipelined schedule)",
    "shape": "N=4680 queries x 12 heads, L=18720 keys (mean prefix of the 21-frame clip)",
    "fetch_size_kib": round(a["FETCH_SIZE"][1]), "write_size_kib": round(a["WRITE_SIZE"][1]),
    "mfma_busy": round(a["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (4 * a["SQ_BUSY_CU_CYCLES"][1]), 4) if "SQ_BUSY_CU_CYCLES" in a else None,
    "kernel_source": ATTN_SOURCES[0], "kernel_source_sha256": sha(ATTN_SOURCES),
    "source": f"profiles/{tag}_pmc_attn_gemm.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_micro.py)"}}
gem = {k: v for k, v in rows.items() if k.startswith("gemm_") and {"FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"} <= set(v)}
if gem:
    reps = max(v["FETCH_SIZE"][0] for v in gem.values())          # pmc_micro runs every launch of the block `reps` times
    per = {}
    tot = dict(fetch=0.0, write=0.0, busy=0.0, cu=0.0, launches=0)
    for k, v in sorted(gem.items()):
        n = v["FETCH_SIZE"][0] / reps                               # launches of this kernel per block (two 1536^2 epilogues may share one)
        per[k] = {"launches_per_block": n, "fetch_size_kib": round(v["FETCH_SIZE"][1]), "write_size_kib": round(v["WRITE_SIZE"][1]),
                  "mfma_busy": round(v["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (4 * v["SQ_BUSY_CU_CYCLES"][1]), 4)}
        tot["fetch"] += n * v["FETCH_SIZE"][1]
        tot["write"] += n * v["WRITE_SIZE"][1]
        tot["busy"] += n * v["SQ_VALU_MFMA_BUSY_CYCLES"][1]
        tot["cu"] += n * 4 * v["SQ_BUSY_CU_CYCLES"][1]
        tot["launches"] += n
    doc["gemm_block"] = {
        "shape": "the six GEMM launches of one block at 4680 rows: QKV 4608x1536, O+gate / cross-q / cross-o 1536x1536, FFN up 8960x1536 + GELU, FFN down 1536x8960 + gate",
        "launches_per_block": tot["launches"], "fetch_size_kib": round(tot["fetch"]), "write_size_kib": round(tot["write"]),
        "mfma_busy": round(tot["busy"] / tot["cu"], 4), "kernels": per,
        "kernel_sources": GEMM_SOURCES, "kernel_source_sha256": sha(GEMM_SOURCES),
        "source": f"profiles/{tag}_pmc_attn_gemm.md (same passes)"}
json.dump(doc, open(out, "w"), indent=2)
print("wrote", out)
