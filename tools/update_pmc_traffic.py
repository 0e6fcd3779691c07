#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a PMC summary written by tools/profile_bench.sh: the self-attention kernel's FETCH_SIZE / WRITE_SIZE
per launch, stamped with the sha256 of the kernel source it was measured on (bench.py refuses the figure when the source changed).
usage: tools/update_pmc_traffic.py <tag>_pmc.md out.json <tag>"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
md, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
vals = {}
for line in open(md):
    m = re.match(r"\| `void ifx::attn_fwd_pp_kernel<false, false, 2, (\d)>.*` \| (FETCH_SIZE|WRITE_SIZE) \| (\d+) \| ([0-9.e+]+) \|", line)
    if m:
        vals[m.group(2)] = float(m.group(4))
        schedule = m.group(1)
if set(vals) != {"FETCH_SIZE", "WRITE_SIZE"}:
    raise SystemExit(f"attention rows not found in {md}: {vals}")
src = os.path.join(ROOT, "inferix_amd", "csrc", "ifx_attn_pp.hip")
json.dump({"attn_self": {
    "kernel": f"ifx::attn_fwd_pp_kernel<false, false, 2, {schedule}> (software-pipelined schedule)",
    "shape": "N=4680 queries x 12 heads, L=18720 keys (mean prefix of the 21-frame clip)",
    "fetch_size_kib": round(vals["FETCH_SIZE"]), "write_size_kib": round(vals["WRITE_SIZE"]),
    "kernel_source": "inferix_amd/csrc/ifx_attn_pp.hip", "kernel_source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
    "source": f"profiles/{tag}_pmc_attn_gemm.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_micro.py)"}},
    open(out, "w"), indent=2)
print("wrote", out)
