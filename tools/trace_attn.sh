#!/bin/bash
# Step-level timing of the ping-pong attention kernel: builds a PP_TRACE=1 library into tools/bin/abl/ (local, hipcc
# cross-compiles) and, on the GPU box, runs one launch and prints per-wave step durations in cycles.
# usage: tools/trace_attn.sh build | tools/trace_attn.sh run [L]
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/tools/bin/abl
if [ "$1" = "build" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPP_TRACE=1 ${TRACE_FLAGS:-} $R/inferix_amd/csrc/ifx_*.hip -o $R/tools/bin/abl/libtrace.so
else
  IFX_HIP_LIB=${TRACE_LIB:-$R/tools/bin/abl/libtrace.so} python - "$2" <<'PY'
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(os.environ["IFX_HIP_LIB"]))) + "/../..")
sys.path.insert(0, "/root/repo")
import torch
from inferix_amd import hip_ops as ops
L = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else 18720
NG = int(os.environ.get("ATTN_VARIANT", "2"))
ops.set_option("attn_variant", NG)
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device="cuda").to(torch.bfloat16)
q, k, v = rnd(4680, 12, 128), rnd(L, 12, 128), rnd(L, 12, 128)
for _ in range(2):
    out, lse = ops.attention(q, ops.KvCacheView(k, v), L, return_lse=True, splits=1)
torch.cuda.synchronize()
FR = NG == 4
SWP = NG in (5, 7)
if FR or SWP:
    NG = 2
W = 4 * NG
NTR = 48 if SWP else 64
tr = lse.reshape(-1).view(torch.int64)[:NTR * W * 8].view(NTR, W, 8).cpu()
import numpy as np
np.set_printoptions(linewidth=200)
t = tr.numpy().astype(np.int64)
if SWP:
    names = ["wait DMA      ", "barrier       ", "DMA issue     ", "body (PV|softmax|QK)", "lazy check + tail"]
    print("software-pipelined schedule, mean cycles over tiles 8..40, per wave 0..7 (waves w, w+4 share a SIMD)")
    for i, n in enumerate(names):
        print(n, (t[8:40, :, i + 1] - t[8:40, :, i]).mean(0).round(0))
    print("tile period   ", (t[9:41, :, 0] - t[8:40, :, 0]).mean(0).round(0))
    print("tile 20, stamps relative to wave 0's start: rows = wave")
    print(t[20, :, :6] - t[20, 0, 0])
    sys.exit(0)
if FR:
    names = ["wait DMA      ", "barrier       ", "DMA issue     ", "QK            ", "softmax       ", "PV            "]
elif NG == 2:
    names = ["wait DMA (G0) ", "barrier A     ", "M step        ", "wait DMA (G1) ", "barrier B     ", "V step        "]
else:
    names = ["wait+barrier M", "M step        ", "barrier V1    ", "V1 (+DMA wait)", "barrier V2    ", "V2 step       "]
print(f"mean cycles over tiles 8..56, per wave 0..{W-1} (waves w, w+4, w+8 share a SIMD)")
for i, n in enumerate(names):
    print(n, (t[8:56, :, i + 1] - t[8:56, :, i]).mean(0).round(0))
if FR:
    print("one tile, absolute cycle stamps relative to wave 0's start (tile 20): rows = wave, cols = stamps")
    print(t[20, :, :7] - t[20, 0, 0])
print("tile period   ", (t[9:57, :, 0] - t[8:56, :, 0]).mean(0).round(0))
PY
fi
