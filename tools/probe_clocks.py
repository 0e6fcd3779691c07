#!/usr/bin/env python3
"""Which launches are clock(power)-limited?  Loops ONE kernel for a few seconds while a side thread samples the shader clock and the socket
power (`rocm-smi --showclocks --showpower --json`), then prints the median of the samples taken while the loop ran.
usage (GPU box): python tools/probe_clocks.py [seconds per kernel]"""
import json
import os
import statistics
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inferix_amd import _hip, hip_ops as ops  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            j = json.loads(r.stdout)
            c = next(iter(j.values()))
            sclk = next((v for k, v in c.items() if k.startswith("sclk")), None)
            pw = next((v for k, v in c.items() if "ower" in k and "W" in k), None)
            out.append((sclk, pw, time.time()))
        except Exception as e:  # noqa: BLE001
            out.append((repr(e), None, time.time()))
        time.sleep(0.15)


def run(name, fn, flops):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples))
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < SECS:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    dt = time.time() - t0
    stop.set()
    th.join()
    mid = [s for s in samples if t0 + 1.0 < s[2] < t0 + dt - 0.2]
    def num(x):
        import re
        m = re.search(r"([0-9.]+)", str(x))
        return float(m.group(1)) if m else float("nan")
    clk = [num(s[0]) for s in mid if s[0] is not None]
    pw = [num(s[1]) for s in mid if s[1] is not None]
    med = lambda v: statistics.median(v) if v else float("nan")
    print(f"{name:34s} {dt / n * 1e6:8.1f} us  {flops / (dt / n) / 1e12:7.1f} TFLOP/s   sclk median {med(clk):6.0f} MHz  power median {med(pw):6.0f} W   "
          f"({len(mid)} samples; raw {mid[len(mid) // 2][:2] if mid else None})", flush=True)


M, d, f, H, D = 4680, 1536, 8960, 12, 128
x, u = rnd(M, d), rnd(M, f)
print("idle:", subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout[-900:])
w = rnd(f, d) * 0.03
b = rnd(f)
o = torch.empty(M, f, dtype=torch.bfloat16, device=dev)
run("gemm FFN up (gelu) 4680x8960x1536", lambda: ops.linear(x, w, b, out=o, epilogue=_hip.IFX_EPI_GELU_TANH), 2.0 * M * f * d)
run("gemm FFN up (bias) 4680x8960x1536", lambda: ops.linear(x, w, b, out=o), 2.0 * M * f * d)
w2 = rnd(d, f) * 0.01
o2 = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
run("gemm FFN down 4680x1536x8960", lambda: ops.linear(u, w2, None, out=o2), 2.0 * M * f * d)
w3 = rnd(3 * d, d) * 0.03
o3 = torch.empty(M, 3 * d, dtype=torch.bfloat16, device=dev)
run("gemm QKV 4680x4608x1536", lambda: ops.linear(x, w3, None, out=o3), 2.0 * M * 3 * d * d)
q = rnd(M, H, D)
for L in (4680, 32760):
    k, v = rnd(L, H, D), rnd(L, H, D)
    oo = torch.empty_like(q)
    kvv = ops.KvCacheView(k, v)
    run(f"attention L={L}", lambda: ops.attention(q, kvv, L, out=oo), 4.0 * M * L * H * D)
qs = rnd(512, H, D)
L = 32760
run("attention 512 rows (24 of 256 CUs)", lambda: ops.attention(qs, kvv, L, out=oo[:512], splits=1), 4.0 * 512 * L * H * D)
mod = rnd(3, 6, d)
xo = torch.empty_like(x)
run("layernorm 4680x1536", lambda: ops.layernorm(x, 1e-6, mod=mod, rows_per_group=(M + 2) // 3, out=xo), 0.0)
