"""Interleaved A/B of GEMM tiles on one shape (boxes and the power state drift by several per cent inside one process: single passes mislead).
usage: python tools/scratch/ab_gemm_tiles.py M N K v0,v1,...  [rounds]"""
import statistics
import sys

import torch

sys.path.insert(0, "/root/repo")
from inferix_amd import hip_ops as ops

M, N, K = (int(a) for a in sys.argv[1:4])
variants = [int(v) for v in sys.argv[4].split(",")]
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
x, w, b = rnd(M, K), rnd(N, K) * 0.03, rnd(N)
y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
times = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        ops.set_option("gemm_variant", v)
        for _ in range(3):
            ops.linear(x, w, b, out=y)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30):
            ops.linear(x, w, b, out=y)
        e.record()
        torch.cuda.synchronize()
        if r:
            times[v].append(s.elapsed_time(e) / 30 * 1e3)
ops.set_option("gemm_variant", 0)
for v in variants:
    t = times[v]
    print(f"{M}x{N}x{K} variant {v:2d}: median {statistics.median(t):7.1f} us  min {min(t):7.1f}  max {max(t):7.1f}   {2.0 * M * N * K / statistics.median(t) / 1e6:7.1f} TFLOP/s")
