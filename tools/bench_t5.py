"""umT5-XXL encoder (24 layers, dim 4096, 64 heads, ffn 10240; synthetic weights generated on the device) on one MI355X:
time per prompt batch at the pipelines' 512-token padding.     python tools/bench_t5.py [--batch 1] [--detail]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--len", type=int, default=512)
    ap.add_argument("--tokens", type=int, default=60, help="valid tokens per prompt (the rest is padding)")
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--detail", action="store_true")
    a = ap.parse_args()
    from inferix_amd import hip_ops as ops
    from inferix_amd.t5 import HipWanTextEncoder, synthetic_t5_state_dict
    t0 = time.perf_counter()
    sd = synthetic_t5_state_dict(num_layers=a.layers, device="cuda")
    enc = HipWanTextEncoder(sd, None, num_layers=a.layers)
    del sd
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(1, 256384, (a.batch, a.len), generator=g)
    mask = torch.zeros(a.batch, a.len, dtype=torch.long)
    mask[:, :a.tokens] = 1
    ids[:, a.tokens:] = 0

    def run():
        return enc.encode_ids(ids, mask)["prompt_embeds"]

    run()
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        out = run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    M = a.batch * a.len
    flops = a.layers * (2.0 * M * 4096 * (3 * 4096 + 4096 + 2 * 10240) + 2.0 * M * 10240 * 4096 + 4.0 * a.batch * 64 * a.len * a.len * 64)
    wbytes = a.layers * (4 * 4096 * 4096 + 3 * 4096 * 10240) * 2
    res = {"workload": f"umT5-XXL encoder, {a.layers} layers, batch {a.batch} x {a.len} tokens ({a.tokens} valid), bf16, synthetic weights",
           "ms": round(ms, 2), "tflops": round(flops / ms / 1e9, 1), "weight_GB": round(wbytes / 1e9, 2),
           "weight_stream_TBps": round(wbytes / ms / 1e9, 2), "build_s": round(build_s, 1), "finite": bool(torch.isfinite(out.float()).all())}
    if a.detail:
        t = ops.KernelTimer(names=("gemm", "attn_t5", "rmsnorm"))
        ops.set_kernel_timer(t)
        run()
        ops.set_kernel_timer(None)
        for k, d in t.summary().items():
            res[k] = {"launches": d["launches"], "ms": round(d["ms"], 2),
                      "TFLOP/s": round(d["flops"] / d["ms"] / 1e9, 1) if d["flops"] else None}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
