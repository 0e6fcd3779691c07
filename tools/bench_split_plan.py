"""Split-KV attention of a sequence-parallel rank's shard (585 / 1170 rows): time of the split launch + merge against the number of
key chunks, next to what ifx_attn_split_plan picks.  `python tools/bench_split_plan.py [rows]`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inferix_amd import hip_ops as ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 585
    dev = torch.device("cuda:0")
    H, hd = 12, 128
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(rows, H, hd, generator=g, device=dev).to(BF)
    out = torch.empty_like(q)
    for L in (4680, 9360, 18720, 28080, 32760):
        k = torch.randn(L, H, hd, generator=g, device=dev).to(BF)
        v = torch.randn(L, H, hd, generator=g, device=dev).to(BF)
        view = ops.KvCacheView(k, v)
        plan = ops.attention_split_plan(rows, H, L)
        row = []
        for s in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
            if s * 8 * 64 > L and s > 1:
                continue
            row.append(f"s{s}:{timeit(lambda: ops.attention(q, view, L, out=out, splits=s)):6.1f}")
        print(f"rows {rows} L={L:6d} plan={plan:2d}  " + " ".join(row), flush=True)


if __name__ == "__main__":
    main()
