"""Can the 28 CUs that the 228-workgroup attention launch leaves idle take a share of the keys?  Two concurrent launches on two
streams — all tiles over keys [0, c) and all tiles over [c, L) — writing fp32 partials, one merge; against the single launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inferix_amd import hip_ops as ops  # noqa: E402

BF = torch.bfloat16


def main():
    dev = torch.device("cuda:0")
    N, H, hd = 4680, 12, 128
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(N, H, hd, generator=g, device=dev).to(BF)
    side = torch.cuda.Stream()
    for L in (4680, 9360, 18720, 32760):
        k = torch.randn(L, H, hd, generator=g, device=dev).to(BF)
        v = torch.randn(L, H, hd, generator=g, device=dev).to(BF)
        view = ops.KvCacheView(k, v)
        out = torch.empty_like(q)
        ref = ops.attention(q, view, L, splits=1).clone()

        def timeit(fn, n=20):
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

        base = timeit(lambda: ops.attention(q, view, L, out=out, splits=1))
        row = [f"L={L:6d} single {base:7.1f} us"]
        ws = ops.attention_workspace(q, 2)
        for frac in (0.86, 0.88, 0.90, 0.92):
            c = int(L * frac) // 64 * 64

            def lean():
                main_s = torch.cuda.current_stream()
                side.wait_stream(main_s)
                u1 = ops.attention_partial(q, view, c, 0, 1, ws, 0, 2)
                with torch.cuda.stream(side):
                    u2 = ops.attention_partial(q, view, L, c, 1, ws, u1, 2)
                main_s.wait_stream(side)
                ops.attention_merge(ws, 2, u1 + u2, out)
            t = timeit(lean)
            err = float((out.float() - ref.float()).abs().max())
            row.append(f"{frac:.2f}: {t:7.1f} ({err:.1e})")
        print("  ".join(row), flush=True)


if __name__ == "__main__":
    main()
