"""Per-kernel totals of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes NAME_results.db when the
csv writer is not selected): `python tools/rocpd_stats.py gpurun_out/prof/x_results.db [clips]` -> markdown table."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d "
                      f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"total kernel time {tot / 1e6 / div:.2f} ms over {sum(r[1] for r in rows) / div:.0f} dispatches (per 1/{div:g} of the trace)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|")
    for name, n, t, mn, mx in rows:
        if t / tot < 0.003:
            continue
        print(f"| `{name[:110]}` | {n / div:.0f} | {t / 1e6 / div:.3f} | {t / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * t / tot:.2f} |")


if __name__ == "__main__":
    main()
