#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table we commit under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof1/r01_results.db [--steps N] > profiles/r01_....txt
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else None
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                          "max(end-start)/1e3 from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    n = sum(r[1] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
    print(f"# total kernel time {tot:.1f} us over {n} dispatches" + (f" ({tot / steps:.1f} us / step, {steps} steps incl. warm-up)" if steps else ""))
    print(f"{'kernel':110s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for r in rows:
        print(f"{r[0][:110]:110s} {r[1]:6d} {r[2]:11.1f} {r[3]:9.2f} {r[4]:9.2f} {r[5]:9.2f} {100 * r[2] / tot:6.2f}")


if __name__ == "__main__":
    main()
