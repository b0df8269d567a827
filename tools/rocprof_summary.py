#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table we commit under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof1/r01_results.db [--steps N] > profiles/r01_....txt
    python tools/rocprof_summary.py <db> --after 0.5 --steps N      # only the dispatches of the last half of the run (steady state)
    python tools/rocprof_summary.py <db> --last-steps 20 --marker k_vox_init   # the 20 steps before the marker kernel's last launch
    python tools/rocprof_summary.py <db> --timeline k_vox_init      # one step (between the last two launches of that kernel):
                                                                    # every launch with start offset, duration and the idle gap before it
"""
import sqlite3
import sys


def timeline(db, marker):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 3:
        print(f"# fewer than three launches of {marker}")
        return
    a, b = marks[-3], marks[-2]          # the last complete step
    t0, prev_end = rows[a][1], rows[a][1]
    busy = gaps = 0.0
    print(f"# one step = launches {a}..{b - 1} of {db}: start offset, duration, idle gap before the launch (us)")
    for name, st, en in rows[a:b]:
        gap = (st - prev_end) / 1e3
        print(f"{(st - t0) / 1e3:9.2f} {(en - st) / 1e3:8.2f} {gap:7.2f}  {name[:100]}")
        busy += (en - st) / 1e3
        gaps += max(gap, 0.0)
        prev_end = max(prev_end, en)
    print(f"# {b - a} launches, {busy:.1f} us of kernels, {gaps:.1f} us idle between them, {(prev_end - t0) / 1e3:.1f} us first start to last end")


def main():
    db = sys.argv[1]
    if "--timeline" in sys.argv:
        return timeline(db, sys.argv[sys.argv.index("--timeline") + 1])
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else None
    c = sqlite3.connect(db)
    where = ""
    if "--after" in sys.argv:        # only dispatches in the last part of the run, e.g. --after 0.5: skips warm-up (MIOpen's find runs)
        frac = float(sys.argv[sys.argv.index("--after") + 1])
        t0, t1 = c.execute("select min(start), max(end) from kernels").fetchone()
        where = f" where start >= {t0 + (t1 - t0) * frac:.0f}"
    if "--last-steps" in sys.argv:   # --last-steps N --marker k_vox_init: the N steps before the last launch of the marker kernel
        nst = int(sys.argv[sys.argv.index("--last-steps") + 1])
        marker = sys.argv[sys.argv.index("--marker") + 1]
        marks = [r[0] for r in c.execute(f"select start from kernels where name like '%{marker}%' order by start")]
        if len(marks) > nst:
            where = f" where start >= {marks[-nst - 1]} and start < {marks[-1]}"
            steps = nst
    rows = list(c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                          f"max(end-start)/1e3 from kernels{where} group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    n = sum(r[1] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
    print(f"# total kernel time {tot:.1f} us over {n} dispatches" + (f" ({tot / steps:.1f} us / step, {steps} steps incl. warm-up)" if steps else ""))
    print(f"{'kernel':110s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for r in rows:
        print(f"{r[0][:110]:110s} {r[1]:6d} {r[2]:11.1f} {r[3]:9.2f} {r[4]:9.2f} {r[5]:9.2f} {100 * r[2] / tot:6.2f}")


if __name__ == "__main__":
    main()
