#!/usr/bin/env python3
"""L1 (TCP) view of a rocprofv3 --pmc --kernel-trace --output-format csv directory: per (kernel, grid size) the mean over its
launches of every counter (summed over the rows a dispatch has), the launch duration, and the requests sent on to L2 per L1 tag lookup.

    python tools/pmc_l1_summary.py <dir> [min_launches]

TCP_TOTAL_CACHE_ACCESSES counts 128-byte line (tag) lookups, TCP_TCC_READ_REQ the read requests the L1 sends on to L2: their ratio is
the share of lookups that missed.  Streaming kernels of the same pass (fills, copies: no reuse by construction) calibrate what "every
lookup misses" reads on this part."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
min_n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfiles = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not cfiles:
    sys.exit(f"no counter_collection.csv under {d}")
rows = list(csv.DictReader(open(cfiles[0])))
grid_cols = [c for c in (rows[0].keys() if rows else []) if c.startswith("Grid_Size")]
per_disp = collections.defaultdict(lambda: collections.defaultdict(float))     # dispatch -> counter -> sum over instance rows
meta = {}
for r in rows:
    did = int(r["Dispatch_Id"])
    per_disp[did][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[did] = (r["Kernel_Name"][:96], "x".join(r[c] for c in grid_cols))
dur = {}
for kt in glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[:1]:
    for r in csv.DictReader(open(kt)):
        if r.get("Dispatch_Id"):
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# launches re-issued back to back (bench.py's time_kernel / time_rpn_conv loops: >= 20 in a row) are kept apart from the same kernel
# inside a step -- the RPN conv of a step runs on its live-tile lists, the re-issued one convolves every tile
ids_sorted = sorted(meta)
run_len = {}
i = 0
while i < len(ids_sorted):
    j = i
    while j + 1 < len(ids_sorted) and meta[ids_sorted[j + 1]] == meta[ids_sorted[i]]:
        j += 1
    for q in range(i, j + 1):
        run_len[ids_sorted[q]] = j - i + 1
    i = j + 1
groups = collections.defaultdict(list)
for did, key in meta.items():
    groups[(key[0], key[1] + (" burst" if run_len[did] >= 20 else ""))].append(did)
tot = lambda k: sum(dur.get(i, 0.0) for i in groups[k])
print(f"# {cfiles[0]}: {len(rows)} counter rows, {len(meta)} dispatches, {len(groups)} (kernel, grid) groups; durations are under the profiler")
print(f"{'kernel':96s} {'grid':>12s} {'n':>4s} {'us':>8s} {'tag lookups':>12s} {'L2 read req':>12s} {'req/lookup':>10s} {'pend.stall/lookup':>17s}")
for k in sorted(groups, key=tot, reverse=True):
    ids = groups[k]
    if len(ids) < min_n:
        continue
    names = sorted({c for i in ids for c in per_disp[i]})
    m = {c: sum(per_disp[i].get(c, 0.0) for i in ids) / len(ids) for c in names}
    ds = [dur[i] for i in ids if i in dur]
    us = sum(ds) / len(ds) if ds else float("nan")
    look, req, stall = m.get("TCP_TOTAL_CACHE_ACCESSES_sum"), m.get("TCP_TCC_READ_REQ_sum"), m.get("TCP_PENDING_STALL_CYCLES_sum")
    ratio = req / look if look and req is not None else float("nan")
    sr = stall / look if look and stall is not None else float("nan")
    print(f"{k[0]:96s} {k[1]:>12s} {len(ids):4d} {us:8.2f} {look or 0:12.4g} {req or 0:12.4g} {ratio:10.3f} {sr:17.2f}")
    for c in names:
        if c not in ("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum"):
            print(f"    {c:40s} {m[c]:.5g}")
