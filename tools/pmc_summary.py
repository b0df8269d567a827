#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc --output-format csv directory: per-kernel mean of every counter + derived ratios."""
import collections, csv, glob, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(csv.DictReader(open(glob.glob(d + "/*counter_collection.csv")[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if pat in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
kt = list(csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])))
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in kt if r["Kernel_Name"][:110] == k]
    print(k, f"launches={len(dur)} avg_us={sum(dur) / max(len(dur), 1):.1f}")
    for c, v in sorted(m.items()):
        print(f"  {c:28s} {v:.4g}")
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8          # summed over the 8 XCDs
        print(f"  effective clock           {cyc / (sum(dur) / len(dur)) / 1e3:.2f} GHz")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            print(f"  MFMA pipe busy            {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}  (of 1024 SIMDs x cycles)")
        if "SQ_WAVE_CYCLES" in m:
            print(f"  mean waves resident / CU  {4 * m['SQ_WAVE_CYCLES'] / (256 * cyc):.2f}")
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in m:
                    print(f"  {c:18s}/WAVE_CYCLES {m[c] / m['SQ_WAVE_CYCLES']:.3f}")
        if "SQ_LDS_IDX_ACTIVE" in m:
            print(f"  LDS array busy / CU       {m['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):.3f}  conflicts/active {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m['SQ_LDS_IDX_ACTIVE'], 1):.3f}")
