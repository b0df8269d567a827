#!/usr/bin/env python3
"""rocprofv3 --pmc pass directories of tools/pmc_workload.py -> the text report and the traffic json bench.py reads.

    python tools/pmc_report.py <meta.json> <out_prefix> <pass_dir> [<pass_dir> ...]

Per kernel instantiation: mean of every counter over its launches (warm-ups included: identical launches), mean duration from the
kernel trace of the same pass, derived ratios (effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA pipe busy =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles); SQ_* cycle counters are quad-cycles).  HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE,
KiB -> bytes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte read requests tallied at 64 B); the two
counters come from separate passes (TCC slots)."""
import collections
import csv
import glob
import json
import sys


def load(d):
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not cc:
        return {}, {}
    # values in DISPATCH order (the same instantiation serves several layers: pmc_workload.py's meta says how many launches each
    # layer issued, main() cuts the sequence accordingly)
    tmp = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(cc[0])):
        d_ = tmp[r["Kernel_Name"]][r["Counter_Name"]]
        did = int(r["Dispatch_Id"])
        d_[did] = d_.get(did, 0.0) + float(r["Counter_Value"])
    acc = {k: {c: [v[i] for i in sorted(v)] for c, v in cs.items()} for k, cs in tmp.items()}
    dur = collections.defaultdict(list)
    if kt:
        rows = sorted(csv.DictReader(open(kt[0])), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc, dur


def main():
    meta = json.load(open(sys.argv[1]))
    prefix, dirs = sys.argv[2], sys.argv[3:]
    passes = [load(d) for d in dirs]
    names = sorted({k for acc, _ in passes for k in acc})
    cursor = collections.defaultdict(int)        # kernel name -> launches already attributed to earlier meta entries
    lines, entries = [], []
    for l in meta["launches"]:
        sig = l["kernel_signature"]
        full = [k for k in names if sig in k]
        if not full:
            lines.append(f"== {sig}: not found in the counter files")
            continue
        k = full[0]
        lo = cursor[k]
        hi = lo + int(l.get("count", 10 ** 9))
        cursor[k] = hi
        m, dd = {}, []
        for acc, dur in passes:
            for c, v in acc.get(k, {}).items():
                seg = v[lo:hi]
                if seg:
                    m[c] = sum(seg) / len(seg)
            seg = dur.get(k, [])[lo:hi]
            if seg:
                dd.append(sum(seg) / len(seg))
        if not m:
            lines.append(f"== {sig}: launches {lo}..{hi} not found in the counter files")
            continue
        durs = {k: dd}
        us = sum(durs[k]) / max(len(durs[k]), 1)
        lines.append(f"== {sig}   ({l['workload']})")
        lines.append(f"  launch under the profiler    {us:.2f} us (mean over {len(durs[k])} passes)")
        for c, v in sorted(m.items()):
            lines.append(f"  {c:28s} {v:.5g}")
        ent = dict(l, launch_us_under_profiler=round(us, 2))
        if "GRBM_GUI_ACTIVE" in m and us > 0:
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            ghz = cyc / us / 1e3
            if ghz > 2.4:
                # GRBM_GUI_ACTIVE also counts the dispatch / drain around a launch this short: the ratio exceeds the 2.4 GHz
                # maximum clock.  The busy fractions below then use duration x 2.4 GHz (an upper bound on the kernel's cycles,
                # so the fractions are lower bounds).
                lines.append(f"  GRBM_GUI_ACTIVE / 8 / duration = {ghz:.2f} GHz > 2.4 GHz max: counter spans more than the launch; "
                             f"fractions below use duration x 2.4 GHz")
                cyc, ghz = us * 2400.0, 2.4
                ent["clock_note"] = "GRBM_GUI_ACTIVE spans more than the launch; 2.4 GHz assumed"
            lines.append(f"  effective clock              {ghz:.2f} GHz")
            ent["effective_clock_ghz"] = round(ghz, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)
                lines.append(f"  MFMA pipe busy               {busy:.3f}  (of 1024 SIMDs x cycles)")
                ent["mfma_pipe_busy"] = round(busy, 4)
            if "SQ_WAVE_CYCLES" in m:
                lines.append(f"  mean waves resident / CU     {4 * m['SQ_WAVE_CYCLES'] / (256 * cyc):.2f}")
                for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                    if c in m:
                        lines.append(f"  {c:18s}/WAVE_CYCLES {m[c] / m['SQ_WAVE_CYCLES']:.3f}")
            if "SQ_LDS_IDX_ACTIVE" in m:
                lines.append(f"  LDS array busy / CU          {m['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):.3f}  conflicts/active "
                             f"{m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m['SQ_LDS_IDX_ACTIVE'], 1):.3f}")
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
            lines.append(f"  L2 hit rate                  {m['TCC_HIT_sum'] / max(m['TCC_HIT_sum'] + m['TCC_MISS_sum'], 1):.3f}")
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            traffic = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
            ent.update(FETCH_SIZE_KiB=round(m["FETCH_SIZE"]), WRITE_SIZE_KiB=round(m["WRITE_SIZE"]), traffic_bytes_per_launch=traffic)
            lines.append(f"  HBM traffic per launch       {traffic / 1e6:.1f} MB  (2 x FETCH_SIZE + WRITE_SIZE)"
                         + (f" = {traffic / l['alg_bytes']:.2f} x algorithmic {l['alg_bytes'] / 1e6:.1f} MB; {traffic / us / 1e3:.0f} GB/s on the HBM side"
                            if "alg_bytes" in l else ""))
        entries.append(ent)
    open(prefix + "_pmc.txt", "w").write("\n".join(lines) + "\n")
    json.dump({"method": "tools/pmc_workload.py under rocprofv3 --pmc (separate passes: SQ set a, SQ set b, FETCH_SIZE, WRITE_SIZE, TCC hit/miss; "
                         "--kernel-trace only, no other trace domain); means over all launches of the instantiation; FETCH_SIZE / WRITE_SIZE "
                         "in KiB, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B)",
               "entries": [e for e in entries if "traffic_bytes_per_launch" in e]}, open(prefix + "_traffic.json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
