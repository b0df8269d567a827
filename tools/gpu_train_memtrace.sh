#!/bin/bash
# kernel + memory-copy trace of the captured training step: what sits in the idle gaps between kernels (copy / memset nodes)
TAG=${1:-train_mem}
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof -- python $R/bench.py --profile-run --workload car.fhd.train --dtype bf16 --steps 12 --warmup 3 > $O/prof.log 2>&1
cd $R
python - <<PY
import csv, glob
d = "$O/prof"
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
print("files", mc, kt)
rows = list(csv.DictReader(open(mc[0]))) if mc else []
print("memory copies:", len(rows))
k = list(csv.DictReader(open(kt[0])))
# last k_vox_init start
starts = [int(r["Start_Timestamp"]) for r in k if "k_vox_init" in r["Kernel_Name"]]
t0 = starts[-2] if len(starts) > 1 else starts[-1]
t1 = starts[-1] if len(starts) > 1 else t0 + 4000000
print("step window", t0, t1)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s < t1:
        print("copy at %.1f us dur %.1f us" % ((s - t0) / 1e3, (e - s) / 1e3), r.get("Direction"), r.get("Bytes") or r.get("Size"))
PY
rm -rf $O/prof
