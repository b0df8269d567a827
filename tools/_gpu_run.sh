export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/g
O=$PWD/gpurun_out/g
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 900 python tools/inflight_stress.py 1000 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_bs8_inflight3.json 2>/dev/null
timeout 600 python bench.py --inflight 1 --no-cpu-baseline > $O/bench_bs8_inflight1.json 2>/dev/null
timeout 600 python bench.py --workload nusc.fhd --dtype fp16 --no-cpu-baseline > $O/bench_nusc.fhd.json 2>/dev/null
timeout 600 python bench.py --workload nusc.pp --no-cpu-baseline > $O/bench_nusc.pp.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof3 -- python /root/repo/bench.py --steps 50 --warmup 10 --no-kernel-table --no-cpu-baseline > $O/prof3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof1 -- python /root/repo/bench.py --steps 50 --warmup 10 --inflight 1 --no-kernel-table --no-cpu-baseline > $O/prof1.log 2>&1
cd /root/repo
for n in 1 3; do db=$(find $O/prof$n -name "*.db" | head -1); python tools/rocprof_summary.py $db --steps 60 > $O/kernel_stats_inflight$n.txt 2>&1; done
rm -rf $O/prof1 $O/prof3
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], d['config'].get('single_step_latency_ms'), d.get('roofline',{}).get('frac'), d.get('roofline_mfma',{}).get('frac'), d.get('cpu_baseline',{}).get('sample','')[:40])"; done
