#!/bin/bash
# The whole -m gpu suite N times on one box (the driver's own command), to catch order- or timing-dependent failures; then smoke().
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r06_rep}; mkdir -p $O
for i in $(seq 1 ${2:-3}); do
  timeout 1200 python -m pytest tests/ -x -q -m gpu > $O/pytest_$i.log 2>&1; echo "run $i rc=$? $(tail -1 $O/pytest_$i.log)"
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
