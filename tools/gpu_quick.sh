#!/bin/bash
# gpurun -- 'bash tools/gpu_quick.sh <tag> <pytest args...>': a subset of the -m gpu suite, log under gpurun_out/<tag>/
TAG=$1; shift
export PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest -m gpu -q -x "$@" > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "^E   \|amdgpu.ids" $O/pytest.log | tail -60
