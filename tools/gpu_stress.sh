#!/bin/bash
# determinism stress of the shipped build: concurrent replays (every op result bit-compared) + the NMS stress beside the RPN conv
export PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/${1:-r03_al}; mkdir -p $O
timeout 600 python tools/inflight_stress.py 1000 3 > $O/inflight_stress.txt 2>&1; echo "inflight rc=$?"; tail -5 $O/inflight_stress.txt
timeout 600 python tools/nms_stress.py > $O/nms_stress.txt 2>&1; echo "nms rc=$?"; tail -5 $O/nms_stress.txt
