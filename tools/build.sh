#!/bin/bash
# Build libsecond_hip.so and fail loudly (use as: tools/build.sh && gpurun ...)
set -e
cd "$(dirname "$0")/.."
python second.pytorch_amd/build.py > /tmp/sec_build.log 2>&1 || { grep -E "error" -A6 /tmp/sec_build.log | head -40; echo BUILD FAILED; exit 1; }
echo "build ok: $(ls -la second.pytorch_amd/lib/libsecond_hip.so | awk '{print $5, $6, $7, $8}')"
