export PYTHONUNBUFFERED=1
timeout 300 python tools/gather_conv_probe.py 2>&1 | grep -v amdgpu
