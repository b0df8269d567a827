export PYTHONUNBUFFERED=1
run() { timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'roofline', d['roofline']['launch_us'], d['roofline']['frac'])
print([k['us'] for k in d['kernels'] if k['op']=='indice_conv'])"; }
echo "== shipped (rows kernel keeps packed fp32)"; run
echo "== old flags"; SEC_HIP_LIB=$PWD/second.pytorch_amd/lib/libsecond_hip_oldflags.so run
LANES=3 timeout 900 python tools/inflight_stress.py 3000 3 2>&1 | grep -v amdgpu.ids | tail -2
LOADKIND=conv LOAD=2 timeout 900 python tools/nms_stress.py 300 2>&1 | grep -v amdgpu.ids | tail -1
