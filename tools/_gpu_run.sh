export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 900 python tools/inflight_stress.py 2000 2>&1 | tail -1
