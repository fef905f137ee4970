#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_resume.py -x -q 2>&1 | tail -5
NSR_LATE_IMAGES=8 timeout 200 python tools/late_regime.py 600 120 2>/dev/null | tail -1
NSR_ASYNC_PYTHON_STEP=1 NSR_LATE_IMAGES=8 timeout 200 python tools/late_regime.py 600 120 2>/dev/null | tail -1
