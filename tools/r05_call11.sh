#!/bin/bash
set -u
out=/root/repo/gpurun_out/r05n; mkdir -p "$out"
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_round5.py tests/test_gpu_resume.py tests/test_gpu_two_ranks.py -x -q 2>&1 | tail -5
timeout 600 python tools/host_breakdown.py 600 400 2>/dev/null | tail -1 > "$out/host_breakdown.json"; python -c "
import json; d=json.load(open('$out/host_breakdown.json'))
print('host', d['host_us_per_step'], 'inside', d['inside_wrapped_calls_us'], 'python', d['python_remainder_us'])
for c in d['calls'][:8]: print(c)"
timeout 600 python tools/forms_ab_debug.py 800 1 1 900 2>/dev/null | tail -1
NSR_VARIANTS=round4_forms,round5_forms timeout 600 python tools/step_variants.py 2500 320 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['settings']
for k,v in d.items(): print(k, v['mean_ms'], v['ms_per_step'], v['host_ms_per_step'], v['kept_per_step'])"
