#!/bin/bash
# ON THE GPU BOX: last validation after the key-10 code went in (defaults unchanged): full GPU suite + the default bench line
set -u
out=/root/repo/gpurun_out/r05_final4; mkdir -p "$out"
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu.log" 2>&1; tail -3 "$out/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > "$out/bench_w20_s200.json" 2> "$out/bench_w20_s200.stderr"; grep '^{' "$out/bench_w20_s200.json" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['steady_state']['ms_per_step'], d['step_forms_ab']['round5_over_round4'], d['late_regime']['ms_per_step'], d['whole_run']['seconds'], d['roofline']['frac'], d['boundary_path']['ms_per_step'], d['boundary_path']['late']['ms_per_step'])"
