#!/bin/bash
# ON THE GPU BOX: final validation of round 5: build check, full GPU suite, smoke(), both bench command lines
set -u
out=/root/repo/gpurun_out/r05_final3; mkdir -p "$out"
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu.log" 2>&1; tail -3 "$out/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1; tail -2 "$out/smoke.log"
python bench.py > "$out/bench_w20_s200.json" 2> "$out/bench_w20_s200.stderr"; grep '^{' "$out/bench_w20_s200.json" | tail -1 | cut -c1-300
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_w5_s20.json" 2> "$out/bench_w5_s20.stderr"; grep '^{' "$out/bench_w5_s20.json" | tail -1 | cut -c1-300
