#!/bin/bash
set -u
out=/root/repo/gpurun_out/r05b; mkdir -p "$out"
cd /root/repo
for p in 0 1 2; do NSR_DENSE_PROBE=$p timeout 300 python tools/dense_levels_bench.py > "$out/dense_bench_probe$p.json" 2> "$out/dense_bench_probe$p.err"; cat "$out/dense_bench_probe$p.json"; tail -2 "$out/dense_bench_probe$p.err"; done
