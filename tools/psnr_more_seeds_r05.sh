#!/bin/bash
# ON THE GPU BOX: seven further seeds of the fused / modular PSNR tiers (profiles/r05_psnr_paths.json has seeds 42, 1..4)
out=/root/repo/gpurun_out/r05psnr2; mkdir -p $out
cd /root/repo
: > "$out/psnr_runs.jsonl"
for seed in 5 6 7 8 9 10 11; do
  for p in fused modular; do
    timeout 600 python tools/train_psnr.py --path $p --steps 20000 --seed $seed --test-views 16 2>> "$out/psnr.err" | tail -1 >> "$out/psnr_runs.jsonl"
  done
done
cut -c1-200 "$out/psnr_runs.jsonl"
