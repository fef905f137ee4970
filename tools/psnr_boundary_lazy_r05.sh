#!/bin/bash
# ON THE GPU BOX: the model-interface entry with its default (non-synchronising) outputs on the seeds the fused / modular tiers have
out=/root/repo/gpurun_out/r05psnr3; mkdir -p $out
cd /root/repo
: > "$out/psnr_runs.jsonl"
for seed in 3 4 5 6 7 8 9 10 11; do
  timeout 600 python tools/train_psnr.py --path boundary --steps 20000 --seed $seed --test-views 16 2>> "$out/psnr.err" | grep '^{' | tail -1 >> "$out/psnr_runs.jsonl"
done
python - "$out/psnr_runs.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d["seed"], round(d["test_psnr"], 3), round(d["train_seconds"], 2))
PY
