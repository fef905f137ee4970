#!/bin/bash
# ON THE GPU BOX: the bf16 wire format of the table gradient against fp32 over the WHOLE 20,000-step schedule -- one rank in an
# nccl group (NSR_FORCE_SHARDED: the trainer's multi-GPU exchange with its real dtypes; the sum over ranks is not exercised)
out=/root/repo/gpurun_out/r05transport; mkdir -p $out
cd /root/repo
: > "$out/runs.jsonl"
for seed in 42 1 2; do
  for t in bf16 fp32; do
    NSR_FORCE_SHARDED=1 NSR_TRANSPORT=$t timeout 600 python tools/train_psnr.py --path fused --steps 20000 --seed $seed --test-views 16 2>> "$out/err.txt" | grep '^{' | tail -1 >> "$out/runs.jsonl"
  done
done
python - "$out/runs.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d["seed"], d.get("exchange"), round(d["test_psnr"], 3), round(d["train_seconds"], 2), d["final_train_loss"])
PY
