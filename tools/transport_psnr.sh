#!/bin/bash
# bf16 vs fp32 wire format of the table gradient at world 2 (two ranks sharing this box's one GPU over gloo), same seed, same
# schedule: tools/train_psnr.py --path fused under torch.distributed.run.  usage: tools/transport_psnr.sh [steps] [out.jsonl] ["seeds"]
steps="${1:-3000}"; out="${2:-/root/repo/gpurun_out/transport_psnr.jsonl}"; seeds="${3:-42}"
cd /root/repo; : > "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29541
for seed in $seeds; do
for tr in bf16 fp32; do
  port=$((port + 1))
  NSR_TRANSPORT=$tr timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
    tools/train_psnr.py --path fused --steps "$steps" --seed "$seed" --test-views 8 --res 400 2> "/tmp/transport_$tr.err" | grep '^{' >> "$out" || tail -5 "/tmp/transport_$tr.err"
done
done
cat "$out"
