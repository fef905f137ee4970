#!/bin/bash
# ON THE GPU BOX: fused NeuS steps with the row-major (0) and the tile-major (2) encoding layout, same box
out=/root/repo/gpurun_out/ab_enc_layout.jsonl; : > $out
for cfg in neus-blender neus-dtu neuralangelo; do
  for lay in 0 2 0 2; do
    NSR_NEUS_ENC_LAYOUT=$lay python /root/repo/tools/neus_step_bench.py --config $cfg --rays 4096 --steps 40 --warmup 60 2>/dev/null | tail -1 | sed "s/^{/{\"layout\": $lay, /" >> $out
  done
done
cat $out | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], 'layout', d['layout'], round(d['ms_per_step'], 3), 'ms', int(d['samples_per_step']))
"
