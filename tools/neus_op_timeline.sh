#!/bin/bash
# ON THE GPU BOX: kernel timeline summary of a NeuS workload at the operating point.  usage: neus_op_timeline.sh <config> ...
for c in "$@"; do
  echo "=== $c"
  bash /root/repo/tools/timeline_tail.sh /root/repo/gpurun_out/neus_op_timeline_$c.csv 3000 -- python /root/repo/tools/neus_operating_point.py $c 60 | head -34
done
