#!/bin/bash
# ON THE GPU BOX: fork-bubble microbenchmark, the model-entry step measured directly / as bench.py's child, PSNR of the lazy model entry
set -u
out=/root/repo/gpurun_out/r05n; mkdir -p "$out"
cd /root/repo
[ -x build/tmp/fork_bubble ] || { mkdir -p build/tmp; hipcc --offload-arch=gfx950 -O2 tools/fork_bubble.hip -o build/tmp/fork_bubble 2>/dev/null; }
timeout 120 build/tmp/fork_bubble 300 > "$out/fork_bubble.txt" 2>&1; cat "$out/fork_bubble.txt"
python - > "$out/boundary_direct.json" 2> "$out/boundary_direct.err" <<'PY'
import json, sys, torch
sys.argv = ['bench.py']
sys.path.insert(0, '/root/repo')
import bench
print(json.dumps(bench.boundary_path(torch.device('cuda', 0))))
PY
echo direct; tail -c 400 "$out/boundary_direct.json"; echo
python - > "$out/boundary_child_of_gpu_parent.json" 2> "$out/boundary_child_of_gpu_parent.err" <<'PY'
import json, sys, torch
sys.argv = ['bench.py']
sys.path.insert(0, '/root/repo')
x = torch.zeros(1 << 28, device='cuda'); torch.cuda.synchronize()
import bench
print(json.dumps(bench.side_measurement('boundary_path')))
PY
echo child of a parent with a GPU context; tail -c 400 "$out/boundary_child_of_gpu_parent.json"; echo
python - > "$out/boundary_child_of_cpu_parent.json" 2> "$out/boundary_child_of_cpu_parent.err" <<'PY'
import json, sys
sys.argv = ['bench.py']
sys.path.insert(0, '/root/repo')
import bench
print(json.dumps(bench.side_measurement('boundary_path')))
PY
echo child of a parent without a GPU context; tail -c 400 "$out/boundary_child_of_cpu_parent.json"; echo
: > "$out/psnr_boundary_lazy.jsonl"
for seed in 42 1 2; do
  timeout 600 python tools/train_psnr.py --path boundary --steps 20000 --seed $seed --test-views 16 2>> "$out/psnr.err" | tail -1 >> "$out/psnr_boundary_lazy.jsonl"
done
cat "$out/psnr_boundary_lazy.jsonl" | cut -c1-400
