#!/bin/bash
# ON THE GPU BOX: masked-loss kernels: tests, then the model-entry steps with / without them (same box)
set -u
out=/root/repo/gpurun_out/r05p; mkdir -p "$out"
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_round5.py -k "masked or neus_model_entry or lazy" -x -q 2>&1 | tail -15 > "$out/pytest.txt"; tail -8 "$out/pytest.txt"
run_bp() {  # $1 = function, $2 = tag
  python - "$1" > "$out/$1_$2.json" 2> "$out/$1_$2.err" <<'PY'
import json, sys, torch
fn = sys.argv[1]
sys.argv = ['bench.py']
sys.path.insert(0, '/root/repo')
import bench
print(json.dumps(getattr(bench, fn)(torch.device('cuda', 0))))
PY
  python - "$out/$1_$2.json" "$1 $2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], d["ms_per_step"], d["samples_per_sec"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run_bp boundary_path warmup_process
for rep in 1 2; do
  run_bp boundary_path fused_loss_$rep
  NSR_MASKED_LOSS_TORCH=1 run_bp boundary_path torch_loss_$rep
done
NSR_BOUNDARY_EAGER=1 run_bp boundary_path eager
run_bp boundary_path_neus fused_loss
NSR_BOUNDARY_EAGER=1 run_bp boundary_path_neus eager
run_bp boundary_path_neus fused_loss_2
python tools/masked_loss_bench.py > "$out/masked_loss_bench.json" 2>/dev/null; cat "$out/masked_loss_bench.json"
