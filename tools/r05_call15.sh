#!/bin/bash
# ON THE GPU BOX: fork-bubble microbenchmark (second form), the model-entry step: wgrad cap A/B, GPU busy time per step, timeline
set -u
out=/root/repo/gpurun_out/r05o; mkdir -p "$out"
cd /root/repo
timeout 120 build/tmp/fork_bubble 300 > "$out/fork_bubble.txt" 2>&1; cat "$out/fork_bubble.txt"
run_bp() {  # $1 = tag
  python - > "$out/boundary_$1.json" 2> "$out/boundary_$1.err" <<'PY'
import json, sys, torch
sys.argv = ['bench.py']
sys.path.insert(0, '/root/repo')
import bench
print(json.dumps(bench.boundary_path(torch.device('cuda', 0))))
PY
  python - "$out/boundary_$1.json" "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[2], d["ms_per_step"], d["samples_per_sec"])
PY
}
run_bp warmup_process
run_bp default
NSR_WGRAD_MAX_BLOCKS=128 run_bp wgrad_cap_128
NSR_BOUNDARY_EAGER=1 run_bp eager
run_bp default_again
python tools/boundary_profile.py > "$out/boundary_phases.json" 2>/dev/null; cat "$out/boundary_phases.json"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && NSR_BP_NOSYNC=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python /root/repo/tools/boundary_profile.py > /dev/null 2>&1
cp "$(find /tmp/pk -name '*kernel_stats.csv' | head -1)" "$out/boundary_kernel_stats.csv"
python /root/repo/tools/trace_tail.py "$(find /tmp/pk -name '*kernel_trace.csv' | head -1)" "$out/boundary_timeline_tail.csv" 4000
head -30 "$out/boundary_kernel_stats.csv" | cut -c1-150
