#!/bin/bash
# table-backward A/B on the GPU box: parity tests, isolated timings per knob setting, isolated PMC traffic, lean bench per placement
# usage: tools/r04_tb.sh <tag> [bench placements, e.g. "2 0 r03"]
set -u
tag="${1:-r04b}"; places="${2:-2 1 0 r03}"
out=/root/repo/gpurun_out/$tag; mkdir -p "$out"
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_hashgrid.py -x -q 2>&1 | tail -15 > "$out/pytest_hashgrid.txt"; tail -2 "$out/pytest_hashgrid.txt"
timeout 900 python tools/table_backward_variants.py build/variants/libnsr_hip_r03.so instant-nsr-pl_amd/nsr_hip/libnsr_hip.so > "$out/tb_variants.jsonl" 2> "$out/tb_variants.err"
python - "$out/tb_variants.jsonl" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if "error" in d: print(d); continue
    print(d["lib"][-14:], d.get("setting"), {k.split("_")[0][:5] + k.split(":")[1]: (v["bin_us"], v["accumulate_us"], v["accumulate_adam_us"]) for k, v in d.items() if ":" in k})
PY
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for attempt in 1 2 3; do
    rm -rf /tmp/pt && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pt -o t -- python /root/repo/tools/table_backward_pmc.py > /dev/null 2>&1
    f="$(find /tmp/pt -name '*counter_collection.csv' 2>/dev/null | head -1)"
    if [ -n "$f" ]; then cp "$f" "/tmp/tb_$c.csv"; break; fi
  done
done
python /root/repo/tools/table_backward_pmc_summary.py /tmp/tb_FETCH_SIZE.csv /tmp/tb_WRITE_SIZE.csv > "$out/tb_pmc_isolated.json" 2> "$out/tb_pmc_isolated.err"
python - "$out/tb_pmc_isolated.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k, "total", round(v["total_MB"]), "alg", round(v["algorithmic_MB"]), "ratio", round(v["ratio"], 3), {kk[:12]: (round(vv["fetch_MB"]), round(vv["write_MB"])) for kk, vv in v.items() if isinstance(vv, dict)})
PY
cd /root/repo
LEAN="--no-cpu-baseline --no-other-workloads --no-boundary-path"
for v in $places; do
  case $v in
    r03) envs="NSR_HIP_LIB=/root/repo/build/variants/libnsr_hip_r03.so";;
    *) envs="NSR_OWN_TUNE=0=$v";;
  esac
  env $envs timeout 600 python bench.py --steps 200 --warmup 20 $LEAN > "$out/bench_p${v}.json" 2> "$out/bench_p${v}.err"
done
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["kernels"]
        print(os.path.basename(f), "ms/step", round(d["ms_per_step"], 4), "steady", d["steady_state"] and round(d["steady_state"]["ms_per_step"], 4),
              "host", round(d["host_enqueue_ms_per_step"], 3), "roof", round(d["roofline"]["avg_launch_us"], 1), d["roofline"]["frac"] and round(d["roofline"]["frac"], 3),
              "bin", round(k["hashgrid_backward_bin"]["avg_us"], 1), "owner", round(k["hashgrid_backward_params"]["avg_us"], 1), "loss", round(d["final_loss"], 5),
              "transient", d.get("transient") and round(d["transient"]["ms_per_step"], 4))
    except Exception as e:
        print(os.path.basename(f), "??", repr(e)[:200])
PY
