cd /root/repo
timeout 600 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_fused.py -x -q 2>&1 | tail -2
NSR_VARIANT_DATA=build/step_inputs.pt timeout 900 python tools/table_backward_variants.py instant-nsr-pl_amd/nsr_hip/libnsr_hip.so > gpurun_out/tb_real7.jsonl 2> gpurun_out/tb_real7.err
python - gpurun_out/tb_real7.jsonl <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if "error" in d: print(d); continue
    print(d["lib"][-14:], d.get("setting"), {k.split("_")[0][:5] + k.split(":")[1]: (v["bin_us"], v["accumulate_us"], v["accumulate_adam_us"]) for k, v in d.items() if ":" in k})
PY
tail -3 gpurun_out/tb_real7.err
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for attempt in 1 2 3; do
    rm -rf /tmp/pt && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pt -o t -- python /root/repo/tools/table_backward_pmc.py > /dev/null 2>&1
    f="$(find /tmp/pt -name '*counter_collection.csv' 2>/dev/null | head -1)"
    if [ -n "$f" ]; then cp "$f" "/tmp/tb_$c.csv"; break; fi
  done
done
python /root/repo/tools/table_backward_pmc_summary.py /tmp/tb_FETCH_SIZE.csv /tmp/tb_WRITE_SIZE.csv > /root/repo/gpurun_out/tb_pmc_isolated2.json
python - /root/repo/gpurun_out/tb_pmc_isolated2.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k, "total", round(v["total_MB"]), "alg", round(v["algorithmic_MB"]), "ratio", round(v["ratio"], 3), {kk[:12]: (round(vv["fetch_MB"]), round(vv["write_MB"])) for kk, vv in v.items() if isinstance(vv, dict)})
PY
cd /root/repo
LEAN="--no-cpu-baseline --no-other-workloads --no-boundary-path"
for v in dealt striped; do
  case $v in dealt) envs="A=1";; striped) envs="NSR_OWN_TUNE=0=2";; esac
  env $envs timeout 600 python bench.py --steps 200 --warmup 20 $LEAN > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err
done
python - <<'PY'
import json
for f in ("dealt","striped"):
    try:
        d=json.loads(open(f"gpurun_out/bench_{f}.json").read().strip().splitlines()[-1])
        print(f, "ms", round(d["ms_per_step"],4), "steady", round(d["steady_state"]["ms_per_step"],4), "host", round(d["host_enqueue_ms_per_step"],3), "loss", round(d["final_loss"],5), {k:round(v["avg_us"],1) for k,v in d["kernels"].items() if k.startswith("hashgrid")}, "value %.3g" % d["value"])
    except Exception as e:
        print(f, "??", repr(e)[:300]); print(open(f"gpurun_out/bench_{f}.err").read()[-800:])
PY
