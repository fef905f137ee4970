cd /root/repo
NSR_VARIANT_SETTINGS=default NSR_VARIANT_DATA=build/step_inputs.pt timeout 900 python tools/table_backward_variants.py instant-nsr-pl_amd/nsr_hip/libnsr_hip.so build/variants/libnsr_hip_l10.so build/variants/libnsr_hip_l12.so build/variants/libnsr_hip_l10b128.so build/variants/libnsr_hip_l11b512.so > gpurun_out/tb_real8.jsonl 2> gpurun_out/tb_real8.err
python - gpurun_out/tb_real8.jsonl <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if "error" in d: print(d); continue
    print(d["lib"][-18:], d.get("setting"), {k.split("_")[0][:5] + k.split(":")[1]: (v["bin_us"], v["accumulate_us"], v["accumulate_adam_us"]) for k, v in d.items() if ":" in k})
PY
tail -2 gpurun_out/tb_real8.err
LEAN="--no-cpu-baseline --no-other-workloads --no-boundary-path"
for v in base l10 l12; do
  case $v in base) envs="A=1";; *) envs="NSR_HIP_LIB=/root/repo/build/variants/libnsr_hip_$v.so";; esac
  env $envs NSR_BENCH_NO_STEADY=1 timeout 600 python bench.py --steps 200 --warmup 20 $LEAN 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k.startswith('hashgrid')})"
done
