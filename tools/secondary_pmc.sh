#!/bin/bash
# ON THE GPU BOX: PMC evidence for the hash gather (fabric fetch / write, L2 hit rate) and the fused fp16 MLP (MFMA busy share)
# -> gpurun_out/<tag>/secondary_pmc.json   (tools/secondary_pmc.py is the workload; one counter per pass)
tag="${1:-r04}"; out=/root/repo/gpurun_out/$tag/secondary; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for kind in E1 E2; do
  python /root/repo/tools/secondary_pmc.py $kind "$out/${kind}_times.json" > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16; do
    for attempt in 1 2 3; do
      rm -rf /tmp/ps && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/ps -o s -- python /root/repo/tools/secondary_pmc.py $kind > /dev/null 2>&1
      f="$(find /tmp/ps -name '*counter_collection.csv' 2>/dev/null | head -1)"
      if [ -n "$f" ]; then python /root/repo/tools/pmc_summary.py "$f" $c > "$out/${kind}_$c.json" && break; fi
    done
  done
done
python - $out /root/repo/gpurun_out/$tag/secondary_pmc.json <<'PY'
import json, glob, os, sys
src, dst = sys.argv[1], sys.argv[2]
res = {"_what": "tools/secondary_pmc.sh: per-dispatch averages at 2^18 samples (E1 uniform / E2 ray-coherent positions), nerf-blender "
                "density network; FETCH / WRITE in MB (FETCH x2 per MI355X_MICROARCH.md: the counter tallies 128-B requests at 64 B), "
                "L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS); MFMA: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES) = share of the "
                "cycles a wave is resident during which the SIMD's MFMA pipe works for it (SQ_WAVE_CYCLES counts in units of 4 cycles)"}
for kind in ("E1", "E2"):
    ent = {"times_us": json.load(open(f"{src}/{kind}_times.json")) if os.path.exists(f"{src}/{kind}_times.json") else None}
    per = {}
    for f in sorted(glob.glob(f"{src}/{kind}_*.json")):
        c = os.path.basename(f)[len(kind) + 1:-5]
        if c == "times":
            continue
        for k, v in json.load(open(f)).items():
            if k.startswith(("k_grid_forward", "k_mlp_", "k_grid_mlp_forward")):
                per.setdefault(k, {})[c] = v["avg"]
    for k, v in per.items():
        if "FETCH_SIZE" in v:
            v["fetch_MB"] = round(2 * 1024 * v["FETCH_SIZE"] / 1e6, 2)
        if "WRITE_SIZE" in v:
            v["write_MB"] = round(1024 * v["WRITE_SIZE"] / 1e6, 2)
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v and v["TCC_HIT_sum"] + v["TCC_MISS_sum"] > 0:
            v["l2_hit_rate"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("SQ_WAVE_CYCLES"):
            v["mfma_busy_per_wave_cycle"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * v["SQ_WAVE_CYCLES"]), 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("SQ_BUSY_CYCLES"):
            v["mfma_busy_over_sq_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CYCLES"], 4)
    ent["kernels"] = per
    res[kind] = ent
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps({k: {kk: {c: vv.get(c) for c in ("fetch_MB", "write_MB", "l2_hit_rate", "mfma_busy_per_wave_cycle", "mfma_busy_over_sq_busy")} for kk, vv in v["kernels"].items()} for k, v in res.items() if k in ("E1", "E2")}, indent=1))
PY
