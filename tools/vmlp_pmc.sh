#!/bin/bash
# ON THE GPU BOX: MFMA-busy share of the fp32 MLP kernels (SDF network, 1,048,576 rows, all three encoding layouts)
out=/root/repo/gpurun_out/vmlp_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32; do
  for attempt in 1 2 3; do
    rm -rf /tmp/pv && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pv -o v -- python /root/repo/tools/vmlp_layout_bench.py > /dev/null 2>&1
    f="$(find /tmp/pv -name '*counter_collection.csv' 2>/dev/null | head -1)"
    if [ -n "$f" ]; then python /root/repo/tools/pmc_summary.py "$f" $c > "$out/$c.json" && break; fi
  done
done
python - $out <<'PY'
import json, glob, os, sys
res = {}
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    c = os.path.basename(f)[:-5]
    for k, v in json.load(open(f)).items():
        if k.startswith("k_vmlp"):
            res.setdefault(k, {})[c] = v["avg"]
for k, v in res.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v and v["SQ_BUSY_CYCLES"]:
        v["mfma_busy_over_sq_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CYCLES"], 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"]:
        v["mfma_busy_over_wave_cycles"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_WAVE_CYCLES"], 4)
json.dump(res, open("/root/repo/gpurun_out/vmlp_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
