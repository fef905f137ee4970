#!/bin/bash
out=/root/repo/gpurun_out/vmlp_pmc2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM; do
  rm -rf /tmp/pv && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pv -o v -- python /root/repo/tools/vmlp_layout_bench.py > /dev/null 2>&1
  f="$(find /tmp/pv -name '*counter_collection.csv' 2>/dev/null | head -1)"
  if [ -n "$f" ]; then python /root/repo/tools/pmc_summary.py "$f" $c > "$out/$c.json"; else echo "no data for $c"; fi
done
python - $out <<'PY'
import json, glob, os, sys
res = {}
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    c = os.path.basename(f)[:-5]
    try:
        for k, v in json.load(open(f)).items():
            if k.startswith("k_vmlp_backward") or k.startswith("k_vmlp_forward"):
                res.setdefault(k, {})[c] = v["avg"]
    except Exception as e:
        print(c, "??", e)
json.dump(res, open("/root/repo/gpurun_out/vmlp_pmc2.json", "w"), indent=1)
for k, v in res.items():
    w = v.get("SQ_WAVE_CYCLES", 1)
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:26s} {x:16.0f}  per wave-quad-cycle {x / w:8.4f}")
PY
