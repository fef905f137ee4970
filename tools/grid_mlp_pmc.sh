#!/bin/bash
# ON THE GPU BOX: HBM-side fetch / write and L2 hit counters of the encode + MLP pair vs the fused kernel -> gpurun_out/grid_mlp_pmc.json
out=/root/repo/gpurun_out/grid_mlp_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for kind in coherent uniform; do
  for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
    for attempt in 1 2 3; do
      rm -rf /tmp/pg && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pg -o g -- python /root/repo/tools/grid_mlp_pmc.py $kind > /dev/null 2>&1
      f="$(find /tmp/pg -name '*counter_collection.csv' 2>/dev/null | head -1)"
      if [ -n "$f" ]; then python /root/repo/tools/pmc_summary.py "$f" $c > "$out/${kind}_$c.json" && break; fi
    done
  done
done
python - $out <<'PY'
import json, glob, os, sys
res = {"_what": "per-dispatch averages, 262,144 samples, nerf-blender density network, inference; FETCH/WRITE in KiB (FETCH as counted: "
                "x2 for wide coalesced reads per MI355X_MICROARCH.md), TCC_* in requests"}
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    kind, c = os.path.basename(f)[:-5].split("_", 1)
    d = json.load(open(f))
    for k, v in d.items():
        if k.startswith(("k_grid_forward", "k_mlp_forward", "k_grid_mlp_forward")):
            res.setdefault(kind, {}).setdefault(k, {})[c] = round(v["avg"], 1)
json.dump(res, open("/root/repo/gpurun_out/grid_mlp_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
