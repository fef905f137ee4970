#!/bin/bash
# ON THE GPU BOX: per-kernel averages of a fused NeuS step with both encoding layouts
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
for lay in 0 2; do
  rm -rf /tmp/pe && NSR_NEUS_ENC_LAYOUT=$lay rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o e -- python /root/repo/tools/neus_step_bench.py --config $cfg --rays 4096 --steps 40 --warmup 60 > /dev/null 2>&1
  f="$(find /tmp/pe -name '*kernel_stats.csv' | head -1)"
  echo "== $cfg layout $lay"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    print(f"{n[:50]:50s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us")
PY
done; done
