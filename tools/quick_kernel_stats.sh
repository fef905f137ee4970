#!/bin/bash
# ON THE GPU BOX: rocprofv3 kernel averages of a lean bench run (top 16)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq && NSR_BENCH_NO_STEADY=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o q -- python /root/repo/bench.py --no-cpu-baseline --no-other-workloads --no-boundary-path > /dev/null 2>&1
python - "$(find /tmp/pq -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:18]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    print(f"{n[:46]:46s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY
