#!/bin/bash
cd /tmp && export TMPDIR=/tmp
python /root/repo/tools/bg_refresh_profile.py 300 2>/dev/null | tail -1
NSR_NEUS_TORCH_REFRESH=1 python /root/repo/tools/bg_refresh_profile.py 300 2>/dev/null | tail -1
rm -rf /tmp/pb && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python /root/repo/tools/bg_refresh_profile.py 40 > /dev/null 2>&1
python - "$(find /tmp/pb -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms (40 steps + 7 refreshes):", tot / 1e6)
for r in rows[:22]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    print(f"{n[:70]:70s} {int(r['Calls']):4d} {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
