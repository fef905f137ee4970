#!/bin/bash
# usage: tools/kstats.sh <pattern> -- <command...> : average duration of the LAST $LAST (default 100) dispatches of every
# kernel matching pattern (rocprofv3 --kernel-trace), so that warm-up / set-up launches of the command do not mix in
pat="$1"; shift; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -o ks -- "$@" > /tmp/ks.log 2>&1
python - "$pat" "${LAST:-100}" <<'PY'
import csv, glob, sys, re, collections
f = glob.glob("/tmp/ks/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    if re.search(sys.argv[1], r["Kernel_Name"]):
        by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    t = v[-int(sys.argv[2]):]
    print(f"{k[:72]:72s} n {len(v):5d} last-avg {sum(t)/len(t):8.1f} us")
PY
