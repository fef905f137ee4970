#!/bin/bash
# usage: tools/timeline_tail.sh <out.csv> <n_dispatches> -- <command...> : rocprofv3 kernel trace of the command, reduced to
# the last n dispatches (start_us,end_us,queue,kernel) + per-kernel totals of that tail on stdout
out="$1"; n="$2"; shift; shift; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tt && rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o tt -- "$@" > /tmp/tt.log 2>&1
f="$(find /tmp/tt -name '*kernel_trace.csv' | head -1)"
python /root/repo/tools/trace_tail.py "$f" "$out" "$n"
python - "$out" <<'PY'
import csv, sys, collections
rows = [(r[0], r[1], r[2], ",".join(r[3:])) for r in csv.reader(open(sys.argv[1]))]
tot = collections.defaultdict(lambda: [0.0, 0])
for s, e, q, k in rows:
    tot[k][0] += float(e) - float(s); tot[k][1] += 1
span = float(rows[-1][1]) - float(rows[0][0])
busy = sum(v[0] for v in tot.values())
print(f"span {span:.0f} us, sum of kernel durations {busy:.0f} us ({100*busy/span:.0f} %), {len(rows)} dispatches")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{k[:48]:48s} n {v[1]:5d} total {v[0]:9.1f} us avg {v[0]/v[1]:7.1f}")
PY
