"""average a rocprofv3 --pmc counter per kernel over the LAST `last` dispatches of that kernel (default 90: the
post-timing profiling steps of bench.py, i.e. the regime its roofline object is computed in) -> JSON"""
import csv, json, re, sys, collections
src, counter = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 90
rows = collections.defaultdict(list)
for r in csv.DictReader(open(src)):
    if r.get("Counter_Name") != counter:
        continue
    m = re.search(r"(k_[a-z_0-9]+(?:<[^>(]*>)?)", r["Kernel_Name"])
    if not m:
        continue
    rows[m.group(1)].append((int(r.get("Dispatch_Id", 0) or 0), float(r["Counter_Value"])))
out = {}
for k, v in rows.items():
    v.sort()
    tail = [x for _, x in v[-last:]]
    out[k] = {"avg": sum(tail) / len(tail), "dispatches": len(tail), "of": len(v)}
print(json.dumps(out))
