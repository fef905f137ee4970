"""average a rocprofv3 --pmc counter per kernel -> JSON.  With a regime file (bench.py, NSR_BENCH_REGIME_OUT) that carries
`roofline_dispatch_window` [a, b): over the dispatches a..b-1 of every kernel that is launched once per step (the steps
bench.py's roofline object is computed on); otherwise / for other kernels over the LAST `last` dispatches (default 90)."""
import csv, json, re, sys, collections
src, counter = sys.argv[1], sys.argv[2]
regime = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and sys.argv[3].endswith(".json") else {}
last = 90
win = regime.get("roofline_dispatch_window")
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
    if win and win[1] <= len(v) <= win[1] + 80 and k.startswith(("k_own_bin", "k_grid_backward_owner", "k_grid_reduce_slabs")):
        sel, how = [x for _, x in v[win[0]:win[1]]], "window"
    else:
        sel, how = [x for _, x in v[-last:]], "last"
    out[k] = {"avg": sum(sel) / len(sel), "dispatches": len(sel), "of": len(v), "selection": how}
print(json.dumps(out))
