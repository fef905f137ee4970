"""print per-kernel averages of every counter in a rocprofv3 counter_collection.csv (kernels matching a substring)"""
import csv, re, sys, collections
src, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(src)):
    if pat not in r["Kernel_Name"]:
        continue
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", r["Kernel_Name"])
    a = acc[m.group(1) if m else r["Kernel_Name"][:40]][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k, {c: round(v[0] / v[1]) for c, v in d.items()}, "dispatches", max(v[1] for v in d.values()))
