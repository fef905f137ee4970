"""average a rocprofv3 --pmc counter per kernel from *counter_collection.csv -> JSON lines"""
import csv, json, re, sys, collections
src, counter = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(src)):
    if r.get("Counter_Name") != counter:
        continue
    m = re.search(r"(k_[a-z_0-9]+(?:<[^>(]*>)?)", r["Kernel_Name"])
    if not m:
        continue
    a = acc[m.group(1)]
    a[0] += float(r["Counter_Value"]); a[1] += 1
print(json.dumps({k: {"avg": v[0] / v[1], "dispatches": v[1]} for k, v in acc.items()}))
