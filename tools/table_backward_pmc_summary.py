"""rocprofv3 counter CSVs of tools/table_backward_pmc.py -> JSON: per (placement, samples) the average FETCH_SIZE / WRITE_SIZE
of the binning kernel, the owner kernel and the slab reduction, in MB per launch (read side x2 as MI355X_MICROARCH.md
prescribes for gfx950) next to the algorithmic bytes.   usage: ... <fetch.csv> <write.csv>"""
import csv, json, re, sys, collections
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from table_backward_pmc import CASES, REPS

N_TABLE = 12599920


def per_kernel(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        m = re.search(r"(k_own_bin|k_grid_backward_owner|k_grid_reduce_slabs)", r["Kernel_Name"])
        if m:
            rows[m.group(1)].append((int(r.get("Dispatch_Id", 0) or 0), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in rows.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for ci, (placement, n) in enumerate(CASES):
    ent = {}
    for k in ("k_own_bin", "k_grid_backward_owner", "k_grid_reduce_slabs"):
        per_case = {"k_own_bin": REPS, "k_grid_backward_owner": REPS, "k_grid_reduce_slabs": REPS}[k]
        f = fetch.get(k, [])[ci * per_case + 1:(ci + 1) * per_case]  # (first launch of a case: cold code / buffers)
        w = write.get(k, [])[ci * per_case + 1:(ci + 1) * per_case]
        if f and w:
            ent[k] = {"fetch_MB": 2 * 1024 * sum(f) / len(f) / 1e6, "write_MB": 1024 * sum(w) / len(w) / 1e6}
    tot = sum(v["fetch_MB"] + v["write_MB"] for v in ent.values())
    alg = (140 * n + 26 * N_TABLE) / 1e6
    out[f"{('dealt', 'listed', 'striped')[placement]}:{n}"] = dict(ent, total_MB=tot, algorithmic_MB=alg, ratio=tot / alg if alg else None)
print(json.dumps(out, indent=1))
