#!/bin/bash
# ON THE GPU BOX (round 5): BASELINE.json's PSNR leg with a standard error -- the three tiers of the path trained on the same
# scene / schedule for 20,000 steps over FIVE seeds, PSNR on SIXTEEN unseen 400x400 views -> gpurun_out/<tag>/psnr_paths.json
tag="${1:-r05psnr}"; steps="${2:-20000}"; out=/root/repo/gpurun_out/$tag; mkdir -p $out
cd /root/repo
: > "$out/psnr_runs.jsonl"
for seed in 42 1 2 3 4; do
  for p in fused boundary modular; do
    timeout 1500 python tools/train_psnr.py --path $p --steps $steps --seed $seed --test-views 16 2>> "$out/psnr.err" | tail -1 >> "$out/psnr_runs.jsonl"
  done
done
python - "$out" <<'PY'
import json, sys, collections, math
out = sys.argv[1]
runs = [json.loads(l) for l in open(f"{out}/psnr_runs.jsonl") if l.strip().startswith("{")]
by = collections.defaultdict(list)
for r in runs:
    by[r["path"]].append(r)
mean = lambda v: sum(v) / len(v)
def se(v):
    m = mean(v)
    return math.sqrt(sum((x - m) ** 2 for x in v) / max(len(v) - 1, 1) / len(v))
table = {}
for p, rs in by.items():
    ps = [r["test_psnr"] for r in rs]
    table[p] = {"seeds": [r["seed"] for r in rs], "test_psnr_mean": mean(ps), "test_psnr_se": se(ps),
                "test_psnr_per_seed": {r["seed"]: round(r["test_psnr"], 3) for r in rs},
                "train_seconds_mean": mean([r["train_seconds"] for r in rs]), "samples_per_sec_mean": mean([r["samples_per_sec"] for r in rs])}
if "modular" in by:
    ref = {r["seed"]: r["test_psnr"] for r in by["modular"]}
    for p, rs in by.items():
        d = [r["test_psnr"] - ref[r["seed"]] for r in rs if r["seed"] in ref]  # paired by seed (same scene, same test views)
        table[p]["delta_vs_modular_dB"] = mean(d)
        table[p]["delta_vs_modular_se"] = se(d) if len(d) > 1 else None
        table[p]["within_0.1_dB_at_2_se"] = bool(abs(mean(d)) + 2 * (se(d) if len(d) > 1 else 0.0) <= 0.1) if p != "modular" else True
json.dump({"_what": "tools/psnr_paths_r05.sh: nerf-blender config, procedural scene (100 training views 400x400), the reference's 20,000-step "
                    "schedule, PSNR on 16 unseen 400x400 views, seeds 42 / 1 / 2 / 3 / 4; fused = the asynchronous trainer bench.py times, "
                    "boundary = nsr.models.FusedNeRFModel behind the reference's model interface with the system's own statements, "
                    "modular = the reference's model statements on the drop-in packages under autocast + GradScaler(65536) + torch AdamW "
                    "(the reference-semantics path); deltas are paired by seed; SE = standard error over the seeds",
           "table": table, "runs": runs}, open(f"{out}/psnr_paths.json", "w"), indent=1)
print(json.dumps(table, indent=1))
PY
