#!/bin/bash
# ON THE GPU BOX: BASELINE.json's PSNR leg -- the three tiers of the path trained on the same scene / schedule for N steps over
# several seeds, PSNR on unseen views -> gpurun_out/<tag>/psnr_paths.json   (tools/train_psnr.py)
tag="${1:-r04}"; steps="${2:-20000}"; out=/root/repo/gpurun_out/$tag; mkdir -p $out
cd /root/repo
: > "$out/psnr_runs.jsonl"
for seed in 42 1 2; do
  for p in fused boundary modular; do
    timeout 1500 python tools/train_psnr.py --path $p --steps $steps --seed $seed 2>> "$out/psnr.err" | tail -1 >> "$out/psnr_runs.jsonl"
  done
done
NSR_GRAD_SCALE=128 timeout 600 python tools/train_psnr.py --path fused --steps $steps --seed 42 2>> "$out/psnr.err" | tail -1 | sed 's/"path": "fused"/"path": "fused_grad_scale_128"/' >> "$out/psnr_runs.jsonl"
python - "$out" <<'PY'
import json, sys, collections
out = sys.argv[1]
runs = [json.loads(l) for l in open(f"{out}/psnr_runs.jsonl") if l.strip().startswith("{")]
by = collections.defaultdict(list)
for r in runs:
    by[r["path"]].append(r)
mean = lambda v: sum(v) / len(v)
ref = mean([r["test_psnr"] for r in by["modular"]]) if by.get("modular") else None
table = {p: {"test_psnr_mean": mean([r["test_psnr"] for r in rs]), "test_psnr_per_seed": {r["seed"]: round(r["test_psnr"], 3) for r in rs},
             "delta_vs_modular_dB": (mean([r["test_psnr"] for r in rs]) - ref) if ref else None,
             "train_seconds_mean": mean([r["train_seconds"] for r in rs]), "samples_per_sec_mean": mean([r["samples_per_sec"] for r in rs])}
         for p, rs in by.items()}
json.dump({"_what": "tools/psnr_paths.sh: nerf-blender config, procedural scene (100 views 400x400), reference schedule, PSNR on 4 unseen "
                    "views, seeds 42 / 1 / 2; modular = the reference's model statements on the drop-in packages under autocast + "
                    "GradScaler(65536) + torch AdamW (the reference-semantics path); fused_grad_scale_128 = the fused trainer with the "
                    "round-3 scale of dL/dy in its fp16 MLP backward (tcnn's 128 WITHOUT Lightning's GradScaler on top)",
           "table": table, "runs": runs}, open(f"{out}/psnr_paths.json", "w"), indent=1)
print(json.dumps(table, indent=1))
PY
