#!/bin/bash
# copy the judged summaries of a tools/collect_profiles_r06.sh run from gpurun_out/<tag>/ into profiles/ (tracked), prefixed r06_
set -e
tag="${1:-r06_final}"; src="gpurun_out/$tag"; dst="profiles"
for f in bench_w20_s200.json bench_w5_s20.json bench_under_rocprof_w20_s200.json bench_under_rocprof_w5_s20.json \
         kernel_stats_w20_s200.csv kernel_stats_w5_s20.csv pmc_traffic.json \
         neus_op_neus-blender.json neus_op_neus-dtu.json neus_op_neuralangelo.json neus_op_neus-blender_kernel_stats.csv \
         neus_op_neus-dtu_kernel_stats.csv neus_op_neuralangelo_kernel_stats.csv microbench.json \
         step_variants_2500.json step_variants_450.json late_regime.json fetch_calibration.json secondary_pmc.json vmlp_bench.json small_kernels.json vmlp_pmc_SQ_VALU_MFMA_BUSY_CYCLES.json vmlp_pmc_SQ_WAVE_CYCLES.json; do
  [ -f "$src/$f" ] && cp "$src/$f" "$dst/r06_$f"
done
# timeline: three steps of the timed region (the tail of the trace is the instrumented part of the run)
python - "$src" "$dst" <<'PY'
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
for reg in ("w20_s200", "w5_s20"):
    try:
        rows = [(float(r[0]), float(r[1]), r[2], r[3]) for r in csv.reader(open(f"{src}/timeline_tail_{reg}.csv"))]
    except FileNotFoundError:
        continue
    # steps with the optimizer fused into the table backward: the scheduled AdamW launch covers the MLP weights only (< 35 us)
    idx = [i for i, r in enumerate(rows) if r[3] == "k_adamw_scheduled" and r[1] - r[0] < 35]
    if len(idx) < 8:
        continue
    mid = len(idx) // 2
    a, b = idx[mid], idx[mid + 3]
    with open(f"{dst}/r06_timeline_3steps_{reg}.csv", "w") as out:
        out.write("start_us,duration_us,queue,kernel\n")
        for r in rows[a:b + 1]:
            out.write("%.1f,%.1f,%s,%s\n" % (r[0] - rows[a][0], r[1] - r[0], r[2], r[3]))
    per = sorted(rows[idx[i + 1]][0] - rows[idx[i]][0] for i in range(len(idx) - 1))
    print(reg, "fused steps in the tail:", len(idx), "median period us:", per[len(per) // 2])
PY
ls -la $dst | grep r06_
