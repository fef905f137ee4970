#!/bin/bash
# copy the judged summaries of a tools/collect_profiles.sh run from gpurun_out/<tag>/ into profiles/ (tracked), prefixed r03_
set -e
tag="${1:-r03_final}"; src="gpurun_out/$tag"; dst="profiles"
for f in bench_w300_s200.json bench_w5_s20.json bench_under_rocprof_w300_s200.json bench_under_rocprof_w5_s20.json \
         kernel_stats_w300_s200.csv kernel_stats_w5_s20.csv timeline_tail_w300_s200.csv timeline_tail_w5_s20.csv pmc_traffic.json \
         fused_neus-blender_kernel_stats.csv fused_neus-dtu_kernel_stats.csv fused_neuralangelo_kernel_stats.csv \
         neus_step_neus-blender.json neus_step_neus-dtu.json neus_step_neuralangelo.json boundary_phases.json \
         boundary_timeline_summary.txt microbench.json table_backward_isolated.json \
         neus_op_neus-blender.json neus_op_neus-dtu.json neus_op_neuralangelo.json neus_op_neus-blender_kernel_stats.csv \
         neus_op_neus-dtu_kernel_stats.csv neus_op_neuralangelo_kernel_stats.csv bg_refresh.json vmlp_layout_bench.json; do
  [ -f "$src/$f" ] && cp "$src/$f" "$dst/r03_$f"
done
head -c 300000 "$src/boundary_timeline_tail.csv" > "$dst/r03_boundary_timeline_tail.csv" 2>/dev/null || true
ls -la $dst | grep r03_
