cd /root/repo
mkdir -p gpurun_out/r04
timeout 1500 python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err; tail -c 400 gpurun_out/r04/bench_default.json; echo
bash tools/secondary_pmc.sh r04 2>&1 | tail -40
