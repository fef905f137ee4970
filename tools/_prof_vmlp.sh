cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ONLY_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_vmlp -o vmlp -- python tools/vmlp_split_bench.py > gpurun_out/prof_vmlp.log 2>&1
find gpurun_out/prof_vmlp -name "*kernel_stats*" | head
f=$(find gpurun_out/prof_vmlp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    print(f"{n[:100]:100s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
P
