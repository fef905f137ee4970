# usage: bash tools/_neus_prof.sh <tag> <configs...>   -> gpurun_out/<tag>_neus_op_<config>.json + _kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
for c in "$@"; do
  python tools/neus_operating_point.py $c 100 > gpurun_out/${tag}_neus_op_$c.json 2>gpurun_out/${tag}_neus_op_$c.err
  cat gpurun_out/${tag}_neus_op_$c.json
  rm -rf /tmp/pn && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn -o k -- python tools/neus_operating_point.py $c 60 > /dev/null 2>&1
  f=$(find /tmp/pn -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/${tag}_neus_op_${c}_kernel_stats.csv
  python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    n=n[:n.find("(")] if "(" in n else n
    print(f"  {n[:64]:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
P
done
