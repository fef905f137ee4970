cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_neuralangelo.py -x -q -m gpu 2>&1 | tail -2
for c in neuralangelo; do
  for k in 1 2; do echo "$c $(timeout 300 python tools/neus_operating_point.py $c 100 2>/dev/null | python3 -c "import json,sys;d=json.load(sys.stdin);print(d.get('ms_per_step'))")"; done
  rm -rf /tmp/pn && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn -o k -- python tools/neus_operating_point.py $c 60 > /tmp/op.json 2>/dev/null
  f=$(find /tmp/pn -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    n=n[:n.find("(")] if "(" in n else n
    print(f"  {n[:64]:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
P
done
