# scratch: k_own_bin per variant on neuralangelo (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in default "$@"; do
  if [ $v = default ]; then unset NSR_HIP_LIB; else export NSR_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/$v/libnsr_hip.so; fi
  for c in neuralangelo neus-blender; do
  rm -rf /tmp/pn && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn -o k -- python tools/neus_operating_point.py $c 60 > /tmp/op.json 2>/dev/null
  f=$(find /tmp/pn -name "*kernel_stats.csv" | head -1)
  echo "== $v $c $(python3 -c "import json;d=json.load(open('/tmp/op.json'));print(d.get('ms_per_step'))")"
  python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    n=n[:n.find("(")] if "(" in n else n
    if "own" in n or "tap" in n: print(f"  {n[:64]:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
P
  done
done
