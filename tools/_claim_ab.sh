# scratch: claimed placement A/B
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_hashgrid.py -x -q -m gpu 2>&1 | tail -3
for t in 0=4 0=2 0=4; do
  export NSR_OWN_TUNE=$t
  for c in neuralangelo neus-dtu neus-blender; do
    echo "$t $c $(timeout 300 python tools/neus_operating_point.py $c 100 2>/dev/null | python3 -c "import json,sys;d=json.load(sys.stdin);print(d.get('ms_per_step'))")"
  done
  echo "$t nerf $(timeout 300 python bench.py --steps 200 --warmup 20 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['steady_state']['ms_per_step'])")"
done
