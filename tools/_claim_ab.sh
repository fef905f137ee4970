cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_neuralangelo.py -x -q -m gpu 2>&1 | tail -1
for k in 1 2; do
  for c in neuralangelo neus-dtu neus-blender; do
    echo "$c $(timeout 300 python tools/neus_operating_point.py $c 100 2>/dev/null | tail -1 | cut -c1-70)"
  done
done
