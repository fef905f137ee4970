cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fused_neus_bg.py tests/test_gpu_occupancy.py tests/test_capi.py -x -q 2>&1 | tail -2
for k in 1 2 3; do
echo "$(timeout 300 python tools/neus_operating_point.py neus-dtu 100 2>/dev/null | tail -1 | cut -c1-180)"
done
