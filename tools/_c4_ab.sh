cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fused_neus_bg.py tests/test_gpu_models_entry.py tests/test_gpu_fused_neus.py -x -q -m gpu 2>&1 | tail -2
for k in 1 2 3; do
echo "$(timeout 300 python tools/neus_operating_point.py neus-dtu 100 2>/dev/null | tail -1 | cut -c1-180)"
done
