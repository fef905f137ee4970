cd $GRAFT_REPO_ROOT
for c in neus-blender neuralangelo; do
echo "default        $c $(python tools/neus_operating_point.py $c 100 2>/dev/null | tail -1 | cut -c1-70)"
export NSR_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/medium/libnsr_hip.so
echo "medium striped $c $(NSR_OWN_TUNE=0=2 python tools/neus_operating_point.py $c 100 4294967295 2>/dev/null | tail -1 | cut -c1-70)"
echo "medium claimed $c $(NSR_OWN_TUNE=0=3 python tools/neus_operating_point.py $c 100 4294967295 2>/dev/null | tail -1 | cut -c1-70)"
unset NSR_HIP_LIB
done
