cd $GRAFT_REPO_ROOT
for k in 1 2; do
for c in neus-blender neus-dtu; do
echo "base  $c $(python tools/neus_operating_point.py $c 100 2>/dev/null | tail -1 | cut -c1-70)"
echo "after $c $(NSR_NEUS_BIN_AFTER_ENCODE=1 python tools/neus_operating_point.py $c 100 2>/dev/null | tail -1 | cut -c1-70)"
done; done
