"""Host-side profile (cProfile) of bench.py's modular_path: the reference's own model statements on the drop-in packages.
    python tools/modular_host_profile.py [steps]"""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda")
print(bench.modular_path(dev, warmup=60, steps=steps))
pr = cProfile.Profile()
pr.enable()
res = bench.modular_path(dev, warmup=60, steps=steps)
pr.disable()
print(res)
for key in ("tottime", "cumulative"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(45)
    print(buf.getvalue())
