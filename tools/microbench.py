"""Isolated kernel timings (HIP events, median of 20) at the bench's steady-state sizes.  Not part of the test suite."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "instant-nsr-pl_amd"))
import torch
import nsr_hip
from nsr_hip import ops, lib, ptr, stream_ptr, check

def bench(fn, iters=10, warm=3, rep=20):
    """median over `iters` of (time of `rep` back-to-back launches / rep): the queue stays full, so short kernels are
    not masked by the ~20 us host cost of a Python launch"""
    for _ in range(warm): fn()
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize()
        big = torch.empty(1 << 26, device="cuda").zero_()  # ~1 ms of GPU work so the host runs ahead
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(rep): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / rep)
    ts.sort()
    return ts[len(ts) // 2]

which = sys.argv[1] if len(sys.argv) > 1 else "all"
S, M = 88000, 300000
gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
table = (torch.randn(gd.n_entries * 2, device="cuda") * 0.1).half()
# ray-coherent samples: 8192 rays x ~11 / ~37 consecutive samples
def coherent(n, per_ray):
    r = n // per_ray
    o = torch.rand(r, 1, 3, device="cuda") * 0.6 + 0.2
    d = torch.nn.functional.normalize(torch.randn(r, 1, 3, device="cuda"), dim=-1)
    t = torch.arange(per_ray, device="cuda").view(1, -1, 1) * (0.005 / 3)
    return (o + d * t).clamp(0, 1).reshape(-1, 3).contiguous()
xS, xM = coherent(S, 11), coherent(M, 37)
if which in ("all", "grid"):
    for name, x in (("S", xS), ("M", xM)):
        y = torch.empty(x.shape[0], 32, dtype=torch.float16, device="cuda")
        print(f"grid_forward[{name}={x.shape[0]}] coherent: {bench(lambda: ops.hashgrid_forward(x, table, gd, out=y)):.1f} us")
        xr = torch.rand_like(x)
        print(f"grid_forward[{name}] uniform-random: {bench(lambda: ops.hashgrid_forward(xr, table, gd, out=y)):.1f} us")
    dy = torch.randn(16, xS.shape[0], 2, device="cuda")
    g = torch.empty(gd.n_entries * 2, device="cuda")
    print(f"grid_backward owner level-major [S]: {bench(lambda: ops.hashgrid_backward_params(xS, dy, g, gd, accumulate=False, level_major=True)):.1f} us")
    dyr = torch.randn(xS.shape[0], 32, device="cuda")
    print(f"grid_backward atomic [S]: {bench(lambda: ops.hashgrid_backward_params(xS, dyr, g, gd, method='atomic')):.1f} us")
if which in ("all", "mlp"):
    for nh, nout, act in ((1, 16, "none"), (2, 3, "sigmoid")):
        md = nsr_hip.make_mlp_desc(32, nout, nh, act)
        npar = 64 * 32 + (nh - 1) * 4096 + 1024
        w = (torch.randn(npar, device="cuda") * 0.1).half()
        x = torch.randn(xS.shape[0], 32, device="cuda").half()
        out, acts = ops.mlp_forward(x, w, md, True)
        print(f"mlp_forward h{nh} [S]: {bench(lambda: ops.mlp_forward(x, w, md, True)):.1f} us")
        dout = torch.randn(xS.shape[0], nout, device="cuda") * 0.01
        gw = torch.zeros(npar, device="cuda")
        n = xS.shape[0]
        dx = torch.empty(n * 32, device="cuda")
        part = torch.empty(int(lib.nsr_mlp_backward_workspace_floats(ctypes.byref(md), n)), device="cuda")
        def run(gwp, dxp):
            check(lib.nsr_mlp_backward_ex(ptr(dout), 1, nout, None, ptr(out), ptr(x), 0, 32, 0, ptr(acts), ptr(w), ptr(gwp) if gwp is not None else None,
                                          ptr(dxp) if dxp is not None else None, 32, 0, ptr(part), n, 128.0, ctypes.byref(md), None, stream_ptr()))
        print(f"mlp_backward h{nh} [S] dW+dx: {bench(lambda: run(gw, dx)):.1f} us   dW only: {bench(lambda: run(gw, None)):.1f}   dx only: {bench(lambda: run(None, dx)):.1f}")
if which in ("all", "march"):
    import nsr
    from nsr.scene import SyntheticBlender
    from nsr.fused import gather_train_rays
    data = SyntheticBlender(n_images=20, w=200, h=200, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(0)
    rays, _, _, _ = gather_train_rays(data, 8192, gen)
    ii = torch.stack(torch.meshgrid(*[torch.arange(128, device="cuda")] * 3, indexing="ij"), -1).float()
    c = (ii + 0.5) / 128 * 3 - 1.5
    binary = (c.abs().amax(-1) < 0.9)
    roi = torch.tensor([-1.5] * 3 + [1.5] * 3, device="cuda")
    ro, rd = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
    tmin, tmax = ops.ray_aabb_intersect(ro, rd, roi)
    for meth in ("bricks", "bytes"):
        def f():
            h = ops.ray_march_begin(ro, rd, tmin, tmax, roi, binary, 0, 0.00507421875, 0.0, roi_host=[-1.5] * 3 + [1.5] * 3, method=meth)
            return h
        print(f"march count [{meth}] 8192 rays: {bench(f):.1f} us; samples={int(ops.ray_march_finish(f())[1].shape[0])}")
if which == "gridreal":
    # the table backward on the sample distribution of a trained step (surface-concentrated), cumulative over levels
    import nsr
    from nsr.scene import SyntheticBlender
    from nsr.trainer import Trainer
    from nsr.fused import prepare_train_rays
    torch.manual_seed(42)
    cfg = nsr.configs.get("nerf-blender")
    model = nsr.build(cfg).cuda().train()
    data = SyntheticBlender(n_images=100, w=800, h=800, device="cuda", seed=0)
    tr = Trainer(model, data, cfg, seed=42)
    for _ in range(300): tr.train_step()
    torch.cuda.synchronize()
    gen = torch.Generator(device="cuda").manual_seed(1)
    rays, ro, rd, rgb, fg, bg, t_min, t_max = prepare_train_rays(data, 8192, gen, model)
    with torch.no_grad():
        mp = tr.fused.march_and_prune(ro, rd, keep_rows=True, handle=tr.fused.march_begin(ro, rd, t_min, t_max))
    x = mp["x01"].contiguous()
    n = x.shape[0]
    print("kept samples", n, "marched", mp["M"])
    dy = torch.randn(16, n, 2, device="cuda")
    g = torch.empty(gd.n_entries * 2, device="cuda")
    prev = 0.0
    mcs = [int(v) for v in os.environ["GRIDREAL_MC"].split(",")] if "GRIDREAL_MC" in os.environ else (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 16)
    for mc in mcs:
        t = bench(lambda: ops.hashgrid_backward_params(x, dy, g, gd, mask_count=mc, accumulate=False, level_major=True))
        print(f"levels < {mc:2d}: {t:7.1f} us  (+{t - prev:6.1f})")
        prev = t
    xr = torch.rand_like(x)
    if "GRIDREAL_MC" not in os.environ:
        print(f"uniform-random x, all levels: {bench(lambda: ops.hashgrid_backward_params(xr, dy, g, gd, accumulate=False, level_major=True)):.1f} us")
