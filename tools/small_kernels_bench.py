"""Isolated timings (HIP events, median of 50) of the step's small per-ray / per-sample kernels at the LATE regime's shape: 8,192 ray
slots, ~1/3 of them with samples, `kept` samples in all -- wave-per-ray vs sample-partitioned compositing, scan +
copy vs the copy with the packing folded in, two data-gradient launches vs the pair kernel.   python tools/small_kernels_bench.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
import torch
import nsr_hip
from nsr_hip import check, lib, ops, ptr, stream_ptr
from kernel_microbench import median_us

res = {}
s = stream_ptr()
for kept in (45000, 100000):
    n_rays = 8192
    g = torch.Generator().manual_seed(kept)
    active = torch.rand(n_rays, generator=g) < 0.34
    counts = torch.where(active, torch.poisson(torch.full((n_rays,), kept / (0.34 * n_rays)), generator=g), torch.zeros(n_rays)).long()
    starts = torch.cumsum(counts, 0) - counts
    n = int(counts.sum())
    packed = torch.stack([starts, counts], 1).int().cuda()
    out1 = (torch.randn(n, 16, generator=g) * 2 - 1).half().cuda()
    out2 = torch.rand(n, 16, generator=g).half().cuda()
    t0 = torch.rand(n, generator=g).cuda(); t1 = t0 + 0.01
    bg = torch.tensor([1.0, 0.5, 0.25]).cuda(); gt = torch.rand(n_rays, 3, generator=g).cuda()
    w, tr = torch.zeros(n).cuda(), torch.zeros(n).cuda()
    rgb, op, dp = torch.zeros(n_rays, 3).cuda(), torch.zeros(n_rays).cuda(), torch.zeros(n_rays).cuda()
    acc = torch.zeros(2).cuda(); d_rgb, d_logit = torch.zeros(n, 3).cuda(), torch.zeros(n).cuda()
    part = torch.zeros(int(lib.nsr_composite_l1_partials_floats(n_rays))).cuda()
    r = {"kept": n}
    r["composite_forward_wave_us"] = median_us(lambda: check(lib.nsr_composite_forward_smooth_l1(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(bg), ptr(w), ptr(tr), ptr(rgb), ptr(op), ptr(dp), ptr(gt), ptr(part), n_rays, s), "f"))
    r["composite_backward_wave_us"] = median_us(lambda: check(lib.nsr_composite_backward_smooth_l1_partials(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(bg), ptr(w), ptr(tr), ptr(rgb), ptr(op), ptr(gt), ptr(part), ptr(acc), 1.0, ptr(d_rgb), ptr(d_logit), n_rays, s), "b"))
    ri = torch.repeat_interleave(torch.arange(n_rays), counts).cuda()
    r["composite_forward_samples_us"] = median_us(lambda: check(lib.nsr_composite_forward_samples(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(ri), ptr(bg), ptr(w), ptr(tr), ptr(rgb), ptr(op), ptr(dp), ptr(gt), ptr(part), n_rays, n, None, s), "f"))
    r["composite_backward_samples_us"] = median_us(lambda: check(lib.nsr_composite_backward_samples(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(ri), ptr(bg), ptr(w), ptr(tr), None, None, None, None, ptr(rgb), ptr(op), ptr(gt), ptr(part), ptr(acc), 1.0, ptr(d_rgb), ptr(d_logit), n_rays, n, None, s), "b"))
    if os.environ.get("ONLY_COMPOSITE"):
        res[str(kept)] = r
        continue
    # the two networks' data gradients
    dc, dd = nsr_hip.make_mlp_desc(32, 3, 2, "sigmoid"), nsr_hip.make_mlp_desc(32, 16, 1, "none")
    wc = (torch.randn(7168, generator=g) * 0.2).half().cuda(); wd = (torch.randn(3072, generator=g) * 0.2).half().cuda()
    enc = torch.randn(16, n, 2, generator=g).half().cuda()
    o1 = torch.empty(n, 16).half().cuda(); a1 = torch.empty(1, n, 64).half().cuda()
    check(lib.nsr_mlp_forward_ex(ptr(enc), 0, 32, 2, ptr(wd), ptr(o1), ptr(a1), n, ctypes.byref(dd), None, s), "fwd")
    tex = torch.cat([o1, torch.rand(n, 16, generator=g).half().cuda()], 1).contiguous()
    o2, a2 = ops.mlp_forward(tex, wc, dc, save_acts=True)
    dr = (torch.randn(n, 3, generator=g) * 1e-3).cuda(); dl = (torch.randn(n, generator=g) * 1e-3).cuda()
    ws = lambda d: torch.zeros(int(lib.nsr_mlp_backward_workspace_floats(ctypes.byref(d), n)), device="cuda")
    pc, pd = ws(dc), ws(dd); gc, gdd = torch.zeros(7168).cuda(), torch.zeros(3072).cuda()
    dtex = torch.zeros(n, 32).cuda(); denc = torch.zeros(16, n, 2).cuda()
    def two():
        check(lib.nsr_mlp_backward_phases(ptr(dr), 1, 3, None, ptr(o2), ptr(tex), 0, 32, 0, ptr(a2), ptr(wc), ptr(gc), ptr(dtex), 32, 0, ptr(pc), n, 65536.0, ctypes.byref(dc), None, s, 1), "c")
        check(lib.nsr_mlp_backward_phases(ptr(dtex), 1, 32, ptr(dl), ptr(o1), ptr(enc), 0, 32, 2, ptr(a1), ptr(wd), ptr(gdd), ptr(denc), 32, 2, ptr(pd), n, 65536.0, ctypes.byref(dd), None, s, 1), "d")
    r["dgrad_two_launches_us"] = median_us(two)
    r["dgrad_pair_us"] = median_us(lambda: check(lib.nsr_mlp_dgrad_pair(ptr(dr), ptr(dl), ptr(o2), ptr(a2), ptr(wc), ptr(pc), ptr(a1), ptr(wd), ptr(pd), ptr(denc), n, 65536.0, ctypes.byref(dc), ctypes.byref(dd), None, s), "p"))
    def wg():
        check(lib.nsr_mlp_backward_phases(ptr(dr), 1, 3, None, ptr(o2), ptr(tex), 0, 32, 0, ptr(a2), ptr(wc), ptr(gc), None, 32, 0, ptr(pc), n, 65536.0, ctypes.byref(dc), None, s, 2), "c")
        check(lib.nsr_mlp_backward_phases(ptr(denc), 1, 32, ptr(dl), ptr(o1), ptr(enc), 0, 32, 2, ptr(a1), ptr(wd), ptr(gdd), None, 32, 2, ptr(pd), n, 65536.0, ctypes.byref(dd), None, s, 2), "d")
    for cap in (512, 128, 64):
        lib.nsr_mlp_wgrad_max_blocks(cap)
        r[f"wgrad_both_networks_cap{cap}_us"] = median_us(wg)
    lib.nsr_mlp_wgrad_max_blocks(128)
    r["mlp_forward_color_us"] = median_us(lambda: ops.mlp_forward(tex, wc, dc, save_acts=True))
    res[f"kept_{kept}"] = r
print(json.dumps(res))
