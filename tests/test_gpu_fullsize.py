"""Size-independent properties of the HIP path at BASELINE.json's full sizes (C2: L16 T2^19 F2, 8192 rays, 2^18 samples)
-- where the CPU oracle no longer finishes in seconds: conservation, reproducibility, sortedness, bounds, scaling laws."""
import pytest
import torch

pytestmark = pytest.mark.gpu

C2_GRID = (16, 2, 19, 16, 1.447269237440378)


def test_table_backward_conserves_and_reproduces_at_full_size():
    """binned fixed-point table backward at 2^18 samples: per level, the gradient entries sum to the sum of dy (trilinear
    weights are a partition of unity); two runs are BIT-identical on the hashed levels (integer accumulation; the small
    dense levels sum per-chunk slabs in fp32, whose chunking follows the arrival order of the items); the global-atomic
    kernel agrees"""
    import nsr_hip
    from nsr_hip import ops
    hd = nsr_hip.make_grid_desc(*C2_GRID)
    n = 1 << 18
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(n, 3, device="cuda", generator=g)
    dy = torch.randn(16, n, 2, device="cuda", generator=g) * 1e-3          # level-major [L][n][F]
    grad = torch.empty(hd.n_entries * 2, device="cuda")
    ops.hashgrid_backward_params(x, dy, grad, hd, accumulate=False, level_major=True)
    again = torch.full_like(grad, float("nan"))
    ops.hashgrid_backward_params(x, dy, again, hd, accumulate=False, level_major=True)
    offs, sizes, res = list(hd.offset)[:16], list(hd.size)[:16], list(hd.resolution)[:16]
    for lvl in range(16):
        a, b = grad[offs[lvl] * 2:(offs[lvl] + sizes[lvl]) * 2], again[offs[lvl] * 2:(offs[lvl] + sizes[lvl]) * 2]
        if res[lvl] ** 3 > sizes[lvl]:
            assert torch.equal(a, b), lvl                                    # hashed level: bit-reproducible
        else:
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), lvl  # fp32 sums over differently chunked slabs
    for lvl in range(16):
        sl = grad[offs[lvl] * 2:(offs[lvl] + sizes[lvl]) * 2].view(-1, 2).double().sum(0)
        want = dy[lvl].double().sum(0)
        assert torch.allclose(sl, want, rtol=1e-4, atol=1e-6), (lvl, sl, want)
    atomic = torch.zeros_like(grad)
    ops.hashgrid_backward_params(x, dy.permute(1, 0, 2).reshape(n, 32).contiguous(), atomic, hd, method="atomic")
    assert float((grad - atomic).norm() / atomic.norm()) < 1e-5


def test_marcher_output_is_sorted_and_inside_the_box_at_8192_rays():
    from nsr_hip import ops
    g = torch.Generator().manual_seed(0)
    n_rays, radius, res = 8192, 1.5, 128
    o = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1) * 4.0
    d = torch.nn.functional.normalize(-o + torch.randn(n_rays, 3, generator=g) * 0.6, dim=-1)
    ii = torch.stack(torch.meshgrid(*[torch.arange(res)] * 3, indexing="ij"), -1).float()
    c = (ii + 0.5) / res * 2 * radius - radius
    binary = ((c.norm(dim=-1) < 1.0) & (c[..., 0].abs() > 0.15)).cuda()
    roi = torch.tensor([-radius] * 3 + [radius] * 3).cuda()
    o, d = o.cuda(), d.cuda()
    tmin, tmax = ops.ray_aabb_intersect(o, d, roi)
    step = 1.732 * 2 * radius / 1024
    outs = [ops.ray_march(o, d, tmin, tmax, roi, binary, 0, step, 0.0, roi_host=[-radius] * 3 + [radius] * 3, method=m)
            for m in ("bricks", "bytes", "bricks")]
    packed, ri, t0, t1 = outs[0]
    for other in outs[1:]:                                                   # both grid formats, and idempotence
        assert all(torch.equal(a, b) for a, b in zip(outs[0], other))
    n = ri.numel()
    assert n > 100_000 and int(packed[:, 1].sum()) == n
    assert bool((ri[1:] >= ri[:-1]).all())                                   # ray-major
    same = ri[1:] == ri[:-1]
    assert bool((t0.view(-1)[1:][same] >= t1.view(-1)[:-1][same]).all())     # disjoint, increasing intervals along a ray
    assert bool((t1 > t0).all()) and float((t1 - t0).max()) <= step * 1.001
    mid = o[ri] + d[ri] * ((t0 + t1) / 2)
    assert float(mid.abs().max()) <= radius * (1 + 1e-5)                     # every sample inside the AABB ...
    cell = ((mid + radius) / (2 * radius) * res).long().clamp(0, res - 1)
    assert bool(binary[cell[:, 0], cell[:, 1], cell[:, 2]].all())            # ... and inside an occupied cell
    starts = torch.cumsum(packed[:, 1], 0) - packed[:, 1]
    assert torch.equal(packed[:, 0].long(), starts.long())


def test_compositing_bounds_at_full_size():
    """2^18 samples on 8192 rays: transmittance non-increasing in [0,1], weights >= 0, per-ray weight sums == opacity <= 1"""
    import nerfacc
    g = torch.Generator(device="cuda").manual_seed(1)
    n_rays = 8192
    counts = torch.randint(0, 65, (n_rays,), device="cuda", generator=g)
    ri = torch.repeat_interleave(torch.arange(n_rays, device="cuda"), counts)
    n = ri.numel()
    t0 = torch.rand(n, 1, device="cuda", generator=g)
    t1 = t0 + 0.005
    sigma = torch.rand(n, 1, device="cuda", generator=g) * 200
    w = nerfacc.render_weight_from_density(t0, t1, sigma, ray_indices=ri, n_rays=n_rays)
    trans = nerfacc.render_transmittance_from_density(t0, t1, sigma, ray_indices=ri, n_rays=n_rays)
    opacity = nerfacc.accumulate_along_rays(w, ri, values=None, n_rays=n_rays)
    assert float(w.min()) >= 0 and float(trans.max()) <= 1 and float(trans.min()) >= 0
    same = ri[1:] == ri[:-1]
    assert bool((trans.view(-1)[1:][same] <= trans.view(-1)[:-1][same] + 1e-7).all())
    assert float(opacity.max()) <= 1 + 1e-5
    assert torch.allclose(opacity.view(-1), torch.zeros(n_rays, device="cuda").index_add_(0, ri, w.view(-1)), atol=1e-5)
    assert bool((opacity.view(-1)[counts == 0] == 0).all())


def test_distortion_loss_scaling_laws_at_full_size():
    """L >= 0, L(c w) = c^2 L(w), invariance under a shift of the midpoints"""
    from torch_efficient_distloss import flatten_eff_distloss
    g = torch.Generator(device="cuda").manual_seed(2)
    counts = torch.randint(1, 65, (8192,), device="cuda", generator=g)
    ri = torch.repeat_interleave(torch.arange(8192, device="cuda"), counts)
    n = ri.numel()
    w = torch.rand(n, device="cuda", generator=g) * 0.05
    key = ri.double() * 10 + torch.rand(n, device="cuda", generator=g).double()
    m = (torch.sort(key).values - ri.double() * 10).float() * 4 + 0.5       # sorted within every ray
    dt = torch.full((n,), 0.005, device="cuda")
    base = float(flatten_eff_distloss(w, m, dt, ri))
    assert base > 0
    assert abs(float(flatten_eff_distloss(3 * w, m, dt, ri)) - 9 * base) < 1e-4 * 9 * base
    assert abs(float(flatten_eff_distloss(w, m + 2.5, dt, ri)) - base) < 1e-4 * base
