"""HIP marcher / slab test / contraction vs the C oracle: BIT-EXACT indices and t values."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(n_rays, seed=0, res=128, radius=1.5):
    g = torch.Generator().manual_seed(seed)
    # cameras on a sphere of radius 4 looking roughly at the origin (blender-like)
    o = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1) * 4.0
    d = torch.nn.functional.normalize(-o + torch.randn(n_rays, 3, generator=g) * 0.6, dim=-1)
    ii = torch.stack(torch.meshgrid(*[torch.arange(res)] * 3, indexing="ij"), -1).float()
    c = (ii + 0.5) / res * 2 * radius - radius
    binary = ((c.norm(dim=-1) < 1.0) & (c[..., 0].abs() > 0.15)) | ((c - 0.9).norm(dim=-1) < 0.35)
    roi = torch.tensor([-radius] * 3 + [radius] * 3)
    return o, d, roi, binary


def test_ray_aabb_bit_exact():
    from oracle import nerfacc_ref
    from nsr_hip import ops
    o, d, roi, _ = _scene(5000)
    o[:100] = torch.rand(100, 3) - 0.5  # origins inside the box: t_min clamps to 0
    d[100:110, 0] = 0.0                 # axis-parallel rays: division by zero -> inf, same on both sides
    a, b = nerfacc_ref.ray_aabb_intersect(o, d, roi)
    ga, gb = ops.ray_aabb_intersect(o.cuda(), d.cuda(), roi.cuda())
    assert torch.equal(ga.cpu(), a) and torch.equal(gb.cpu(), b)
    assert bool((a[:100] == 0).all())


@pytest.mark.parametrize("method", ["bricks1", "bricks2", "bytes"])
@pytest.mark.parametrize("n_rays,stepdiv", [(1, 1024), (257, 1024), (4096, 1024), (1000, 128)])
def test_march_aabb_bit_exact(n_rays, stepdiv, method):
    from oracle import nerfacc_ref as N
    from nsr_hip import ops
    o, d, roi, binary = _scene(n_rays, seed=n_rays)
    step = 1.732 * 2 * 1.5 / stepdiv  # reference models/nerf.py:31
    t_min, t_max = N.ray_aabb_intersect(o, d, roi)
    packed_ref, ri_ref, t0_ref, t1_ref = N.march_rays_packed(o, d, t_min, t_max, roi, binary, N.ContractionType.AABB,
                                                             step, 0.0)
    # bricks1: single pass into per-ray scratch; bricks2: brick grid, two passes; bytes: byte grid, two passes
    packed, ri, t0, t1 = ops.ray_march(o.cuda(), d.cuda(), t_min.cuda(), t_max.cuda(), roi.cuda(), binary.cuda(), 0,
                                       step, 0.0, roi_host=roi.tolist() if method == "bricks1" else None,
                                       method="bytes" if method == "bytes" else "bricks")
    assert torch.equal(packed.cpu(), packed_ref)
    assert torch.equal(ri.cpu(), ri_ref)
    assert torch.equal(t0.cpu(), t0_ref) and torch.equal(t1.cpu(), t1_ref)
    assert ri_ref.numel() > 0 or n_rays == 1


def test_march_unbounded_sphere_cone_bit_exact():
    """the background march of NeRF++ (reference models/neus.py:141-171): contracted 256^3-style grid, cone stepping"""
    from oracle import nerfacc_ref as N
    from nsr_hip import ops
    o, d, roi, _ = _scene(700, seed=9, res=64, radius=1.0)
    g = torch.Generator().manual_seed(1)
    binary = torch.rand(64, 64, 64, generator=g) < 0.3
    cone = 10 ** (math.log10(1e3) / 64) - 1.0
    near = torch.full((700,), 0.1)
    far = torch.full((700,), 1e3)
    packed_ref, ri_ref, t0_ref, t1_ref = N.march_rays_packed(o, d, near, far, roi, binary,
                                                             N.ContractionType.UN_BOUNDED_SPHERE, 0.01, cone)
    packed, ri, t0, t1 = ops.ray_march(o.cuda(), d.cuda(), near.cuda(), far.cuda(), roi.cuda(), binary.cuda(), 2, 0.01,
                                       cone)
    assert torch.equal(packed.cpu(), packed_ref) and torch.equal(ri.cpu(), ri_ref)
    assert torch.equal(t0.cpu(), t0_ref) and torch.equal(t1.cpu(), t1_ref)


def test_brick_cache_follows_in_place_grid_edits():
    from oracle import nerfacc_ref as N
    from nsr_hip import ops
    o, d, roi, binary = _scene(300, seed=2)
    bg = binary.cuda()
    t_min, t_max = N.ray_aabb_intersect(o, d, roi)
    args = (o.cuda(), d.cuda(), t_min.cuda(), t_max.cuda(), roi.cuda())
    _, ri_a, _, _ = ops.ray_march(*args, bg, 0, 0.005, 0.0, roi_host=roi.tolist())
    bg[:, :64] = False  # in-place edit bumps the version: the cached bricks must be rebuilt
    binary[:, :64] = False
    _, ri_b, _, _ = ops.ray_march(*args, bg, 0, 0.005, 0.0, roi_host=roi.tolist())
    _, ri_ref, _, _ = N.march_rays_packed(o, d, t_min, t_max, roi, binary, N.ContractionType.AABB, 0.005, 0.0)
    assert torch.equal(ri_b.cpu(), ri_ref) and ri_a.numel() != ri_b.numel()


def test_march_no_grid_and_empty():
    from oracle import nerfacc_ref as N
    from nsr_hip import ops
    o, d, roi, binary = _scene(64)
    empty = torch.zeros_like(binary)
    t_min, t_max = N.ray_aabb_intersect(o, d, roi)
    packed, ri, t0, t1 = ops.ray_march(o.cuda(), d.cuda(), t_min.cuda(), t_max.cuda(), roi.cuda(), empty.cuda(), 0,
                                       0.005, 0.0)
    assert ri.numel() == 0 and bool((packed[:, 1] == 0).all())


@pytest.mark.parametrize("ctype", [0, 2])
def test_contraction_roundtrip_and_oracle(ctype):
    from oracle import nerfacc_ref as N
    from nsr_hip import ops
    roi = torch.tensor([-1.0, -1.5, -0.5, 1.0, 1.5, 2.0])
    x = torch.randn(4000, 3) * 3
    ref = N.contract(x, roi, N.ContractionType(ctype))
    out = ops.contract(x.cuda(), roi.cuda(), ctype).cpu()
    assert torch.equal(out, ref)
    u = torch.rand(4000, 3)
    if ctype == 2:
        u = u[(u - 0.5).norm(dim=1) < 0.499]
    inv_ref = N.contract_inv(u, roi, N.ContractionType(ctype))
    inv = ops.contract(u.cuda(), roi.cuda(), ctype, inverse=True).cpu()
    assert torch.equal(inv, inv_ref)
    back = ops.contract(inv.cuda(), roi.cuda(), ctype).cpu()
    assert torch.allclose(back, u, atol=2e-5)


def test_grid_query_and_pack_and_compact():
    from oracle import nerfacc_ref as N
    from nsr_hip import ops
    o, d, roi, binary = _scene(16)
    x = torch.randn(5000, 3) * 1.2
    assert torch.equal(ops.grid_query(x.cuda(), roi.cuda(), binary.cuda(), 0).cpu(),
                       N.query_grid(x, roi, binary, N.ContractionType.AABB))
    ri = torch.sort(torch.randint(0, 300, (10000,))).values
    assert torch.equal(ops.pack_info(ri.cuda(), 300).cpu(), N.pack_info(ri, 300))
    mask = torch.rand(10000) < 0.37
    t0, t1 = torch.rand(10000, 1), torch.rand(10000, 1)
    a, b, c = ops.compact_samples(mask.cuda(), ri.cuda(), t0.cuda(), t1.cuda())
    assert torch.equal(a.cpu(), ri[mask]) and torch.equal(b.cpu(), t0[mask]) and torch.equal(c.cpu(), t1[mask])
    a, b, c = ops.compact_samples(torch.zeros(10000, dtype=torch.bool).cuda(), ri.cuda(), t0.cuda(), t1.cuda())
    assert a.numel() == 0


def test_march_scratch_overflow_falls_back_to_two_pass():
    """nerfacc's public ``ray_marching`` accepts UN-normalised directions: with |d| = 0.3 a ray emits up to 3.3x the
    samples the single-pass scratch row was sized for (diag/step + 3 assumes unit directions).  The counts stay exact,
    the write pass must notice and re-march: still bit-exact against the oracle"""
    from oracle import nerfacc_ref as N
    from nsr_hip import ops
    o, d, roi, binary = _scene(600, seed=5)
    binary[:] = True
    d = d * 0.3
    step = 1.732 * 2 * 1.5 / 1024
    t_min, t_max = N.ray_aabb_intersect(o, d, roi)
    packed_ref, ri_ref, t0_ref, t1_ref = N.march_rays_packed(o, d, t_min, t_max, roi, binary, N.ContractionType.AABB,
                                                             step, 0.0)
    assert int(packed_ref[:, 1].max()) > 1027  # really beyond the scratch capacity
    packed, ri, t0, t1 = ops.ray_march(o.cuda(), d.cuda(), t_min.cuda(), t_max.cuda(), roi.cuda(), binary.cuda(), 0,
                                       step, 0.0, roi_host=roi.tolist())
    assert torch.equal(packed.cpu(), packed_ref) and torch.equal(ri.cpu(), ri_ref)
    assert torch.equal(t0.cpu(), t0_ref) and torch.equal(t1.cpu(), t1_ref)
