"""Known-answer and property tests that pin the CPU oracle.  No GPU.

The reference ships no vectors for its third-party arithmetic ("parity unpinned"); these hand-derived values and
size-independent properties are what the oracle -- and through it the HIP path -- is anchored on.
"""
import math

import numpy as np
import pytest
import torch

from conftest import COLOR_MLP, DENSITY_MLP, NERF_GRID, NEUS_GRID
from oracle import glue_ref
from oracle import nerfacc_ref as N
from oracle import tcnn_ref as T


def test_hash_known_answers():
    H = T.coherent_prime_hash
    assert H(1, 0, 0) == 1
    assert H(0, 1, 0) == 2654435761 and H(0, 1, 0) % (1 << 19) == 489905
    assert H(0, 0, 1) == 805459861 and H(0, 0, 1) % (1 << 19) == 153493
    assert H(1, 1, 1) == 2922720805 and H(1, 1, 1) % (1 << 19) == 339493
    assert H(101, 57, 4095) == 3475824743 and H(101, 57, 4095) % (1 << 19) == 319591
    assert H(4095, 4095, 4095) == 739598811
    assert H(123456, 7, 99) % (1 << 19) == 159496


def test_dense_index_known_answers():
    assert T.grid_index(3, 2, 1, 16, 4096) == 291            # 3 + 2*16 + 1*256
    assert T.grid_index(48, 48, 48, 49, 117656) == 117648    # C2 level 3, last corner
    assert T.grid_index(16, 16, 16, 16, 4096) == (16 + 16 * 16 + 16 * 256) % 4096  # x == 1.0 border wraps


def test_level_tables_match_configs():
    d = T.GridDesc.from_config(NERF_GRID)   # configs/nerf-blender.yaml:43-49
    assert d.res == [16, 24, 34, 49, 71, 102, 148, 213, 308, 446, 646, 934, 1352, 1956, 2831, 4096]
    assert d.size[:5] == [4096, 13824, 39304, 117656, 357912] and all(s == 1 << 19 for s in d.size[5:])
    assert d.n_entries == 6299960 and d.n_params == 12599920
    d = T.GridDesc.from_config(NEUS_GRID)   # configs/neus-blender.yaml:47-54
    assert d.size[:4] == [32768, 79512, 175616, 405224] and all(s == 1 << 19 for s in d.size[4:])
    assert d.n_entries == 6984576 and d.n_params == 13969152
    assert d.res[5] == 129 and d.res[10] == 513  # fp32 level geometry (fp64 would give 128 / 512)


def test_param_counts_and_layout():
    ewn = T.NetworkWithInputEncoding(3, 16, NERF_GRID, DENSITY_MLP)
    assert ewn.params.numel() == 3072 + 12599920 == 12602992   # [network | grid]
    assert T.Network(32, 3, COLOR_MLP).params.numel() == 7168  # (32+16)*64 + 64*64, network_utils.py:156
    assert T.Network(35, 13, dict(DENSITY_MLP)).desc.in_pad == 48


def test_trilinear_partition_of_unity_and_continuity():
    d = T.GridDesc.from_config(NERF_GRID)
    x = torch.rand(500, 3)
    ones = torch.ones(d.n_entries, 2)
    y = T.hashgrid_encode(x, ones, d, fp16=False)
    assert torch.allclose(y, torch.ones_like(y), atol=1e-6)
    table = torch.randn(d.n_entries, 2) * 0.1
    eps = 1e-6
    y0, y1 = T.hashgrid_encode(x, table, d, fp16=False), T.hashgrid_encode((x + eps).clamp(0, 1), table, d, fp16=False)
    assert (y0 - y1).abs().max() < 0.05  # continuous across cell boundaries


def test_closed_form_input_gradient_equals_autograd():
    d = T.GridDesc.from_config(NEUS_GRID)
    x = torch.rand(64, 3, dtype=torch.float32, requires_grad=True)
    table = (torch.randn(d.n_entries, 2) * 0.1)
    y = T.hashgrid_encode(x, table, d, fp16=False)
    (g,) = torch.autograd.grad(y[:, 6].sum(), x)  # level 3 (dense), feature 0
    # finite differences inside the cell
    h = 1e-4
    for k in range(3):
        xp = x.detach().clone()
        xp[:, k] += h
        fd = (T.hashgrid_encode(xp, table, d, fp16=False)[:, 6] - y[:, 6].detach()) / h
        ok = (torch.floor((xp * d.scale[3] + 0.5)) == torch.floor(x.detach() * d.scale[3] + 0.5)).all(dim=1)
        assert torch.allclose(fd[ok], g[ok, k], rtol=5e-2, atol=5e-3)


def test_sh4_orthonormal():
    n_t, n_p = 64, 128
    ct, wt = np.polynomial.legendre.leggauss(n_t)
    phi = (np.arange(n_p) + 0.5) * 2 * math.pi / n_p
    st = np.sqrt(1 - ct ** 2)
    dirs = np.stack([np.outer(st, np.cos(phi)), np.outer(st, np.sin(phi)), np.outer(ct, np.ones(n_p))], -1).reshape(-1, 3)
    w = np.outer(wt, np.full(n_p, 2 * math.pi / n_p)).reshape(-1)
    Y = T.sh4_encode(torch.from_numpy((dirs + 1) / 2).double(), fp16=False).numpy()
    gram = (Y * w[:, None]).T @ Y
    assert np.allclose(gram, np.eye(16), atol=1e-10)


def test_mlp_padding_constant_one_and_no_bias():
    desc = T.MLPDesc(19, 3, dict(COLOR_MLP, output_activation="none", n_hidden_layers=1))
    p = torch.zeros(desc.n_params)
    w0, wl = desc.split(p)
    w0[0, 19] = 1.0  # weight on the first PADDED input column (constant 1.0)
    wl[0, 0] = 1.0
    out = T.mlp_forward(torch.zeros(4, 19), p, desc)
    assert torch.allclose(out[:, 0], torch.ones(4)) and bool((out[:, 1:] == 0).all())
    assert bool((T.mlp_forward(torch.zeros(4, 19), torch.zeros(desc.n_params), desc) == 0).all())  # no biases


def test_constants_of_the_configs():
    assert abs(1.732 * 2 * 1.5 / 1024 - 0.00507421875) < 1e-12        # models/nerf.py:31, C2
    assert abs(10 ** (math.log10(1e4) / 1024) - 1 - 0.009035044841) < 1e-9   # models/nerf.py:24
    assert abs(10 ** (math.log10(1e3) / 64) - 1 - 0.113973860) < 1e-8        # models/neus.py:59
    assert abs(math.exp(0.3 * 10) - 20.0855) < 1e-3                           # models/neus.py:28


def test_ray_aabb_known_answers():
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    o = torch.tensor([[0.0, 0, 3], [0, 0, 0.5], [5, 5, 5], [0, 0, -3]])
    d = torch.tensor([[0.0, 0, -1], [0, 0, 1], [1, 0, 0], [0, 0, -1]])
    t_min, t_max = N.ray_aabb_intersect(o, d, aabb)
    assert t_min.tolist()[:2] == [2.0, 0.0] and t_max.tolist()[:2] == [4.0, 0.5]  # inside: t_min clamps to 0
    assert t_min[2] == 1e10 and t_max[2] == 1e10                                   # miss sentinel, neus.py:155-157
    assert t_min[3] == 0.0 and t_max[3] < 0                                        # box behind: no samples later


def test_marcher_properties():
    torch.manual_seed(0)
    res, r = 32, 1.5
    roi = torch.tensor([-r] * 3 + [r] * 3)
    ii = torch.stack(torch.meshgrid(*[torch.arange(res)] * 3, indexing="ij"), -1).float()
    c = (ii + 0.5) / res * 2 * r - r
    binary = c.norm(dim=-1) < 0.9
    o = torch.nn.functional.normalize(torch.randn(200, 3), dim=-1) * 4
    d = torch.nn.functional.normalize(-o + torch.randn(200, 3) * 0.4, dim=-1)
    step = 0.01
    t_min, t_max = N.ray_aabb_intersect(o, d, roi)
    packed, ri, t0, t1 = N.march_rays_packed(o, d, t_min, t_max, roi, binary, N.ContractionType.AABB, step, 0.0)
    assert ri.numel() > 1000
    assert bool((ri[1:] >= ri[:-1]).all())                                  # ray-major
    same = ri[1:] == ri[:-1]
    assert bool((t0[1:, 0][same] >= t1[:-1, 0][same] - 1e-6).all())        # ascending, non-overlapping
    assert torch.allclose(t1 - t0, torch.full_like(t0, step), atol=1e-6)   # constant step (cone_angle 0)
    mid = o[ri] + d[ri] * (t0 + t1) / 2
    assert bool(N.query_grid(mid, roi, binary, N.ContractionType.AABB).all())  # every sample sits in an occupied cell
    assert torch.equal(packed, N.pack_info(ri, 200))
    # brute force: same samples as stepping through EVERY lattice cell without the DDA skip (up to boundary steps)
    full = torch.ones_like(binary)
    _, ri_f, t0_f, t1_f = N.march_rays_packed(o, d, t_min, t_max, roi, full, N.ContractionType.AABB, step, 0.0)
    mid_f = o[ri_f] + d[ri_f] * (t0_f + t1_f) / 2
    n_brute = int(N.query_grid(mid_f, roi, binary, N.ContractionType.AABB).sum())
    assert abs(n_brute - ri.numel()) <= 3 * 200


def test_contraction_inverse_roundtrip():
    roi = torch.tensor([-1.0, -2, -0.5, 1, 2, 1.5])
    for ct in (N.ContractionType.AABB, N.ContractionType.UN_BOUNDED_SPHERE):
        x = torch.randn(1000, 3) * 5
        u = N.contract(x, roi, ct)
        assert bool(((u >= 0) & (u <= 1)).all()) or ct == N.ContractionType.AABB
        back = N.contract_inv(u, roi, ct)
        keep = (x.abs() < 50).all(dim=1)
        assert torch.allclose(back[keep], x[keep], rtol=2e-3, atol=2e-3)
    # geometry.py:17-29 and nerfacc's contraction agree (the reference relies on it)
    x = torch.randn(500, 3) * 4
    a = glue_ref.contract_to_unisphere(x, 1.5, N.ContractionType.UN_BOUNDED_SPHERE)
    b = N.contract(x, torch.tensor([-1.5] * 3 + [1.5] * 3), N.ContractionType.UN_BOUNDED_SPHERE)
    assert torch.allclose(a, b, atol=1e-6)


def test_compositing_properties():
    g = torch.Generator().manual_seed(0)
    cnt = torch.randint(0, 200, (50,), generator=g)
    ri = torch.repeat_interleave(torch.arange(50), cnt)
    n = ri.numel()
    t0 = torch.rand(n, 1, generator=g)
    t1 = t0 + 0.01
    sig = torch.rand(n, 1, generator=g) * 50
    w = N.render_weight_from_density(t0, t1, sig, ray_indices=ri, n_rays=50)
    assert bool((w >= 0).all())
    opac = N.accumulate_along_rays(w, ri, None, 50)
    T_end = torch.exp(-N.accumulate_along_rays(sig * (t1 - t0), ri, None, 50))
    assert torch.allclose(opac + T_end, torch.ones_like(opac) * (cnt > 0)[:, None] + T_end * (cnt == 0)[:, None], atol=1e-5)
    a = 1 - torch.exp(-sig * (t1 - t0))
    assert torch.allclose(N.render_weight_from_alpha(a, ray_indices=ri, n_rays=50), w, atol=1e-6)
    # sequential fp32 C scan (nerfacc's naive path) agrees with the fp64 vectorised one
    import ctypes
    T_c = torch.empty(n)
    N._lib.nsro_transmittance_from_sigma(ctypes.c_int64(n), N._p(ri, N._i64), N._p(t0.view(-1).contiguous(), N._f),
                                         N._p(t1.view(-1).contiguous(), N._f), N._p(sig.view(-1).contiguous(), N._f),
                                         N._p(T_c, N._f))
    T_v = N.render_transmittance_from_density(t0, t1, sig, ray_indices=ri).view(-1)
    assert torch.allclose(T_c, T_v, rtol=1e-4, atol=1e-7)
    # gradients against torch autograd on a per-ray python loop
    sig2 = sig.clone().requires_grad_(True)
    N.render_weight_from_density(t0, t1, sig2, ray_indices=ri, n_rays=50).sum().backward()
    sig3 = sig.clone().requires_grad_(True)
    tot = 0
    for r in range(50):
        m = ri == r
        sd = (sig3[m] * (t1[m] - t0[m])).view(-1)
        T = torch.exp(-(torch.cumsum(sd, 0) - sd))
        tot = tot + (T * (1 - torch.exp(-sd))).sum()
    tot.backward()
    assert torch.allclose(sig2.grad, sig3.grad, rtol=1e-4, atol=1e-6)


def test_occupancy_grid_semantics():
    roi = torch.tensor([-1.5] * 3 + [1.5] * 3)
    g = N.OccupancyGrid(roi, 16)
    assert set(g.state_dict().keys()) == {"_roi_aabb", "_binary", "resolution", "occs"}
    assert not bool(g.binary.any()) and g.num_cells == 4096
    g.train()
    fn = lambda x: (x.norm(dim=-1, keepdim=True) < 1.0).float() * 0.5  # noqa: E731
    g.every_n_step(step=0, occ_eval_fn=fn)       # step 0 % 16 == 0 -> full sweep during warm-up
    frac = g.binary.float().mean()
    assert 0.1 < frac < 0.5                       # sphere of radius 1 in a cube of side 3: 15.5 %
    before = g.occs.clone()
    g.every_n_step(step=5, occ_eval_fn=fn)        # not a multiple of 16: no update
    assert torch.equal(before, g.occs)
    g.every_n_step(step=16, occ_eval_fn=lambda x: torch.zeros(x.shape[0], 1))
    assert torch.allclose(g.occs, before * 0.95)  # EMA decay, max(occ*0.95, new)
    g.eval()
    with pytest.raises(RuntimeError):
        g.every_n_step(step=32, occ_eval_fn=fn)


def test_distortion_loss_prefix_form_equals_definition():
    """oracle/distloss_ref.py: the O(n) prefix-sum form and its hand-written gradient == the O(n^2) definition + autograd"""
    import torch
    from oracle import distloss_ref
    g = torch.Generator().manual_seed(0)
    counts = [0, 5, 1, 0, 17, 64, 3]
    ray_id = torch.cat([torch.full((c,), r, dtype=torch.int64) for r, c in enumerate(counts)])
    n = ray_id.numel()
    w = (torch.rand(n, generator=g) * 0.3).double().requires_grad_(True)
    m = torch.cat([torch.sort(torch.rand(c, generator=g) * 4 + 0.5).values for c in counts if c]).double()
    interval = (torch.rand(n, generator=g) * 0.05 + 0.01).double()
    ref = distloss_ref.distortion_definition(w, m, interval, ray_id)
    gref, = torch.autograd.grad(ref, w)
    loss, grad = distloss_ref.flatten_eff_distloss(w.detach(), m, interval, ray_id)
    assert abs(float(loss) - float(ref)) < 1e-12 * max(1.0, abs(float(ref)))
    assert torch.allclose(grad, gref, rtol=1e-10, atol=1e-13)
    # one ray, two unit-weight samples a distance d apart, zero-length intervals: L = 2 * d
    one = distloss_ref.distortion_definition(torch.ones(2).double(), torch.tensor([1.0, 3.5]).double(),
                                             torch.zeros(2).double(), torch.zeros(2, dtype=torch.int64))
    assert abs(float(one) - 5.0) < 1e-12
