"""HIP hash grid vs the oracle (oracle/tcnn_ref.py), through the C ABI.  Tolerances per SURVEY.md A.8."""
import pytest
import torch

from conftest import NERF_GRID, NEUS_GRID

pytestmark = pytest.mark.gpu


def _setup(cfg, n, seed=0, table_scale=0.1, F=None):
    from oracle import tcnn_ref
    import nsr_hip
    cfg = dict(cfg)
    if F is not None:
        cfg["n_features_per_level"] = F
    od = tcnn_ref.GridDesc.from_config(cfg)
    hd = nsr_hip.make_grid_desc(cfg["n_levels"], cfg["n_features_per_level"], cfg["log2_hashmap_size"],
                                cfg["base_resolution"], cfg["per_level_scale"])
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 3, generator=g)
    x[0] = 0.0
    x[1] = 1.0  # the x == 1 border (corner coordinate reaches res)
    table = (torch.randn(od.n_params, generator=g) * table_scale).half().float()
    return od, hd, x, table


@pytest.mark.parametrize("cfg", [NERF_GRID, NEUS_GRID], ids=["nerf16", "neus32"])
def test_desc_matches_oracle(cfg):
    od, hd, _, _ = _setup(cfg, 4)
    assert hd.n_entries == od.n_entries
    for l in range(od.L):
        assert hd.scale[l] == od.scale[l] and hd.resolution[l] == od.res[l]
        assert hd.size[l] == od.size[l] and hd.offset[l] == od.offset[l]


@pytest.mark.parametrize("cfg,F", [(NERF_GRID, 2), (NEUS_GRID, 2), (NERF_GRID, 1), (NERF_GRID, 4), (NERF_GRID, 8)],
                         ids=["nerf16", "neus32", "F1", "F4", "F8"])
def test_forward_parity(cfg, F):
    from oracle import tcnn_ref
    from nsr_hip import ops
    od, hd, x, table = _setup(cfg, 4099, F=F)
    ref = tcnn_ref.hashgrid_encode(x, table.view(-1, od.F), od)
    y = ops.hashgrid_forward(x.cuda(), table.half().cuda(), hd).float().cpu()
    # both round an fp32 blend of the same fp16 table values to fp16: at most 1 fp16 ulp apart
    err = (y - ref).abs()
    tol = ref.abs() * 2 ** -10 + 1e-6
    assert bool((err <= tol).all()), f"max err {err.max()} at {err.argmax()}"
    assert float((y != ref).float().mean()) < 0.01  # and almost always bit-identical


def test_forward_mask_count_zeroes_levels():
    from oracle import tcnn_ref
    from nsr_hip import ops
    od, hd, x, table = _setup(NEUS_GRID, 1000)
    ref = tcnn_ref.hashgrid_encode(x, table.view(-1, od.F), od)
    y = ops.hashgrid_forward(x.cuda(), table.half().cuda(), hd, mask_count=5).float().cpu()
    assert bool((y[:, 10:] == 0).all())
    assert torch.allclose(y[:, :10], ref[:, :10], rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("method", ["owner", "atomic"])
@pytest.mark.parametrize("dy_f32", [False, True])
def test_backward_params_parity(dy_f32, method):
    from oracle import tcnn_ref
    from nsr_hip import ops
    od, hd, x, table = _setup(NERF_GRID, 3001)
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(x.shape[0], 32, generator=g)
    dy[7] = 0.0  # a sample with zero upstream gradient
    if not dy_f32:
        dy = dy.half().float()
    t = table.clone().requires_grad_(True)
    tcnn_ref.hashgrid_encode(x, t.view(-1, od.F), od, fp16=False).backward(dy)
    grad = torch.full((od.n_params,), 7.0, device="cuda")  # garbage: accumulate=False must overwrite everything
    dyc = (dy if dy_f32 else dy.half()).cuda()
    ops.hashgrid_backward_params(x.cuda(), dyc, grad, hd, accumulate=False, method=method)
    g1 = grad.clone()
    ops.hashgrid_backward_params(x.cuda(), dyc, grad, hd, accumulate=True, method=method)
    assert torch.allclose(grad, 2 * g1, rtol=1e-5, atol=1e-7)
    g1 = g1.cpu()
    rel = (g1 - t.grad).norm() / t.grad.norm()
    assert rel < 1e-5, rel  # fp32 accumulation everywhere: only the summation order differs
    assert bool(((g1 != 0) == (t.grad != 0)).all())


def test_backward_params_owner_level_major_masked_and_features():
    from nsr_hip import ops
    for F in (1, 2, 4, 8):
        od, hd, x, table = _setup(NEUS_GRID, 1000, F=F)
        C = 16 * F
        dy = torch.randn(1000, C).cuda()
        xc = x.cuda()
        a = torch.zeros(od.n_params, device="cuda")
        b = torch.empty(od.n_params, device="cuda")
        ops.hashgrid_backward_params(xc, dy, a, hd, mask_count=9, method="atomic")
        dy_lm = dy.view(1000, 16, F).permute(1, 0, 2).contiguous()
        ops.hashgrid_backward_params(xc, dy_lm, b, hd, mask_count=9, accumulate=False, level_major=True)
        assert (a - b).norm() / a.norm() < 1e-5
        assert bool((b[hd.offset[9] * F:] == 0).all())  # masked levels are written as zeros


def test_backward_input_and_double_backward():
    from oracle import tcnn_ref
    from nsr_hip import ops
    od, hd, x, table = _setup(NEUS_GRID, 1531, table_scale=0.05)
    g = torch.Generator().manual_seed(7)
    dy = torch.randn(x.shape[0], 32, generator=g)
    gin = torch.randn(x.shape[0], 3, generator=g)
    xo = x.clone().requires_grad_(True)
    to = table.clone().requires_grad_(True)
    dyo = dy.clone().requires_grad_(True)
    y = tcnn_ref.hashgrid_encode(xo, to.view(-1, od.F), od, fp16=False)
    (dx_ref,) = torch.autograd.grad(y, xo, dyo, create_graph=True)
    d_dy_ref, dt_ref, dx2_ref = torch.autograd.grad(dx_ref, [dyo, to, xo], gin)
    xc, tc, dyc = x.cuda(), table.half().cuda(), dy.cuda()
    dx = ops.hashgrid_backward_input(xc, tc, dyc, hd).cpu()
    assert (dx - dx_ref.detach()).norm() / dx_ref.norm() < 1e-5
    gt = torch.zeros(od.n_params, device="cuda")
    d_dy, dx2 = ops.hashgrid_backward_backward_input(xc, tc, dyc, gin.cuda(), hd, grad_table=gt)
    assert (d_dy.cpu() - d_dy_ref).norm() / d_dy_ref.norm() < 1e-5
    assert (gt.cpu() - dt_ref).norm() / dt_ref.norm() < 1e-5
    assert (dx2.cpu() - dx2_ref).norm() / dx2_ref.norm() < 1e-4


def test_full_size_properties():
    """BASELINE config C2 at one training step's size: partition of unity + linearity (size-independent)."""
    from nsr_hip import ops
    import nsr_hip
    hd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    n = 1 << 18
    x = torch.rand(n, 3, device="cuda")
    ones = torch.ones(hd.n_entries * 2, dtype=torch.float16, device="cuda")
    y = ops.hashgrid_forward(x, ones, hd).float()
    assert float((y - 1).abs().max()) < 2e-3  # trilinear weights sum to one on every level
    a = (torch.randn(hd.n_entries * 2, device="cuda") * 0.1).half()
    ya, y2a = ops.hashgrid_forward(x, a, hd).float(), ops.hashgrid_forward(x, (a.float() * 2).half(), hd).float()
    assert torch.allclose(y2a, 2 * ya, rtol=2e-3, atol=1e-4)


@pytest.mark.parametrize("mask_count", [16, 7])
def test_forward_taps_equals_seven_plain_encodes(mask_count):
    """nsr_hashgrid_forward_taps (one position load, corners shared between a sample and its six finite-difference taps,
    models/geometry.py:181-197) == nsr_hashgrid_forward on the 7 n points, bit for bit; eps = one cell of the finest
    active level (models/geometry.py:231-233), points near the box faces are clamped like the reference clamps them"""
    import ctypes
    import tinycudann as tcnn
    from conftest import NEUS_GRID
    from nsr_hip import check, lib, ops, ptr, stream_ptr
    enc = tcnn.Encoding(3, NEUS_GRID).cuda()
    with torch.no_grad():
        enc.params.normal_(0, 0.1)
    n, radius = 20000, 1.0
    eps = 2 * radius / (32 * 1.3195079107728942 ** (mask_count - 1))
    g = torch.Generator().manual_seed(mask_count)
    o = (torch.rand(n, 3, generator=g) * 2 - 1) * radius
    o[:200] = o[:200].sign() * radius * (1 - 1e-4 * torch.rand(200, 3, generator=g))  # inside eps of the faces
    rays_o, rays_d = o.cuda(), torch.zeros(n, 3, device="cuda")
    ri = torch.arange(n, device="cuda")
    t0 = torch.zeros(n, device="cuda")
    x7 = torch.empty(7 * n, 3, device="cuda")
    check(lib.nsr_neus_points(ptr(rays_o), ptr(rays_d), ptr(ri), ptr(t0), ptr(t0), radius, eps, 1, ptr(x7), None, n, None,
                              stream_ptr()), "nsr_neus_points")
    taps = x7.view(7, n, 3)
    want_pts = ((o[None] + torch.tensor([[0, 0, 0], [eps, 0, 0], [-eps, 0, 0], [0, eps, 0], [0, -eps, 0], [0, 0, eps],
                                         [0, 0, -eps]])[:, None, :]).clamp(-radius, radius) + radius) / (2 * radius)
    assert torch.allclose(taps.cpu(), want_pts, atol=1e-6)
    table = enc.table_half(enc.params)
    want = ops.hashgrid_forward(x7, table, enc.grid_desc, mask_count)
    got = torch.empty_like(want)
    check(lib.nsr_hashgrid_forward_taps(ptr(x7), ptr(table), ptr(got), n, 32, 0, mask_count, ctypes.byref(enc.grid_desc),
                                        None, stream_ptr()), "nsr_hashgrid_forward_taps")
    assert torch.equal(got, want)
    lm = torch.empty(16, 7 * n, 2, dtype=torch.float16, device="cuda")  # level-major variant: same numbers
    check(lib.nsr_hashgrid_forward_taps(ptr(x7), ptr(table), ptr(lm), n, 0, 1, mask_count, ctypes.byref(enc.grid_desc),
                                        None, stream_ptr()), "nsr_hashgrid_forward_taps")
    assert torch.equal(lm.permute(1, 0, 2).reshape(7 * n, 32), want)
    rows = (7 * n + 15) // 16 * 16  # tile-major [rows / 16][L][16][F] (what the fused NeuS steps use): same numbers
    tm = torch.zeros(rows // 16, 16, 16, 2, dtype=torch.float16, device="cuda")
    check(lib.nsr_hashgrid_forward_taps(ptr(x7), ptr(table), ptr(tm), n, 0, 2, mask_count, ctypes.byref(enc.grid_desc),
                                        None, stream_ptr()), "nsr_hashgrid_forward_taps")
    assert torch.equal(tm.permute(0, 2, 1, 3).reshape(rows, 32)[:7 * n], want)
    assert float(got[:, 2 * mask_count:].abs().max() if mask_count < 16 else 0.0) == 0.0


def test_owner_backward_with_second_order_equals_two_passes():
    """first-order (level-major dy) + second-order (directional-derivative weights) table gradient in ONE binning pass
    == the two separate owner passes added up (hashed levels: bit-identical integer accumulation is not expected because
    the two terms are summed before the fixed-point conversion; compare to 1e-6 of the norm)"""
    import ctypes
    import tinycudann as tcnn
    from conftest import NEUS_GRID
    from nsr_hip import check, lib, ptr, stream_ptr
    enc = tcnn.Encoding(3, NEUS_GRID).cuda()
    desc = enc.grid_desc
    n = 30011
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(n, 3, device="cuda", generator=g)
    dy_first = torch.randn(16, n, 2, device="cuda", generator=g)           # level-major
    dy_second = torch.randn(n, 36, device="cuda", generator=g)             # row-major, columns 3..34 used
    gx = torch.randn(n, 3, device="cuda", generator=g) * 1e-3
    nws = int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(desc), n))
    ws = torch.empty(nws, device="cuda")
    a = torch.empty(desc.n_entries * 2, device="cuda")
    s = stream_ptr()
    check(lib.nsr_hashgrid_backward_params_owner(ptr(x), ptr(dy_first), 2, 0, ptr(a), ptr(ws), n, 16, 1.0, 0,
                                                 ctypes.byref(desc), None, s), "first")
    off = ctypes.c_void_p(dy_second.data_ptr() + 12)
    table = enc.table_half(enc.params)
    check(lib.nsr_hashgrid_backward_backward_input_ws(ptr(x), ptr(table), off, 1, 36, ptr(gx), None, 0, ptr(a), None, ptr(ws),
                                                      n, 16, ctypes.byref(desc), s), "second")
    b = torch.empty_like(a)
    check(lib.nsr_hashgrid_backward_params_owner_with_second_order(ptr(x), ptr(dy_first), off, 36, ptr(gx), ptr(b), ptr(ws), n,
                                                                   16, 0, 0, ctypes.byref(desc), s), "merged")
    assert float((a - b).norm() / a.norm()) < 1e-6
    # ... and with the binning done ahead of time (what the fused step queues on a helper stream)
    c = torch.empty_like(a)
    # (binning and accumulation pick the slice configuration from the point count AND the kind of pass: with the threshold
    # between n and 2 n a plain launch takes the small configuration, a second-order one the large -- the binning entry for
    # second-order passes has to follow)
    for thr in (0xffffffff, int(1.5 * n), 0):
        old = lib.nsr_hashgrid_owner_large_from(thr)
        try:
            c = torch.empty_like(a)
            check(lib.nsr_hashgrid_backward_params_owner_bin_second_order(ptr(x), ptr(ws), n, 16, ctypes.byref(desc), None, s),
                  "bin")
            check(lib.nsr_hashgrid_backward_params_owner_with_second_order(ptr(x), ptr(dy_first), off, 36, ptr(gx), ptr(c),
                                                                           ptr(ws), n, 16, 0, 1, ctypes.byref(desc), s),
                  "merged, pre-binned")
        finally:
            lib.nsr_hashgrid_owner_large_from(old)
        assert float((c - b).norm() / b.norm()) < 1e-6, thr
    assert float(a.abs().max()) > 0
    # the second-order dy handed over level-major (dy_stride == 0): what nsr_hashgrid_jac_apply_ex leaves behind
    dy_lm = dy_second[:, 3:35].reshape(n, 16, 2).permute(1, 0, 2).contiguous()
    e = torch.empty_like(a)
    check(lib.nsr_hashgrid_backward_params_owner_with_second_order(ptr(x), ptr(dy_first), ptr(dy_lm), 0, ptr(gx), ptr(e), ptr(ws),
                                                                   n, 16, 0, 0, ctypes.byref(desc), s), "merged, level-major dy")
    assert float((e - b).norm() / b.norm()) < 1e-6  # (the small dense levels sum their slabs in fp32, order not fixed)


@pytest.mark.parametrize("mask_count", [16, 9])
def test_cached_jacobian_reproduces_input_gradient_and_its_double_backward(mask_count):
    """nsr_hashgrid_forward_jac + nsr_hashgrid_jac_apply == nsr_hashgrid_backward_input (J^T dy) and the d_dy output of
    nsr_hashgrid_backward_backward_input (J g), which re-gather the table; encodings unchanged (rel 1e-5: fp32 both ways)"""
    import ctypes
    import tinycudann as tcnn
    from conftest import NEUS_GRID
    from nsr_hip import check, lib, ops, ptr, stream_ptr
    enc = tcnn.Encoding(3, NEUS_GRID).cuda()
    with torch.no_grad():
        enc.params.normal_(0, 0.05)
    desc, n = enc.grid_desc, 12345
    g_ = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(n, 3, device="cuda", generator=g_)
    dy = torch.randn(n, 36, device="cuda", generator=g_)
    gx = torch.randn(n, 3, device="cuda", generator=g_)
    table = enc.table_half(enc.params)
    s = stream_ptr()
    y = torch.empty(16, n, 2, dtype=torch.float16, device="cuda")
    jac = torch.empty(16 * n * 6, device="cuda")
    check(lib.nsr_hashgrid_forward_jac(ptr(x), ptr(table), ptr(y), n, 0, 1, mask_count, ctypes.byref(desc), ptr(jac), None, s),
          "fwd_jac")
    want_y = ops.hashgrid_forward(x, table, desc, mask_count)
    assert torch.equal(y.permute(1, 0, 2).reshape(n, 32), want_y)
    off = ctypes.c_void_p(dy.data_ptr() + 12)
    dx = torch.empty(n, 3, device="cuda")
    d_dy = torch.zeros(n, 36, device="cuda")
    check(lib.nsr_hashgrid_jac_apply(ptr(jac), n, ctypes.byref(desc), off, 36, ptr(dx), ptr(gx),
                                     ctypes.c_void_p(d_dy.data_ptr() + 12), 36, None, s), "jac_apply")
    lm_copy = torch.empty(16, n, 2, device="cuda")  # ..._ex: the same products + a level-major copy of dy
    dx2 = torch.empty_like(dx)
    check(lib.nsr_hashgrid_jac_apply_ex(ptr(jac), n, ctypes.byref(desc), off, 36, ptr(dx2), None, None, 0, ptr(lm_copy), None, s),
          "jac_apply_ex")
    assert torch.equal(dx2, dx)
    assert torch.equal(lm_copy.permute(1, 0, 2).reshape(n, 32), dy[:, 3:35])
    want_dx = torch.empty(n, 3, device="cuda")
    check(lib.nsr_hashgrid_backward_input(ptr(x), ptr(table), off, 1, 36, ptr(want_dx), n, mask_count, ctypes.byref(desc), s),
          "bwd_input")
    want_ddy = torch.zeros(n, 32, device="cuda")
    check(lib.nsr_hashgrid_backward_backward_input(ptr(x), ptr(table), off, 1, 36, ptr(gx), ptr(want_ddy), 32, None, None, n,
                                                   mask_count, ctypes.byref(desc), s), "bwd_bwd")
    assert float((dx - want_dx).norm() / want_dx.norm()) < 1e-5
    assert float((d_dy[:, 3:35] - want_ddy).norm() / want_ddy.norm()) < 1e-5
    assert float(d_dy[:, 3 + 2 * mask_count:35].abs().max() if mask_count < 16 else 0.0) == 0.0


def test_owner_backward_with_fused_adamw_matches_gradient_plus_optimizer():
    """nsr_hashgrid_backward_params_owner_accumulate_adam == owner_accumulate -> nsr_adamw_step (device schedule) on the
    table: parameters, moments and fp16 image bit-identical, over three steps (pow() start + running beta products)"""
    import ctypes
    import nsr_hip
    from nsr_hip import NsrTableAdam, check, lib, ops, ptr, stream_ptr
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    n, n_tab = 30000, gd.n_entries * 2
    g = torch.Generator(device="cuda").manual_seed(3)
    ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(gd), n)), device="cuda")

    def fresh():
        gg = torch.Generator(device="cuda").manual_seed(5)
        p = torch.randn(n_tab, device="cuda", generator=gg) * 0.1
        return dict(p=p, m=torch.zeros_like(p), v=torch.zeros_like(p), h=torch.empty(n_tab, dtype=torch.float16, device="cuda"),
                    step=torch.zeros(1, dtype=torch.int32, device="cuda"), hyper=torch.zeros(12, device="cuda"))

    a, b = fresh(), fresh()
    grad = torch.empty(n_tab, device="cuda")
    ms = (2, 0x7fffffff, 0x7fffffff)
    for it in range(3):
        x = torch.rand(n, 3, device="cuda", generator=g)
        dy = torch.randn(16, n, 2, device="cuda", generator=g) * 1e-3
        check(lib.nsr_hashgrid_backward_params_owner_bin(ptr(x), ptr(ws), n, 16, ctypes.byref(gd), None, stream_ptr()), "bin")
        # (a) gradient, then the optimizer kernel
        check(lib.nsr_hashgrid_backward_params_owner_accumulate(ptr(x), ptr(dy), 2, 0, ptr(grad), ptr(ws), n, 16, 1.0, 0,
                                                                ctypes.byref(gd), None, stream_ptr()), "accumulate")
        ops.adam_tick(a["step"], a["hyper"], 0.01, 0.9, 0.99, 0.33, ms)
        ops.adamw_step(a["p"], grad, a["m"], a["v"], a["h"], 0.01, 0.9, 0.99, 1e-15, 0.01, it + 1, zero_grad=False,
                       hyper=a["hyper"])
        # (b) fused; the schedule is advanced afterwards (here by the tick kernel, in the trainer by the MLP update)
        d = NsrTableAdam()
        d.params, d.exp_avg, d.exp_avg_sq, d.shadow = b["p"].data_ptr(), b["m"].data_ptr(), b["v"].data_ptr(), b["h"].data_ptr()
        d.step, d.hyper = b["step"].data_ptr(), b["hyper"].data_ptr()
        d.base_lr, d.beta1, d.beta2, d.gamma = 0.01, 0.9, 0.99, 0.33
        d.milestone0, d.milestone1, d.milestone2 = ms
        d.eps, d.weight_decay = 1e-15, 0.01
        check(lib.nsr_hashgrid_backward_params_owner_accumulate_adam(ptr(x), ptr(dy), 2, 0, ptr(ws), n, 16, 1.0,
                                                                     ctypes.byref(gd), None, ctypes.byref(d), stream_ptr()),
              "accumulate_adam")
        ops.adam_tick(b["step"], b["hyper"], 0.01, 0.9, 0.99, 0.33, ms)
        off = [int(o) * 2 for o in gd.offset[:17]]
        for k in ("p", "m", "v", "h"):
            if not torch.equal(a[k], b[k]):
                bad = [(lvl, int((a[k][off[lvl]:off[lvl + 1]] != b[k][off[lvl]:off[lvl + 1]]).sum())) for lvl in range(16)]
                raise AssertionError((it, k, [t for t in bad if t[1]]))


@pytest.mark.parametrize("mask_count,eps_level", [(16, 15), (9, 8), (4, 3), (16, 12)])
def test_owner_backward_stencil_mode_matches_plain_backward_over_all_taps(mask_count, eps_level):
    """nsr_hashgrid_backward_params_owner_{bin,accumulate}_taps (in-cell taps folded into their sample's items) == the plain
    owner backward over all 7 N points; eps = one cell of the finest active level (models/geometry.py:224-236), taps clamped
    to the box like k_neus_points does, some samples ON the box faces.  (16, 12): eps is 1.3 - 2.3 cells on levels 13 - 15 --
    taps that land beyond the face neighbour"""
    import ctypes
    import nsr_hip
    from nsr_hip import check, lib, ptr, stream_ptr
    gd = nsr_hip.make_grid_desc(16, 2, 19, 32, 1.3195079107728942)  # neuralangelo's grid
    g = torch.Generator(device="cuda").manual_seed(7)
    n_c = 20011
    x = torch.rand(n_c, 3, device="cuda", generator=g)
    x[:500] = torch.round(x[:500])          # corners / faces of the box: clamped taps
    x[500:900, 0] = 1.0
    res = 32 * 1.3195079107728942 ** eps_level
    eps = 1.0 / res                          # unit-cube eps = 2 r / res / (2 r)
    x7 = x.repeat(7, 1).view(7, n_c, 3).clone()
    for t in range(6):
        x7[t + 1, :, t // 2] += eps if t % 2 == 0 else -eps
    x7 = x7.clamp_(0.0, 1.0).view(-1, 3).contiguous()
    dy = torch.randn(16, 7 * n_c, 2, device="cuda", generator=g)
    dy[:, n_c:] *= 0.3
    n = 7 * n_c
    ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(gd), n)), device="cuda")
    tws = torch.empty(int(lib.nsr_hashgrid_backward_params_taps_workspace_floats(ctypes.byref(gd), n_c)), device="cuda")
    want = torch.empty(gd.n_entries * 2, device="cuda")
    got = torch.full_like(want, 7.0)
    check(lib.nsr_hashgrid_backward_params_owner(ptr(x7), ptr(dy), 2, 0, ptr(want), ptr(ws), n, mask_count, 1.0, 0,
                                                 ctypes.byref(gd), None, stream_ptr()), "plain")
    check(lib.nsr_hashgrid_backward_params_owner_bin_taps(ptr(x7), ptr(ws), ptr(tws), n_c, mask_count, ctypes.byref(gd),
                                                          stream_ptr()), "bin_taps")
    check(lib.nsr_hashgrid_backward_params_owner_accumulate_taps(ptr(x7), ptr(dy), ptr(got), ptr(ws), ptr(tws), n_c,
                                                                 mask_count, 0, ctypes.byref(gd), stream_ptr()), "acc_taps")
    off = [int(o) * 2 for o in gd.offset[:17]]
    for lvl in range(16):
        a, b = got[off[lvl]:off[lvl + 1]], want[off[lvl]:off[lvl + 1]]
        if lvl >= mask_count:
            assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0
            continue
        rel = float((a - b).norm() / b.norm())
        assert rel < 2e-5, (lvl, rel, float((a - b).abs().max()))
    # the 7-tap ENCODE writes the same crossing masks (it holds both cells anyway): bit for bit what the binning pass computes,
    # and binning + accumulation from them give the same gradient bit for bit
    table = (torch.randn(gd.n_entries * 2, device="cuda", generator=g) * 0.1).half()
    enc = torch.empty(7 * n_c, 32, dtype=torch.float16, device="cuda")
    tws2 = torch.zeros_like(tws)
    check(lib.nsr_hashgrid_forward_taps_masks(ptr(x7), ptr(table), ptr(enc), n_c, 32, 0, mask_count, ctypes.byref(gd),
                                              ptr(tws2), stream_ptr()), "fwd masks")
    nb = mask_count * n_c
    m1 = tws.view(torch.uint8)[:nb].clone()
    m2 = tws2.view(torch.uint8)[:nb]
    assert torch.equal(m1, m2), int((m1 != m2).sum())
    assert int(m2.count_nonzero()) > 0
    got2 = torch.full_like(want, 7.0)
    check(lib.nsr_hashgrid_backward_params_owner_bin_taps_masked(ptr(x7), ptr(ws), ptr(tws2), n_c, mask_count,
                                                                 ctypes.byref(gd), stream_ptr()), "bin_taps_masked")
    check(lib.nsr_hashgrid_backward_params_owner_accumulate_taps(ptr(x7), ptr(dy), ptr(got2), ptr(ws), ptr(tws2), n_c,
                                                                 mask_count, 0, ctypes.byref(gd), stream_ptr()), "acc_taps")
    hashed = off[11]  # (levels cut into chunk slabs / row-merged dense levels add in fp32: reproducible to rounding only)
    assert torch.equal(got2[hashed:], got[hashed:])
    assert float((got2 - got).abs().max()) <= 1e-6 * float(got.abs().max())


@pytest.mark.parametrize("groups", [[(0, 16)], [(11, 16), (0, 11)], [(13, 16), (5, 13), (2, 5), (0, 2)]],
                         ids=["one", "two", "four"])
def test_owner_backward_level_ranges_and_bf16_transport(groups):
    """nsr_hashgrid_backward_params_owner_accumulate_range: the table gradient launched as runs of levels (finest first, what
    the ray-sharded step does so that the exchange of the finest levels overlaps the rest of the backward) equals the one
    launch bit for bit as fp32, and as bf16 it is the round-to-nearest-even image of that gradient -- incl. NaN flushing of a
    slice that saw a non-finite dy"""
    import ctypes
    import nsr_hip
    from nsr_hip import check, lib, ptr, stream_ptr
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    n, n_tab = 40000, gd.n_entries * 2
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.rand(n, 3, device="cuda", generator=g)
    dy = torch.randn(16, n, 2, device="cuda", generator=g) * 1e-3
    ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(gd), n)), device="cuda")
    D = ctypes.byref(gd)
    check(lib.nsr_hashgrid_backward_params_owner_bin(ptr(x), ptr(ws), n, 16, D, None, stream_ptr()), "bin")
    want = torch.empty(n_tab, device="cuda")
    check(lib.nsr_hashgrid_backward_params_owner_accumulate(ptr(x), ptr(dy), 2, 0, ptr(want), ptr(ws), n, 16, 1.0, 0, D, None,
                                                            stream_ptr()), "accumulate")
    got = torch.full((n_tab,), float("nan"), device="cuda")
    got16 = torch.full((n_tab + 64,), float("nan"), dtype=torch.bfloat16, device="cuda")
    for lo, hi in groups:
        check(lib.nsr_hashgrid_backward_params_owner_accumulate_range(ptr(x), ptr(dy), ptr(got), None, ptr(ws), n, 16, 1.0, lo,
                                                                      hi, D, None, stream_ptr()), "range fp32")
        check(lib.nsr_hashgrid_backward_params_owner_accumulate_range(ptr(x), ptr(dy), None, ptr(got16), ptr(ws), n, 16, 1.0,
                                                                      lo, hi, D, None, stream_ptr()), "range bf16")
    assert torch.equal(got, want)
    assert torch.equal(got16[:n_tab], want.to(torch.bfloat16))          # torch rounds to nearest even too
    assert bool(torch.isnan(got16[n_tab:].float()).all())                # the padding of the exchange is not touched
    assert float(want.abs().sum()) > 0
    # exactly one of the two outputs
    rc = lib.nsr_hashgrid_backward_params_owner_accumulate_range(ptr(x), ptr(dy), ptr(got), ptr(got16), ptr(ws), n, 16, 1.0, 0, 16,
                                                                 D, None, stream_ptr())
    assert rc != 0
    # a non-finite dy flushes the slices it reaches as NaN in the transport format as well (GradScaler's found_inf)
    dy2 = dy.clone()
    dy2[15, 7, 0] = float("inf")
    check(lib.nsr_hashgrid_backward_params_owner_accumulate_range(ptr(x), ptr(dy2), None, ptr(got16), ptr(ws), n, 16, 1.0, 11, 16,
                                                                  D, None, stream_ptr()), "range bf16 inf")
    off = [int(o) * 2 for o in gd.offset[:17]]
    assert bool(torch.isnan(got16[off[15]:off[16]].float()).any()) and not bool(torch.isnan(got16[off[11]:off[15]].float()).any())


def test_owner_backward_small_and_large_slice_configurations_agree():
    """the two compiled configurations of the owner-computes backward (csrc/hashgrid_owner.inc: 2^11-entry slices x 256 threads
    and 2^13 x 1024, picked per launch by the point count) give the same gradient: bit-identical on the hashed levels (integer
    accumulation per entry), fp32-rounding-close on the dense levels that are summed from chunk slabs -- plain, fused-AdamW,
    second-order and stencil modes"""
    import ctypes
    import nsr_hip
    from nsr_hip import check, lib, ptr, stream_ptr
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    n, n_tab = 50000, gd.n_entries * 2
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.rand(n, 3, device="cuda", generator=g)
    dy = torch.randn(16, n, 2, device="cuda", generator=g) * 1e-3
    dy_rm = torch.randn(n, 32, device="cuda", generator=g) * 1e-3
    gdir = torch.randn(n, 3, device="cuda", generator=g)
    ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(gd), n)), device="cuda")
    D = ctypes.byref(gd)
    off = [int(o) * 2 for o in gd.offset[:17]]
    res = {}
    old = lib.nsr_hashgrid_owner_large_from(0)
    try:
        for name, thr in (("large", 0), ("small", 0xffffffff)):
            lib.nsr_hashgrid_owner_large_from(thr)
            a = torch.empty(n_tab, device="cuda")
            check(lib.nsr_hashgrid_backward_params_owner(ptr(x), ptr(dy), 2, 0, ptr(a), ptr(ws), n, 16, 1.0, 0, D, None,
                                                         stream_ptr()), "owner")
            b = torch.zeros(n_tab, device="cuda")
            check(lib.nsr_hashgrid_backward_params_owner_with_second_order(ptr(x), ptr(dy), ptr(dy_rm), 32, ptr(gdir), ptr(b),
                                                                           ptr(ws), n, 16, 0, 0, D, stream_ptr()), "second")
            res[name] = (a, b)
    finally:
        lib.nsr_hashgrid_owner_large_from(old)
    for k in range(2):
        s_, l_ = res["small"][k], res["large"][k]
        assert float(l_.abs().sum()) > 0
        for lvl in range(16):
            sl = slice(off[lvl], off[lvl + 1])
            dense = (int(gd.resolution[lvl]) ** 3) <= int(gd.size[lvl])
            if dense:
                assert float((s_[sl] - l_[sl]).norm() / l_[sl].norm()) < 1e-5, (k, lvl)
            else:
                assert torch.equal(s_[sl], l_[sl]), (k, lvl)


@pytest.mark.parametrize("F,n", [(2, 4099), (2, 16), (4, 1000), (1, 333), (8, 50)])
def test_forward_tile_major_layout_holds_the_same_values(F, n):
    """y_level_major = 2: [ceil(n/16)][L][16][F] -- plain encode (both forward variants) and the Jacobian-caching one"""
    import ctypes
    import nsr_hip
    from nsr_hip import check, lib, ops, ptr, stream_ptr
    L = 32 // F if F > 1 else 16
    gd = nsr_hip.make_grid_desc(L, F, 15, 16, 1.5)
    g = torch.Generator().manual_seed(F)
    table = (torch.rand(gd.n_entries * F, generator=g) - 0.5).half().cuda()
    x = torch.rand(n, 3, generator=g).cuda()
    want = ops.hashgrid_forward(x, table, gd, L - 1)
    rows = (n + 15) // 16 * 16
    for variant in ((0, 1), (0, 2)):
        lib.nsr_hashgrid_forward_variant(*variant)
        tm = torch.zeros(rows // 16, L, 16, F, dtype=torch.float16, device="cuda")
        check(lib.nsr_hashgrid_forward_ex(ptr(x), ptr(table), ptr(tm), n, 0, 2, L - 1, ctypes.byref(gd), None, stream_ptr()),
              "nsr_hashgrid_forward_ex")
        assert torch.equal(tm.permute(0, 2, 1, 3).reshape(rows, L * F)[:n], want), variant
    lib.nsr_hashgrid_forward_variant(0, 2)
    tm = torch.zeros(rows // 16, L, 16, F, dtype=torch.float16, device="cuda")
    jac = torch.empty(L, n, F, 3, device="cuda")
    check(lib.nsr_hashgrid_forward_jac(ptr(x), ptr(table), ptr(tm), n, 0, 2, L - 1, ctypes.byref(gd), ptr(jac), None,
                                       stream_ptr()), "nsr_hashgrid_forward_jac")
    assert torch.equal(tm.permute(0, 2, 1, 3).reshape(rows, L * F)[:n], want)


def test_owner_backward_placements_agree_and_claim_cursors_reset():
    """every placement of the work units (nsr_hashgrid_owner_tune key 0: dealt / listed / striped / claimed at run time) gives
    the same table gradient in both configurations -- the same bits on the hashed levels, fp32 rounding on the chunked dense
    ones.  The claimed placement takes its units with atomic cursors that the launch itself clears: 300 launches (more than
    the 256 cursor slots, so slots are reused) on two streams at once must all give that gradient, and it must equal the
    oracle's (oracle/hashgrid_ref.py)"""
    import ctypes
    import nsr_hip
    from nsr_hip import check, lib, ptr
    from oracle import tcnn_ref
    gd = nsr_hip.make_grid_desc(NERF_GRID["n_levels"], NERF_GRID["n_features_per_level"], NERF_GRID["log2_hashmap_size"],
                                NERF_GRID["base_resolution"], NERF_GRID["per_level_scale"])
    n, n_tab = 30000, gd.n_entries * 2
    g = torch.Generator(device="cuda").manual_seed(33)
    x = torch.rand(n, 3, device="cuda", generator=g)
    x[: n // 2] = x[: n // 2] * 0.05 + 0.4  # (half of the samples in a few cells: uneven units)
    dy = torch.randn(16, n, 2, device="cuda", generator=g) * 1e-3
    D = ctypes.byref(gd)
    off = [int(o) * 2 for o in gd.offset[:17]]
    n_ws = int(lib.nsr_hashgrid_backward_params_workspace_floats(D, n))

    def run(stream, ws):
        a = torch.empty(n_tab, device="cuda")
        with torch.cuda.stream(stream):
            check(lib.nsr_hashgrid_backward_params_owner(ptr(x), ptr(dy), 2, 0, ptr(a), ptr(ws), n, 16, 1.0, 0, D, None,
                                                         stream.cuda_stream), "owner")
        return a

    def same(a, b, what):
        for lvl in range(16):
            sl = slice(off[lvl], off[lvl + 1])
            if (int(gd.resolution[lvl]) ** 3) <= int(gd.size[lvl]):
                assert float((a[sl] - b[sl]).norm() / b[sl].norm()) < 1e-5, (what, lvl)
            else:
                assert torch.equal(a[sl], b[sl]), (what, lvl)

    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    ws0, ws1 = torch.empty(n_ws, device="cuda"), torch.empty(n_ws, device="cuda")
    torch.cuda.synchronize()
    old_thr = lib.nsr_hashgrid_owner_large_from(0)
    old_pl = lib.nsr_hashgrid_owner_tune(0, 2.0)
    try:
        ref = None
        for thr in (0, 0xffffffff):
            lib.nsr_hashgrid_owner_large_from(thr)
            for pl in (2, 0, 1, 3, 4):
                lib.nsr_hashgrid_owner_tune(0, float(pl))
                a = run(s0, ws0)
                torch.cuda.synchronize()
                if ref is None:
                    ref = a
                    assert float(ref.abs().sum()) > 0
                same(a, ref, (thr, pl))
        lib.nsr_hashgrid_owner_tune(0, 3.0)
        # runs of levels (what the ray-sharded step launches, finest first) under the claimed placement, both configurations:
        # a launch's pair lists hold the levels of its range only
        from nsr_hip import stream_ptr
        for thr in (0, 0xffffffff):
            lib.nsr_hashgrid_owner_large_from(thr)
            check(lib.nsr_hashgrid_backward_params_owner_bin(ptr(x), ptr(ws0), n, 16, D, None, stream_ptr()), "bin")
            part = torch.full((n_tab,), float("nan"), device="cuda")
            for lo, hi in ((11, 16), (5, 11), (0, 5)):
                check(lib.nsr_hashgrid_backward_params_owner_accumulate_range(ptr(x), ptr(dy), ptr(part), None, ptr(ws0), n, 16, 1.0,
                                                                              lo, hi, D, None, stream_ptr()), "range")
            torch.cuda.synchronize()
            same(part, ref, ("ranges", thr))
        for thr, rounds in ((0, 100), (0xffffffff, 50)):
            lib.nsr_hashgrid_owner_large_from(thr)
            outs = []
            for _ in range(rounds):
                outs.append(run(s0, ws0))
                outs.append(run(s1, ws1))
                if len(outs) > 8:
                    same(outs.pop(0), ref, "repeat")
            torch.cuda.synchronize()
            for o in outs:
                same(o, ref, "repeat")
    finally:
        lib.nsr_hashgrid_owner_tune(0, old_pl)
        lib.nsr_hashgrid_owner_large_from(old_thr)
    # the oracle's gradient of the same (x, dy): autograd through oracle/tcnn_ref.py's encode
    od = tcnn_ref.GridDesc.from_config(NERF_GRID)
    t = torch.zeros(od.n_params, requires_grad=True)
    tcnn_ref.hashgrid_encode(x.cpu(), t.view(-1, od.F), od, fp16=False).backward(dy.permute(1, 0, 2).reshape(n, 32).cpu())
    rel = (ref.cpu() - t.grad).norm() / t.grad.norm()
    assert rel < 1e-5, rel
