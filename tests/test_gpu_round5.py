"""Round-5 kernels of the NeRF step against the kernels they replace (which are pinned to the oracle elsewhere):
flat segmented compositing vs one wave per ray, the packing folded into the kept-row copy vs scan + copy, both networks'
data gradients in one kernel vs two launches, the dense levels of the table backward through ray-run merged fixed-point
atomics vs the owner workgroups, and the whole step with every form switched off / on (reference models/nerf.py:95-109,
systems/nerf.py:97, models/network_utils.py:181,209)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _packed(n_rays, max_count, seed, long_every=0):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, max_count, (n_rays,), generator=g)
    counts[::7] = 0  # rays without samples
    if long_every:
        counts[3::long_every] = torch.randint(65, 400, counts[3::long_every].shape, generator=g)  # rays that span 64-sample chunks
    starts = torch.cumsum(counts, 0) - counts
    return torch.stack([starts, counts], 1).int().cuda(), int(counts.sum()), g


@pytest.mark.parametrize("n_rays,long_every", [(8192, 0), (1147, 5), (3, 0), (9, 2), (64, 1)])
@pytest.mark.parametrize("mode", ["folded", "acc", "upstream"])
def test_flat_compositing_matches_wave_per_ray(n_rays, long_every, mode):
    from nsr_hip import check, lib, ptr, stream_ptr
    packed, n, g = _packed(n_rays, 40, 100 + n_rays, long_every)
    m = max(n, 1)
    out1 = (torch.randn(m, 16, generator=g) * 2 - 1).half().cuda()
    out2 = torch.rand(m, 16, generator=g).half().cuda()
    t0 = torch.rand(m, generator=g).cuda()
    t1 = t0 + 0.01
    bg = torch.tensor([1.0, 0.5, 0.25]).cuda()
    gt = torch.rand(n_rays, 3, generator=g).cuda()
    up = dict(c=(torch.randn(n_rays, 3, generator=g) * 0.1).cuda(), o=(torch.randn(n_rays, generator=g) * 0.1).cuda(),
              d=(torch.randn(n_rays, generator=g) * 0.1).cuda(), w=(torch.randn(m, generator=g) * 0.1).cuda())
    s = stream_ptr()

    def buffers():
        return dict(w=torch.zeros(m).cuda(), tr=torch.zeros(m).cuda(), rgb=torch.full((n_rays, 3), -7.0).cuda(),
                    op=torch.full((n_rays,), -7.0).cuda(), dp=torch.full((n_rays,), -7.0).cuda(),
                    acc=torch.full((2,), -1.0).cuda(), d_rgb=torch.zeros(m, 3).cuda(), d_logit=torch.zeros(m).cuda())

    a, b = buffers(), buffers()
    part_a = torch.full((int(lib.nsr_composite_l1_partials_floats(n_rays)),), float("nan")).cuda()
    part_b = part_a.clone()
    # wave per ray
    if mode == "folded":
        check(lib.nsr_composite_forward_smooth_l1(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(bg),
                                                  ptr(a["w"]), ptr(a["tr"]), ptr(a["rgb"]), ptr(a["op"]), ptr(a["dp"]), ptr(gt),
                                                  ptr(part_a), n_rays, s), "fwd")
        check(lib.nsr_composite_backward_smooth_l1_partials(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed),
                                                            ptr(bg), ptr(a["w"]), ptr(a["tr"]), ptr(a["rgb"]), ptr(a["op"]),
                                                            ptr(gt), ptr(part_a), ptr(a["acc"]), 2.0, ptr(a["d_rgb"]),
                                                            ptr(a["d_logit"]), n_rays, s), "bwd")
    else:
        check(lib.nsr_composite_forward(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(bg), ptr(a["w"]),
                                        ptr(a["tr"]), ptr(a["rgb"]), ptr(a["op"]), ptr(a["dp"]), n_rays, s), "fwd")
        if mode == "acc":
            check(lib.nsr_smooth_l1_valid_set(ptr(a["rgb"]), ptr(a["op"]), ptr(gt), ptr(a["acc"]), n_rays, s), "l1")
            check(lib.nsr_composite_backward_smooth_l1(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed),
                                                       ptr(bg), ptr(a["w"]), ptr(a["tr"]), ptr(a["rgb"]), ptr(a["op"]), ptr(gt),
                                                       ptr(a["acc"]), 2.0, ptr(a["d_rgb"]), ptr(a["d_logit"]), n_rays, s), "bwd")
        else:
            check(lib.nsr_composite_backward_ex(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(bg),
                                                ptr(a["w"]), ptr(a["tr"]), ptr(up["c"]), ptr(up["o"]), ptr(up["d"]), ptr(up["w"]),
                                                ptr(a["d_rgb"]), ptr(a["d_logit"]), n_rays, s), "bwd")
    # flat
    check(lib.nsr_composite_forward_flat(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(bg), ptr(b["w"]),
                                         ptr(b["tr"]), ptr(b["rgb"]), ptr(b["op"]), ptr(b["dp"]),
                                         ptr(gt) if mode == "folded" else None, ptr(part_b) if mode == "folded" else None,
                                         n_rays, s), "fwd flat")
    if mode == "folded":
        check(lib.nsr_composite_backward_flat(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(bg),
                                              ptr(b["w"]), ptr(b["tr"]), None, None, None, None, ptr(b["rgb"]), ptr(b["op"]),
                                              ptr(gt), ptr(part_b), ptr(b["acc"]), 2.0, ptr(b["d_rgb"]), ptr(b["d_logit"]),
                                              n_rays, s), "bwd flat")
    elif mode == "acc":
        check(lib.nsr_smooth_l1_valid_set(ptr(b["rgb"]), ptr(b["op"]), ptr(gt), ptr(b["acc"]), n_rays, s), "l1")
        check(lib.nsr_composite_backward_flat(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(bg),
                                              ptr(b["w"]), ptr(b["tr"]), None, None, None, None, ptr(b["rgb"]), ptr(b["op"]),
                                              ptr(gt), None, ptr(b["acc"]), 2.0, ptr(b["d_rgb"]), ptr(b["d_logit"]), n_rays, s),
              "bwd flat")
    else:
        check(lib.nsr_composite_backward_flat(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(bg),
                                              ptr(b["w"]), ptr(b["tr"]), ptr(up["c"]), ptr(up["o"]), ptr(up["d"]), ptr(up["w"]),
                                              None, None, None, None, None, 1.0, ptr(b["d_rgb"]), ptr(b["d_logit"]), n_rays, s),
              "bwd flat")
    torch.cuda.synchronize()
    # the scans associate differently (64-lane segmented scan vs per-ray scan): agreement to fp32 rounding
    for k, tol in (("w", 1e-6), ("tr", 1e-6), ("rgb", 2e-6), ("op", 2e-6), ("dp", 2e-6)):
        assert torch.allclose(a[k], b[k], rtol=2e-5, atol=tol), (k, (a[k] - b[k]).abs().max())
    for k in ("d_rgb", "d_logit"):
        scale = float(a[k].abs().max()) + 1e-12
        assert float((a[k] - b[k]).abs().max()) <= 3e-5 * scale, (k, float((a[k] - b[k]).abs().max()), scale)
    assert float(a["d_rgb"].abs().max()) > 0 or n == 0
    if mode != "upstream":
        assert float(a["acc"][1]) == float(b["acc"][1]) == float((a["op"] > 0).sum())
        assert abs(float(a["acc"][0]) - float(b["acc"][0])) <= 1e-5 * abs(float(a["acc"][0])) + 1e-7
    # rays without samples: background, opacity 0 -- written, not left over
    empty = packed[:, 1] == 0
    assert bool((b["op"][empty] == 0).all()) and bool((b["rgb"][empty] == bg).all()) and bool((b["dp"][empty] == 0).all())


def test_flat_compositing_overflowed_density_does_not_poison_the_ray():
    """exp(logit) = inf: T = 0 behind the sample (nerfacc's sequential loop), never NaN"""
    from nsr_hip import check, lib, ptr, stream_ptr
    packed = torch.tensor([[0, 100]], dtype=torch.int32).cuda()
    out1 = torch.zeros(100, 16).half().cuda()
    out1[40, 0] = 200.0
    out2 = torch.rand(100, 16).half().cuda()
    t0 = torch.arange(100).float().cuda() * 0.01
    t1 = t0 + 0.01
    bg = torch.zeros(3).cuda()
    w, tr = torch.zeros(100).cuda(), torch.zeros(100).cuda()
    rgb, op, dp = torch.zeros(1, 3).cuda(), torch.zeros(1).cuda(), torch.zeros(1).cuda()
    check(lib.nsr_composite_forward_flat(ptr(out1), 16, 0.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(bg), ptr(w), ptr(tr),
                                         ptr(rgb), ptr(op), ptr(dp), None, None, 1, stream_ptr()), "fwd flat")
    assert bool(torch.isfinite(w).all()) and bool(torch.isfinite(rgb).all()) and float(tr[41:].abs().max()) == 0.0


@pytest.mark.parametrize("n_rays,cap", [(8192, 0), (1147, 0), (8192, 30000), (5, 0), (9, 17)])
@pytest.mark.parametrize("nh", [1, 2])
def test_packing_folded_into_the_kept_row_copy(n_rays, cap, nh):
    """nsr_visibility_prefix_sums + nsr_nerf_copy_kept_rows_scan == nsr_visibility_prefix + nsr_pack_from_counts_capped +
    nsr_nerf_copy_kept_rows: kept counts, packed_info, total, statistics and every copied row bit for bit"""
    from nsr_hip import check, lib, ptr, stream_ptr
    packed_m, M, g = _packed(n_rays, 60, 7 + n_rays, long_every=11)
    Mc = max(M, 1) + 13  # capacity > live rows
    out1 = (torch.randn(Mc, 16, generator=g) * 2).half().cuda()
    acts = torch.rand(nh, Mc, 64, generator=g).half().cuda()
    enc = torch.randn(16, Mc, 2, generator=g).half().cuda()
    x01 = torch.rand(Mc, 3, generator=g).cuda()
    t0 = torch.rand(Mc, generator=g).cuda()
    t1 = t0 + 0.02
    rays_d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1).cuda()
    s = stream_ptr()
    S = cap if cap else Mc

    def run(fold):
        kept = torch.full((n_rays,), -1, dtype=torch.int32).cuda()
        packed2 = torch.full((n_rays, 2), -1, dtype=torch.int32).cuda()
        total = torch.full((1,), -1, dtype=torch.int32).cuda()
        stats = torch.zeros(8, dtype=torch.int32).cuda()
        o = dict(t0=torch.zeros(S).cuda(), t1=torch.zeros(S).cuda(), x01=torch.zeros(S, 3).cuda(),
                 enc=torch.zeros(16, S, 2).half().cuda(), out1=torch.zeros(S, 16).half().cuda(),
                 acts=torch.zeros(nh, S, 64).half().cuda(), ri=torch.full((S,), -1, dtype=torch.int64).cuda(),
                 tex=torch.zeros(S, 32).half().cuda())
        if fold:
            sums = torch.full(((n_rays + 7) // 8 + 4,), -1, dtype=torch.int32).cuda()
            check(lib.nsr_visibility_prefix_sums(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(packed_m), 1e-4, ptr(kept), ptr(sums),
                                                 n_rays, s), "vis sums")
            check(lib.nsr_nerf_copy_kept_rows_scan(ptr(packed_m), ptr(kept), ptr(sums), ptr(packed2), ptr(total), ptr(stats),
                                                   ptr(t0), ptr(t1), ptr(x01), ptr(enc), ptr(out1), ptr(acts), ptr(o["t0"]),
                                                   ptr(o["t1"]), ptr(o["x01"]), ptr(o["enc"]), ptr(o["out1"]), ptr(o["acts"]),
                                                   16, nh, Mc, S, ptr(rays_d), ptr(o["ri"]), ptr(o["tex"]), n_rays, s), "copy scan")
        else:
            check(lib.nsr_visibility_prefix(ptr(out1), 16, -1.0, ptr(t0), ptr(t1), ptr(packed_m), 1e-4, ptr(kept), n_rays, s), "vis")
            check(lib.nsr_pack_from_counts_capped(ptr(kept), ptr(packed2), ptr(total), n_rays, cap, ptr(stats), None, s), "pack")
            check(lib.nsr_nerf_copy_kept_rows(ptr(packed_m), ptr(packed2), ptr(t0), ptr(t1), ptr(x01), ptr(enc), ptr(out1),
                                              ptr(acts), ptr(o["t0"]), ptr(o["t1"]), ptr(o["x01"]), ptr(o["enc"]), ptr(o["out1"]),
                                              ptr(o["acts"]), 16, nh, Mc, S, ptr(rays_d), ptr(o["ri"]), ptr(o["tex"]), n_rays, s),
                  "copy")
        torch.cuda.synchronize()
        return dict(kept=kept, packed=packed2, total=total, stats=stats, **o)

    a, b = run(False), run(True)
    for k in a:
        assert torch.equal(a[k], b[k]), (k, a[k].flatten()[:8], b[k].flatten()[:8])
    assert int(a["total"]) > 0
    if cap:
        assert int(a["total"]) == cap and int(a["stats"][2]) == 1  # truncated, and reported


@pytest.mark.parametrize("nhc,nhd,n", [(2, 1, 100000), (2, 1, 4099), (1, 1, 777), (2, 2, 5000), (1, 2, 31), (2, 1, 7)])
def test_dgrad_pair_matches_the_two_launch_sequence(nhc, nhd, n):
    """nsr_mlp_dgrad_pair == colour dgrad (d_tex through HBM) + density dgrad: d_enc and the saved pre-activation gradients
    bit for bit; the weight gradients that follow agree to the float-atomic noise of their reduction"""
    import nsr_hip
    from nsr_hip import check, lib, ops, ptr, stream_ptr
    g = torch.Generator().manual_seed(n + nhc * 10 + nhd)
    dc = nsr_hip.make_mlp_desc(32, 3, nhc, "sigmoid")
    dd = nsr_hip.make_mlp_desc(32, 16, nhd, "none")
    assert lib.nsr_mlp_dgrad_pair_supported(ctypes.byref(dc), ctypes.byref(dd)) == 1
    npc, npd = 64 * 32 + (nhc - 1) * 4096 + 1024, 64 * 32 + (nhd - 1) * 4096 + 1024
    wc = (torch.randn(npc, generator=g) * 0.2).half().cuda()
    wd = (torch.randn(npd, generator=g) * 0.2).half().cuda()
    enc = torch.randn(16, n, 2, generator=g).half().cuda()  # level-major density input
    s = stream_ptr()
    out1 = torch.empty(n, 16).half().cuda()
    acts1 = torch.empty(nhd, n, 64).half().cuda()
    check(lib.nsr_mlp_forward_ex(ptr(enc), 0, 32, 2, ptr(wd), ptr(out1), ptr(acts1), n, ctypes.byref(dd), None, s), "fwd density")
    sh = torch.rand(n, 16, generator=g).half().cuda()
    tex_in = torch.cat([out1, sh], 1).contiguous()
    out2, acts2 = ops.mlp_forward(tex_in, wc, dc, save_acts=True)
    d_rgb = (torch.randn(n, 3, generator=g) * 1e-3).cuda()
    d_logit = (torch.randn(n, generator=g) * 1e-3).cuda()
    scale = 65536.0

    def ws(desc):
        return torch.zeros(int(lib.nsr_mlp_backward_workspace_floats(ctypes.byref(desc), n)), device="cuda")

    # (a) two launches
    pa_c, pa_d = ws(dc), ws(dd)
    ga_c, ga_d = torch.zeros(npc).cuda(), torch.zeros(npd).cuda()
    d_tex = torch.zeros(n, 32).cuda()
    d_enc_a = torch.zeros(16, n, 2).cuda()
    check(lib.nsr_mlp_backward_phases(ptr(d_rgb), 1, 3, None, ptr(out2), ptr(tex_in), 0, 32, 0, ptr(acts2), ptr(wc), ptr(ga_c),
                                      ptr(d_tex), 32, 0, ptr(pa_c), n, scale, ctypes.byref(dc), None, s, 3), "colour bwd")
    check(lib.nsr_mlp_backward_phases(ptr(d_tex), 1, 32, ptr(d_logit), ptr(out1), ptr(enc), 0, 32, 2, ptr(acts1), ptr(wd),
                                      ptr(ga_d), ptr(d_enc_a), 32, 2, ptr(pa_d), n, scale, ctypes.byref(dd), None, s, 3),
          "density bwd")
    # (b) pair + the weight-gradient halves
    pb_c, pb_d = ws(dc), ws(dd)
    gb_c, gb_d = torch.zeros(npc).cuda(), torch.zeros(npd).cuda()
    d_enc_b = torch.zeros(16, n, 2).cuda()
    check(lib.nsr_mlp_dgrad_pair(ptr(d_rgb), ptr(d_logit), ptr(out2), ptr(acts2), ptr(wc), ptr(pb_c), ptr(acts1), ptr(wd),
                                 ptr(pb_d), ptr(d_enc_b), n, scale, ctypes.byref(dc), ctypes.byref(dd), None, s), "pair")
    torch.cuda.synchronize()
    assert torch.equal(d_enc_a, d_enc_b)
    assert float(d_enc_a.abs().max()) > 0
    check(lib.nsr_mlp_backward_phases(ptr(d_rgb), 1, 3, None, ptr(out2), ptr(tex_in), 0, 32, 0, ptr(acts2), ptr(wc), ptr(gb_c),
                                      None, 32, 0, ptr(pb_c), n, scale, ctypes.byref(dc), None, s, 2), "colour wgrad")
    check(lib.nsr_mlp_backward_phases(ptr(d_enc_b), 1, 32, ptr(d_logit), ptr(out1), ptr(enc), 0, 32, 2, ptr(acts1), ptr(wd),
                                      ptr(gb_d), None, 32, 2, ptr(pb_d), n, scale, ctypes.byref(dd), None, s, 2), "density wgrad")
    torch.cuda.synchronize()
    for a, b in ((ga_c, gb_c), (ga_d, gb_d)):
        assert float((a - b).norm()) <= 1e-5 * float(a.norm()) + 1e-12, (float((a - b).norm()), float(a.norm()))
        assert float(a.abs().max()) > 0


def _adam_desc(st):
    from nsr_hip import NsrTableAdam
    d = NsrTableAdam()
    d.params, d.exp_avg, d.exp_avg_sq, d.shadow = st["p"].data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(), st["h"].data_ptr()
    d.step, d.hyper = st["step"].data_ptr(), st["hyper"].data_ptr()
    d.base_lr, d.beta1, d.beta2, d.gamma = 0.01, 0.9, 0.99, 0.33
    d.milestone0, d.milestone1, d.milestone2 = 2, 0x7fffffff, 0x7fffffff
    d.eps, d.weight_decay = 1e-15, 0.01
    return d


def _ray_ordered_positions(n, gen):
    """positions along rays through the unit cube, 64-256 samples per ray: what the step hands to the table backward"""
    xs = []
    left = n
    while left > 0:
        k = min(left, int(torch.randint(12, 256, (1,), generator=gen)))
        o = torch.rand(3, generator=gen)
        d = torch.nn.functional.normalize(torch.randn(3, generator=gen), dim=0)
        t = torch.arange(k).float() * (1.7 / 1024) + torch.rand(1, generator=gen) * 0.3
        xs.append((o[None] + d[None] * t[:, None]).clamp(0.0, 1.0))
        left -= k
    return torch.cat(xs)[:n].contiguous()


@pytest.mark.parametrize("n,mask", [(100000, 16), (30011, 16), (257, 16), (5000, 3), (63, 16)])
def test_dense_levels_through_merged_atomics_match_the_owner_workgroups(n, mask):
    """nsr_hashgrid_backward_params_dense (levels [0, D)) + the owner launch over [D, L) == the owner launch over all levels:
    hashed levels bit for bit (integer sums), dense levels to fp32 rounding (both add runs in fp32, in different orders); the
    same through AdamW; masked levels get exactly zero"""
    import nsr_hip
    from nsr_hip import check, lib, ptr, stream_ptr
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    D = int(lib.nsr_hashgrid_dense_levels(ctypes.byref(gd)))
    assert D == 5
    g = torch.Generator().manual_seed(n)
    x = _ray_ordered_positions(n, g).cuda()
    x[:7] = torch.round(x[:7])  # corners / faces of the box
    dy = (torch.randn(16, n, 2, generator=g) * 1e-3).cuda()
    n_tab = gd.n_entries * 2
    ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(gd), n)), device="cuda")
    s = stream_ptr()
    off = [int(o) * 2 for o in gd.offset[:17]]
    # (a) owner, all levels
    ga = torch.full((n_tab,), 7.0, device="cuda")
    check(lib.nsr_hashgrid_backward_params_owner_bin(ptr(x), ptr(ws), n, mask, ctypes.byref(gd), None, s), "bin")
    check(lib.nsr_hashgrid_backward_params_owner_accumulate(ptr(x), ptr(dy), 2, 0, ptr(ga), ptr(ws), n, mask, 1.0, 0,
                                                            ctypes.byref(gd), None, s), "accumulate")
    # (b) dense levels through the atomics path, the rest through a ranged owner launch
    gb = torch.full((n_tab,), 7.0, device="cuda")
    check(lib.nsr_hashgrid_backward_params_owner_bin_range(ptr(x), ptr(ws), n, mask, D, 16, ctypes.byref(gd), None, s), "bin range")
    check(lib.nsr_hashgrid_backward_params_dense(ptr(x), ptr(dy), ptr(gb), None, None, ptr(ws), n, mask, 1.0, 0, ctypes.byref(gd),
                                                 None, 7, s), "dense")
    check(lib.nsr_hashgrid_backward_params_owner_accumulate_range(ptr(x), ptr(dy), ptr(gb), None, ptr(ws), n, mask, 1.0, D, 16,
                                                                  ctypes.byref(gd), None, s), "accumulate range")
    torch.cuda.synchronize()
    for lvl in range(16):
        a, b = ga[off[lvl]:off[lvl + 1]], gb[off[lvl]:off[lvl + 1]]
        if lvl >= mask:
            assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0, lvl
        elif lvl >= D:
            assert torch.equal(a, b), lvl
        else:
            assert float((a - b).norm()) <= 2e-5 * float(a.norm()), (lvl, float((a - b).norm()), float(a.norm()))
            assert float(a.abs().max()) > 0
    # bf16 transport image of the dense levels
    gh = torch.zeros(n_tab, dtype=torch.bfloat16, device="cuda")
    check(lib.nsr_hashgrid_backward_params_dense(ptr(x), ptr(dy), None, ptr(gh), None, ptr(ws), n, mask, 1.0, 0, ctypes.byref(gd),
                                                 None, 7, s), "dense bf16")
    torch.cuda.synchronize()
    assert torch.equal(gh[:off[D]], gb[:off[D]].bfloat16())
    # a non-finite gradient is SEEN: the level it reached leaves as NaN
    dy2 = dy.clone()
    dy2[1, n // 2, 0] = float("inf")
    gc = torch.zeros(n_tab, device="cuda")
    check(lib.nsr_hashgrid_backward_params_dense(ptr(x), ptr(dy2), ptr(gc), None, None, ptr(ws), n, mask, 1.0, 0, ctypes.byref(gd),
                                                 None, 7, s), "dense inf")
    torch.cuda.synchronize()
    if mask > 1:
        assert bool(torch.isnan(gc[off[1]:off[2]]).all()) and bool(torch.isfinite(gc[off[0]:off[1]]).all())


def test_dense_levels_with_adamw_match_gradient_plus_optimizer():
    """the write-out of the dense path as AdamW == its gradient + nsr_adamw_step on those levels, bit for bit, over three
    steps (pow() start + running beta products); the ranged owner launch updates the hashed levels as before"""
    import nsr_hip
    from nsr_hip import check, lib, ops, ptr, stream_ptr
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    D = int(lib.nsr_hashgrid_dense_levels(ctypes.byref(gd)))
    n, n_tab = 30000, gd.n_entries * 2
    g = torch.Generator().manual_seed(3)
    ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(gd), n)), device="cuda")
    s = stream_ptr()

    def fresh():
        gg = torch.Generator(device="cuda").manual_seed(5)
        p = torch.randn(n_tab, device="cuda", generator=gg) * 0.1
        return dict(p=p, m=torch.zeros_like(p), v=torch.zeros_like(p), h=torch.empty(n_tab, dtype=torch.float16, device="cuda"),
                    step=torch.zeros(1, dtype=torch.int32, device="cuda"), hyper=torch.zeros(12, device="cuda"))

    a, b = fresh(), fresh()
    grad = torch.empty(n_tab, device="cuda")
    ms = (2, 0x7fffffff, 0x7fffffff)
    for it in range(3):
        x = _ray_ordered_positions(n, g).cuda()
        dy = (torch.randn(16, n, 2, generator=g) * 1e-3).cuda()
        check(lib.nsr_hashgrid_backward_params_owner_bin_range(ptr(x), ptr(ws), n, 16, D, 16, ctypes.byref(gd), None, s), "bin")
        # (a) gradient of every level (dense path + ranged owner), then the optimizer kernel
        check(lib.nsr_hashgrid_backward_params_dense(ptr(x), ptr(dy), ptr(grad), None, None, ptr(ws), n, 16, 1.0, 0,
                                                     ctypes.byref(gd), None, 7, s), "dense grad")
        check(lib.nsr_hashgrid_backward_params_owner_accumulate_range(ptr(x), ptr(dy), ptr(grad), None, ptr(ws), n, 16, 1.0, D, 16,
                                                                      ctypes.byref(gd), None, s), "range grad")
        ops.adam_tick(a["step"], a["hyper"], 0.01, 0.9, 0.99, 0.33, ms)
        ops.adamw_step(a["p"], grad, a["m"], a["v"], a["h"], 0.01, 0.9, 0.99, 1e-15, 0.01, it + 1, zero_grad=False,
                       hyper=a["hyper"])
        # (b) AdamW inside both write-outs
        d = _adam_desc(b)
        check(lib.nsr_hashgrid_backward_params_dense(ptr(x), ptr(dy), None, None, ctypes.byref(d), ptr(ws), n, 16, 1.0, 0,
                                                     ctypes.byref(gd), None, 7, s), "dense adam")
        check(lib.nsr_hashgrid_backward_params_owner_accumulate_adam_range(ptr(x), ptr(dy), ptr(ws), n, 16, 1.0, D, 16,
                                                                           ctypes.byref(gd), None, ctypes.byref(d), s), "range adam")
        ops.adam_tick(b["step"], b["hyper"], 0.01, 0.9, 0.99, 0.33, ms)
        torch.cuda.synchronize()
        off = [int(o) * 2 for o in gd.offset[:17]]
        for k in ("p", "m", "v", "h"):
            if not torch.equal(a[k], b[k]):
                bad = [(lvl, int((a[k][off[lvl]:off[lvl + 1]] != b[k][off[lvl]:off[lvl + 1]]).sum())) for lvl in range(16)]
                raise AssertionError((it, k, [t for t in bad if t[1]]))


def _model(seed=0):
    import nsr
    import refmirror
    torch.manual_seed(seed)
    cfg = nsr.configs.get("nerf-blender")
    model = refmirror.NeRFModel(cfg).cuda().train()
    with torch.no_grad():
        model.geometry.encoding_with_network.params[3072:].normal_(0, 0.08)
    model.randomized = False
    g = model.occupancy_grid
    ii = torch.stack(torch.meshgrid(*[torch.arange(128)] * 3, indexing="ij"), -1).float().cuda()
    c = (ii + 0.5) / 128 * 3 - 1.5
    g._binary = (c.norm(dim=-1) < 1.1)
    return model, cfg


def _rays(n, seed=1):
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 4.0
    d = torch.nn.functional.normalize(-o + torch.randn(n, 3, generator=g) * 0.5, dim=-1)
    return torch.cat([o, d], -1).cuda(), torch.rand(n, 3, generator=g).cuda()


def test_the_step_with_the_round5_forms_matches_the_step_without_them():
    """one fused step (native orchestration) with every nsr_nerf_step_variant off vs on: outputs to scan rounding, gradients of
    the hashed levels and of the networks to the noise of the fp16 chain's float-atomic weight-gradient reduction"""
    from nsr.fused import FusedNeRFStep
    from nsr_hip import lib
    model, cfg = _model()
    rays, gt = _rays(700)
    bg = torch.tensor([0.3, 0.6, 0.9], device="cuda")
    res, grads = {}, {}
    old = [lib.nsr_nerf_step_variant(k, -1) for k in range(4)]
    try:
        for on in (0, 1):
            for k in range(4):
                lib.nsr_nerf_step_variant(k, on)
            model.zero_grad(set_to_none=True)
            step = FusedNeRFStep(model, native=True)
            r = step.forward_backward(rays, gt, bg)
            torch.cuda.synchronize()
            res[on] = {k: r[k].clone() if torch.is_tensor(r[k]) else r[k] for k in ("comp_rgb", "opacity", "depth", "weights",
                                                                                    "ray_indices", "loss_acc", "num_samples")}
            grads[on] = (model.geometry.encoding_with_network.params.grad.clone(), model.texture.network.params.grad.clone())
    finally:
        for k in range(4):
            lib.nsr_nerf_step_variant(k, old[k])
    assert res[0]["num_samples"] == res[1]["num_samples"] > 0 and torch.equal(res[0]["ray_indices"], res[1]["ray_indices"])
    for k in ("comp_rgb", "opacity", "depth", "weights"):
        assert torch.allclose(res[0][k], res[1][k], rtol=2e-5, atol=2e-6), k
    assert abs(float(res[0]["loss_acc"][0]) - float(res[1]["loss_acc"][0])) <= 1e-5 * abs(float(res[0]["loss_acc"][0]))
    for a, b in zip(grads[0], grads[1]):
        cos = torch.nn.functional.cosine_similarity(a, b, dim=0)
        assert cos > 0.99999 and float((a - b).norm()) <= 2e-4 * float(a.norm()), (float(cos), float((a - b).norm() / a.norm()))


def test_async_trainer_with_and_without_the_round5_forms():
    """the asynchronous trainer (deferred packing, dense levels on their own stream, pair dgrad, flat compositing) against the
    same trainer with the forms off: same first loss, trajectories within the run-to-run noise of the step"""
    import nsr
    import refmirror
    from nsr.scene import SyntheticBlender
    from nsr.trainer import Trainer
    from nsr_hip import lib
    data = SyntheticBlender(n_images=6, w=80, h=80, device="cuda", seed=1)
    cfg = dict(nsr.configs.get("nerf-blender"))
    cfg["train_num_rays"], cfg["max_train_num_rays"] = 512, 2048
    out = {}
    old = [lib.nsr_nerf_step_variant(k, -1) for k in range(4)]
    try:
        for on in (0, 1):
            for k in range(4):
                lib.nsr_nerf_step_variant(k, on)
            torch.manual_seed(0)
            model = refmirror.NeRFModel(cfg).cuda().train()
            tr = Trainer(model, data, cfg, fused=True, seed=7, async_mode=True)
            tr.fused.defer_pack = bool(on)
            tr.defer_weights_wait = bool(on)
            lib.nsr_nerf_step_variant(4, on)  # (with it: the weight-gradient kernels behind the table backward)
            losses = [float(tr.train_step()["loss"]) for _ in range(48)]
            torch.cuda.synchronize()
            c = tr.counters()
            out[on] = dict(losses=losses, samples=c["samples"], rays=c["rays"], truncated=c["truncated"],
                           p=tr.fused.ewn.params.detach().clone())
    finally:
        for k in range(4):
            lib.nsr_nerf_step_variant(k, old[k])
        lib.nsr_nerf_step_variant(4, 0)
    a, b = out[0], out[1]
    assert abs(a["losses"][0] - b["losses"][0]) <= 1e-5 * abs(a["losses"][0])
    for x, y in zip(a["losses"], b["losses"]):
        assert abs(x - y) <= 3e-2 * abs(x) + 1e-6, (x, y)
    assert b["losses"][-1] < 0.7 * b["losses"][0] and a["truncated"] == b["truncated"] == 0
    assert abs(a["samples"] - b["samples"]) <= 0.02 * a["samples"]


def test_model_entry_lazy_outputs_match_the_synchronising_ones():
    """nsr.models.FusedNeRFModel with ``lazy_outputs``: the reference system's own statements (systems/nerf.py:87-99) on the
    non-synchronising outputs give the loss and the parameter gradients of the same statements on the eager outputs;
    ``num_samples.sum().item()`` is the previous forward's count, ``.current()`` this one's; per-sample outputs are sliced to
    the live count when read"""
    import nsr
    import nsr.models
    torch.manual_seed(0)
    cfg = nsr.configs.get("nerf-blender")
    model = nsr.models.FusedNeRFModel(cfg).cuda().train()
    with torch.no_grad():
        model.geometry.encoding_with_network.params[3072:].normal_(0, 0.08)
    model.randomized = False
    ii = torch.stack(torch.meshgrid(*[torch.arange(128)] * 3, indexing="ij"), -1).float().cuda()
    model.occupancy_grid._binary = (((ii + 0.5) / 128 * 3 - 1.5).norm(dim=-1) < 1.1)
    model.background_color = torch.tensor([0.3, 0.6, 0.9], device="cuda")
    res = {}
    for lazy in (False, True, True):
        rays, gt = _rays(700, seed=3 if lazy else 1)
        if len(res) == 2:
            rays, gt = _rays(700, seed=1)  # third pass: the eager pass's rays again, now lazily
        model.lazy_outputs = lazy
        model.zero_grad(set_to_none=True)
        out = model(rays)
        n_item = out["num_samples"].sum().item()
        valid = out["rays_valid"][..., 0]
        loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"][out["rays_valid"][..., 0]], gt[out["rays_valid"][..., 0]])
        loss.backward()
        torch.cuda.synchronize()
        res[len(res)] = dict(lazy=lazy, n_item=int(n_item), loss=float(loss), out=out, valid=valid,
                             current=(out["num_samples"].current() if lazy else int(n_item)),
                             g1=model.geometry.encoding_with_network.params.grad.clone(), g2=model.texture.network.params.grad.clone(),
                             weights=out["weights"].detach().clone(), ri=out["ray_indices"].clone())
    e, l1, l2 = res[0], res[1], res[2]
    assert type(l1["out"]).__name__ == "_LazyOutputs" and type(l1["valid"]).__name__ == "_ValidMask"
    assert l1["n_item"] == l1["current"] > 0            # first lazy forward: no predecessor, its own count
    assert l2["n_item"] == l1["current"]                # second: the PREVIOUS forward's count ...
    assert l2["current"] == e["n_item"]                 # ... while .current() is this forward's (same rays as the eager pass)
    assert abs(l2["loss"] - e["loss"]) <= 1e-6 * abs(e["loss"]) + 1e-9
    assert torch.equal(l2["ri"], e["ri"]) and l2["weights"].shape == e["weights"].shape
    assert torch.allclose(l2["weights"], e["weights"], rtol=1e-6, atol=1e-8)
    for k in ("g1", "g2"):
        assert float((l2[k] - e[k]).norm()) <= 2e-4 * float(e[k].norm()), (k, float((l2[k] - e[k]).norm() / e[k].norm()))
    # anything else done to a deferred selection gathers it (the reference's behaviour)
    sel = l2["out"]["comp_rgb"][l2["valid"]]
    assert sel.shape == (int(l2["valid"].sum()), 3) and torch.equal(sel.detach(), l2["out"]["comp_rgb"].detach()[l2["valid"].as_subclass(torch.Tensor)])
    assert model._runner().render_truncated == 0


@pytest.mark.parametrize("n", [100000, 4097, 5])
def test_encode_in_two_level_halves_equals_the_one_launch_encode(n):
    import nsr_hip
    from nsr_hip import check, lib, ptr, stream_ptr
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, 3, generator=g).cuda()
    table = (torch.randn(gd.n_entries * 2, generator=g) * 0.1).half().cuda()
    a = torch.full((16, n, 2), 7.0).half().cuda()
    b = torch.full((16, n, 2), -7.0).half().cuda()
    s = stream_ptr()
    check(lib.nsr_hashgrid_forward_ex(ptr(x), ptr(table), ptr(a), n, 32, 1, 16, ctypes.byref(gd), None, s), "fwd")
    for half in (1, 2):
        check(lib.nsr_hashgrid_forward_half(ptr(x), ptr(table), ptr(b), n, 32, 1, 16, half, ctypes.byref(gd), None, s), "half")
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_async_trainer_with_the_pipelined_encode_matches_without():
    """key 8 (two table-backward launches, the next encode's first level half beside the second) is scheduling only: the same
    trajectory as without, to the run-to-run noise of the step"""
    import nsr
    import refmirror
    from nsr.scene import SyntheticBlender
    from nsr.trainer import Trainer
    from nsr_hip import lib
    data = SyntheticBlender(n_images=6, w=80, h=80, device="cuda", seed=1)
    cfg = dict(nsr.configs.get("nerf-blender"))
    cfg["train_num_rays"], cfg["max_train_num_rays"] = 512, 2048
    out = {}
    try:
        for on in (0, 1):
            lib.nsr_nerf_step_variant(8, on)
            torch.manual_seed(0)
            model = refmirror.NeRFModel(cfg).cuda().train()
            tr = Trainer(model, data, cfg, fused=True, seed=7, async_mode=True)
            losses = [float(tr.train_step()["loss"]) for _ in range(48)]
            torch.cuda.synchronize()
            c = tr.counters()
            out[on] = dict(losses=losses, samples=c["samples"], truncated=c["truncated"])
    finally:
        lib.nsr_nerf_step_variant(8, 0)
    a, b = out[0], out[1]
    assert abs(a["losses"][0] - b["losses"][0]) <= 1e-5 * abs(a["losses"][0])
    for x, y in zip(a["losses"], b["losses"]):
        assert abs(x - y) <= 3e-2 * abs(x) + 1e-6, (x, y)
    assert b["losses"][-1] < 0.7 * b["losses"][0] and a["truncated"] == b["truncated"] == 0
    assert abs(a["samples"] - b["samples"]) <= 0.02 * a["samples"]


def test_async_trainer_with_the_table_backward_on_the_helper_stream_matches_without():
    """key 10 (table backward on the helper stream behind its binning, weight gradients and the MLP optimizer launch on the
    step's stream, one meeting of the two streams per step) is scheduling only: the pass reports the form, the trajectory is
    the one without it to the run-to-run noise of the step, checkpoints and evaluation see the finished table"""
    import nsr
    import refmirror
    from nsr.scene import SyntheticBlender
    from nsr.trainer import Trainer
    from nsr_hip import lib
    data = SyntheticBlender(n_images=6, w=80, h=80, device="cuda", seed=1)
    cfg = dict(nsr.configs.get("nerf-blender"))
    cfg["train_num_rays"], cfg["max_train_num_rays"] = 512, 2048
    out = {}
    try:
        for on in (0, 1):
            lib.nsr_nerf_step_variant(10, on)
            torch.manual_seed(0)
            model = refmirror.NeRFModel(cfg).cuda().train()
            tr = Trainer(model, data, cfg, fused=True, seed=7, async_mode=True)
            losses, forms = [], set()
            for _ in range(48):
                losses.append(float(tr.train_step()["loss"]))
                forms.add(int(lib.nsr_nerf_last_pass_form()))
            sd = tr.state_dict()  # (settles: the table update on the helper stream is behind it)
            torch.cuda.synchronize()
            c = tr.counters()
            table = model.geometry.encoding_with_network.params.detach().float().clone()
            out[on] = dict(losses=losses, samples=c["samples"], truncated=c["truncated"], forms=forms, table=table,
                           finite=all(bool(torch.isfinite(v).all()) for v in sd.values() if torch.is_tensor(v) and v.is_floating_point()))
    finally:
        lib.nsr_nerf_step_variant(10, 0)
    a, b = out[0], out[1]
    assert a["forms"] == {0} and b["forms"] == {1}
    assert abs(a["losses"][0] - b["losses"][0]) <= 1e-5 * abs(a["losses"][0])
    for x, y in zip(a["losses"], b["losses"]):
        assert abs(x - y) <= 3e-2 * abs(x) + 1e-6, (x, y)
    assert b["losses"][-1] < 0.7 * b["losses"][0] and a["truncated"] == b["truncated"] == 0
    assert abs(a["samples"] - b["samples"]) <= 0.02 * a["samples"]
    assert a["finite"] and b["finite"]
    # (no entry-wise comparison of the tables: with Adam's eps = 1e-15 a last-bit difference in a rarely-hit entry's gradient
    # moves it by a full learning-rate step -- the two runs differ in summation order of the weight-gradient partials)
    assert bool(torch.isfinite(b["table"]).all()) and float(b["table"].norm()) > 0


@pytest.mark.parametrize("kind", ["smooth_l1", "smooth_l1_beta", "mse", "l1", "huber"])
@pytest.mark.parametrize("n,ch", [(8192, 3), (777, 1), (5, 3)])
def test_masked_loss_of_deferred_selections_equals_the_loss_of_the_gathered_rows(kind, n, ch):
    """systems/nerf.py:97, systems/neus.py:98,102: ``F.<loss>(pred[valid], target[valid])`` on two deferred selections over one
    validity mask (nsr_masked_loss_forward / _backward) against the same statement on plain tensors: value and d / d pred"""
    import torch.nn.functional as F
    from nsr.models import _MaskedRows, _ValidMask
    fn, kw = {"smooth_l1": (F.smooth_l1_loss, {}), "smooth_l1_beta": (F.smooth_l1_loss, {"beta": 0.05}), "mse": (F.mse_loss, {}),
              "l1": (F.l1_loss, {}), "huber": (F.huber_loss, {"delta": 0.1})}[kind]
    g = torch.Generator().manual_seed(n + ch)
    shape = (n, ch) if ch > 1 else (n,)
    pred = torch.rand(shape, generator=g).cuda().requires_grad_(True)
    target = torch.rand(shape, generator=g).cuda()
    mask = (torch.rand(n, generator=g) < 0.6).cuda()
    ref = fn(pred[mask], target[mask], **kw) * 3.0
    g_ref, = torch.autograd.grad(ref, pred)
    valid = mask.clone().as_subclass(_ValidMask)
    a, b = pred[valid], target[valid]
    assert isinstance(a, _MaskedRows) and isinstance(b, _MaskedRows)
    out = fn(a, b, **kw) * 3.0
    assert out.grad_fn is not None and "MaskedLoss" in type(out.grad_fn.next_functions[0][0]).__name__
    g_out, = torch.autograd.grad(out, pred)
    assert abs(float(out) - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-9
    assert torch.allclose(g_out, g_ref, rtol=1e-5, atol=1e-9)
    assert torch.equal(g_out[~mask], torch.zeros_like(g_out[~mask]))
    # the fixed summation order: the same bits every time
    assert float(fn(pred[valid], target[valid], **kw)) == float(fn(pred[valid], target[valid], **kw))
    # nothing valid: 0 and a zero gradient (torch: NaN)
    none = torch.zeros(n, dtype=torch.bool).cuda().as_subclass(_ValidMask)
    z = fn(pred[none], target[none], **kw)
    gz, = torch.autograd.grad(z, pred)
    assert float(z) == 0.0 and not gz.any()


def test_neus_model_entry_masks_defer_the_selections():
    """nsr.models.FusedNeuSModel in training: ``rays_valid_full`` is a _ValidMask, the system's MSE / L1 statements
    (systems/neus.py:98,102) give the values and the parameter gradients of the same statements on plain masks"""
    import torch.nn.functional as F
    from test_gpu_models_entry import _neus_pair
    _, model, cfg = _neus_pair("neus-blender", 7001)
    g = torch.Generator().manual_seed(1)
    o = torch.nn.functional.normalize(torch.randn(300, 3, generator=g), dim=-1) * 4.0
    d = torch.nn.functional.normalize(-o + torch.randn(300, 3, generator=g) * 0.45, dim=-1)
    rays, gt = torch.cat([o, d], -1).cuda(), torch.rand(300, 3, generator=g).cuda()
    res = []
    for plain in (False, True):
        model.zero_grad(set_to_none=True)
        out = model(rays)
        valid = out["rays_valid_full"][..., 0]
        assert type(valid).__name__ == "_ValidMask"
        if plain:
            valid = valid.as_subclass(torch.Tensor)
        mse = F.mse_loss(out["comp_rgb_full"][valid], gt[valid])
        l1 = F.l1_loss(out["comp_rgb_full"][valid], gt[valid])
        (10.0 * mse + l1).backward()
        torch.cuda.synchronize()
        res.append((float(mse), float(l1), [p.grad.clone() for p in model.parameters() if p.grad is not None and p.numel()]))
    (m0, a0, g0), (m1, a1, g1) = res
    assert 0 < int(out["rays_valid_full"].sum()) < 300
    assert abs(m0 - m1) <= 2e-6 * abs(m1) and abs(a0 - a1) <= 2e-6 * abs(a1)
    assert len(g0) == len(g1) > 0
    for x, y in zip(g0, g1):
        assert float((x - y).norm()) <= 2e-4 * float(y.norm()) + 1e-12
