"""Forms of the NeRF step's kernels against the forms they replace: the packing folded into the kept-row copy vs scan + copy,
both networks' data gradients in one kernel vs two launches, and the whole step with every switchable form off / on
(reference models/nerf.py:95-109, systems/nerf.py:97, models/network_utils.py:181,209).  The compositing pair is pinned to the
oracle directly in tests/test_gpu_composite_samples.py, the data / weight gradients in tests/test_gpu_mlp_pair_oracle.py."""
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
    # ... and DIRECTLY against the definition (nerfacc's pack_info semantics, reference models/nerf.py:95-103): the kept counts
    # are the oracle's transmittance cut, packed_info is their exclusive cumsum (clamped at the capacity), the rows a ray keeps
    # are the first `kept` rows of its marched segment, ray_indices name it
    from oracle import nerfacc_ref as R
    pm = packed_m.cpu().long()
    ri_m = torch.repeat_interleave(torch.arange(n_rays), pm[:, 1])
    sigma = torch.exp(out1[:M, 0].float().cpu() - 1.0).view(-1, 1)
    alpha = 1.0 - torch.exp(-sigma * 0.02)
    vis = R.render_visibility(alpha, ray_indices=ri_m, early_stop_eps=1e-4) if M else torch.zeros(0, dtype=torch.bool)
    want_kept = torch.zeros(n_rays, dtype=torch.long).index_add_(0, ri_m, vis.long())
    got_kept = b["kept"].cpu().long()
    # (fp32 prefix products against the oracle's fp64: a sample exactly at the 1e-4 cut may fall on either side)
    assert int((got_kept - want_kept).abs().max()) <= 1 and int((got_kept != want_kept).sum()) <= max(1, n_rays // 500)
    starts = torch.cumsum(got_kept, 0) - got_kept
    limit = cap if cap else S
    want_start = starts.clamp(max=limit)
    want_count = (starts + got_kept).clamp(max=limit) - want_start
    pk = b["packed"].cpu().long()
    assert torch.equal(pk[:, 0], want_start) and torch.equal(pk[:, 1], want_count)
    assert int(b["total"]) == int(want_count.sum())
    src = torch.cat([pm[r, 0] + torch.arange(int(want_count[r])) for r in range(n_rays)]) if int(want_count.sum()) else torch.zeros(0, dtype=torch.long)
    tot = int(want_count.sum())
    assert torch.equal(b["t0"][:tot].cpu(), t0.cpu()[src]) and torch.equal(b["out1"][:tot].cpu(), out1.cpu()[src])
    assert torch.equal(b["ri"][:tot].cpu(), torch.repeat_interleave(torch.arange(n_rays), want_count))


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


def test_the_step_with_the_current_forms_matches_the_step_with_the_round4_forms():
    """one fused step (native orchestration) with every nsr_nerf_step_variant off vs on: outputs to scan rounding, gradients of
    the hashed levels and of the networks to the noise of the fp16 chain's float-atomic weight-gradient reduction"""
    from nsr.fused import FusedNeRFStep
    from nsr_hip import lib
    model, cfg = _model()
    rays, gt = _rays(700)
    bg = torch.tensor([0.3, 0.6, 0.9], device="cuda")
    res, grads = {}, {}
    KEYS = (0, 2, 5)  # pair dgrad, sample-partitioned compositing, fork events riding on kernels
    old = [lib.nsr_nerf_step_variant(k, -1) for k in KEYS]
    try:
        for on in (0, 1):
            for k in KEYS:
                lib.nsr_nerf_step_variant(k, on)
            model.zero_grad(set_to_none=True)
            step = FusedNeRFStep(model, native=True)
            r = step.forward_backward(rays, gt, bg)
            torch.cuda.synchronize()
            res[on] = {k: r[k].clone() if torch.is_tensor(r[k]) else r[k] for k in ("comp_rgb", "opacity", "depth", "weights",
                                                                                    "ray_indices", "loss_acc", "num_samples")}
            grads[on] = (model.geometry.encoding_with_network.params.grad.clone(), model.texture.network.params.grad.clone())
    finally:
        for k, v in zip(KEYS, old):
            lib.nsr_nerf_step_variant(k, v)
    assert res[0]["num_samples"] == res[1]["num_samples"] > 0 and torch.equal(res[0]["ray_indices"], res[1]["ray_indices"])
    for k in ("comp_rgb", "opacity", "depth", "weights"):
        assert torch.allclose(res[0][k], res[1][k], rtol=2e-5, atol=2e-6), k
    assert abs(float(res[0]["loss_acc"][0]) - float(res[1]["loss_acc"][0])) <= 1e-5 * abs(float(res[0]["loss_acc"][0]))
    for a, b in zip(grads[0], grads[1]):
        cos = torch.nn.functional.cosine_similarity(a, b, dim=0)
        assert cos > 0.99999 and float((a - b).norm()) <= 2e-4 * float(a.norm()), (float(cos), float((a - b).norm() / a.norm()))


def test_async_trainer_with_and_without_the_current_forms():
    """the asynchronous trainer (deferred packing, pair dgrad, sample-partitioned compositing, riding events) against the
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
    KEYS = (0, 2, 5)
    old = [lib.nsr_nerf_step_variant(k, -1) for k in KEYS]
    try:
        for on in (0, 1):
            for k in KEYS:
                lib.nsr_nerf_step_variant(k, on)
            torch.manual_seed(0)
            model = refmirror.NeRFModel(cfg).cuda().train()
            tr = Trainer(model, data, cfg, fused=True, seed=7, async_mode=True)
            tr.fused.defer_pack = bool(on)
            tr.defer_weights_wait = bool(on)
            losses = [float(tr.train_step()["loss"]) for _ in range(48)]
            torch.cuda.synchronize()
            c = tr.counters()
            out[on] = dict(losses=losses, samples=c["samples"], rays=c["rays"], truncated=c["truncated"],
                           p=tr.fused.ewn.params.detach().clone())
    finally:
        for k, v in zip(KEYS, old):
            lib.nsr_nerf_step_variant(k, v)
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
    assert l1["n_item"] == e["n_item"] and l1["current"] > 0   # first lazy forward: the count of the (eager) forward before it
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
