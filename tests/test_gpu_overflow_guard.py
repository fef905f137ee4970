"""Lightning's `precision: 16` protocol (reference configs/nerf-blender.yaml:103: torch.cuda.amp.GradScaler -- a step whose
gradients hold inf / NaN is SKIPPED, the loss scale halves, and doubles again after growth_interval clean steps) inside the
asynchronous fused NeRF trainer, on the device: csrc/mlp.hip k_mlp_dgrad_pair raises the step's flag, the table backward's
fused AdamW and the scheduled AdamW launch skip, the latter updates the scale (csrc/util.hip k_adamw_scheduled)."""
import struct

import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(seed=42):
    import nsr
    from nsr.scene import SyntheticBlender
    from nsr.trainer import Trainer
    torch.manual_seed(0)
    cfg = nsr.configs.get("nerf-blender")
    model = nsr.build(cfg).cuda().train()
    data = SyntheticBlender(n_images=6, w=64, h=64, device="cuda", seed=0)
    return Trainer(model, data, cfg, seed=seed, async_mode=True), model


def _snapshot(tr, model):
    tr.settle()
    torch.cuda.synchronize()
    ewn, tex = model.geometry.encoding_with_network, model.texture.network
    out = [ewn.params.detach().clone(), tex.params.detach().clone(), ewn.half_params(ewn.params).clone()]
    for m in (ewn, tex):
        out += [t.clone() for t in tr.opt.state[m.params][:2]]
    return out


def test_an_overflowing_step_is_skipped_the_scale_backs_off_and_training_goes_on():
    tr, model = _trainer()
    losses = []
    for _ in range(40):
        tr.train_step()
    st = tr.overflow_guard_stats()
    assert st == {"scale": 65536.0, "clean_steps": 40, "skipped_steps": 0}, st
    losses.append(float(tr.last["loss"]))
    before = _snapshot(tr, model)
    step_dev = int(tr.opt._step_dev.item())
    # an inf in dL/dy: the loss scale itself is made too large for fp16 (what GradScaler's own tests do)
    g = tr._overflow_guard_state()
    g[2:3] = torch.tensor([struct.unpack("<i", struct.pack("<f", 2.0 ** 60))[0]], dtype=torch.int32, device=g.device)
    tr.train_step()
    after = _snapshot(tr, model)
    for a, b in zip(before, after):  # weights, fp16 image, both moments of both modules: bit for bit untouched
        assert torch.equal(a, b)
    st = tr.overflow_guard_stats()
    assert st["skipped_steps"] == 1 and st["scale"] == 2.0 ** 59 and st["clean_steps"] == 0, st
    assert int(tr.opt._step_dev.item()) == step_dev  # the optimizer's step count did not advance (GradScaler.step)
    # the scale keeps halving until the chain fits fp16 again, then training goes on
    for _ in range(120):
        tr.train_step()
    st2 = tr.overflow_guard_stats()
    # (the scale settles just under what fp16 carries: later steps with larger gradients overflow again now and then and are
    # skipped too -- GradScaler's own dynamics; clean_steps counts from the last one)
    assert 1 < st2["skipped_steps"] < 60 and 1.0 <= st2["scale"] < 2.0 ** 45, st2
    moved = _snapshot(tr, model)
    assert not torch.equal(moved[0], after[0]) and bool(torch.isfinite(moved[0]).all()) and bool(torch.isfinite(moved[1]).all())
    # growth: from a small scale, doubled after every growth_interval clean steps
    bits = lambda v: struct.unpack("<i", struct.pack("<f", v))[0]  # noqa: E731
    g[2:6] = torch.tensor([bits(1024.0), 0, st2["skipped_steps"], 50], dtype=torch.int32, device=g.device)
    for _ in range(160):
        tr.train_step()
    st3 = tr.overflow_guard_stats()
    assert st3["scale"] == 1024.0 * 8 and st3["skipped_steps"] == st2["skipped_steps"] and st3["clean_steps"] == 10, st3
    losses.append(float(tr.last["loss"]))
    assert losses[1] < losses[0], losses
    assert all(bool(torch.isfinite(t).all()) for t in _snapshot(tr, model))


def test_guard_off_leaves_the_fixed_scale_path_alone():
    tr, model = _trainer()
    tr.overflow_guard = False
    for _ in range(10):
        tr.train_step()
    assert tr.overflow_guard_stats() is None
    tr.settle()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(model.geometry.encoding_with_network.params).all())


def test_guarded_and_unguarded_steps_agree_while_nothing_overflows():
    """same seeds, one trainer after the other: with the scale at its initial value the guarded trainer takes the same steps as the
    fixed-scale one, up to the run-to-run noise of the weight gradients' float atomics (measured here between two fixed-scale
    runs)"""
    def run(guard):
        tr, m = _trainer()
        tr.overflow_guard = guard
        for _ in range(25):
            tr.train_step()
        tr.settle()
        torch.cuda.synchronize()
        return m.geometry.encoding_with_network.params.detach().clone(), float(tr.last["loss"])

    (pa, la), (pb, lb), (pc, lc) = run(True), run(False), run(False)
    noise = float((pb - pc).norm() / pc.norm())
    rel = float((pa - pb).norm() / pb.norm())
    assert rel <= max(3.0 * noise, 1e-4), (rel, noise)
    assert abs(la - lb) <= max(3.0 * abs(lb - lc), 0.02 * abs(lb)), (la, lb, lc)


@pytest.mark.parametrize("name", ["neus-blender", "neus-dtu"])
def test_neus_trainer_skips_a_step_with_a_non_finite_loss_gradient(name):
    """the same protocol in the fused NeuS trainer (nsr/fused_neus.py NeuSTrainer._overflow_guard_state): target colours of a
    few rays are made inf for ONE step -- every parameter (hash tables, fp16 colour network / fp32 heads, weight-norm
    tensors, variance), every optimizer moment and the optimizers' step counts come out of that step bit-unchanged, the scale
    halves, the host's copy of the scale follows within a few steps, and training goes on"""
    import nsr
    from nsr.fused_neus import NeuSTrainer
    from nsr.scene import SyntheticBlender
    torch.manual_seed(0)
    cfg = nsr.configs.get(name)
    model = nsr.build(cfg).cuda().train()
    data = SyntheticBlender(n_images=8, h=64, w=64, device="cuda", environment=bool(cfg["learned_background"]), seed=0)
    tr = NeuSTrainer(model, data, cfg, {"lambda_rgb_l1": 1.0, "lambda_rgb_mse": 0.0, "lambda_eikonal": 0.1},
                     config_name=name)
    for _ in range(3):
        assert tr.train_step()["n_samples"] > 0
    st = tr.overflow_guard_stats()
    assert st == {"scale": 65536.0, "clean_steps": 3, "skipped_steps": 0, "optimizer_steps": 3}, st

    def snapshot():
        torch.cuda.synchronize()
        out = [p.detach().clone() for p in model.parameters()]
        for m in tr.opt.tcnn_modules:
            out += [t.clone() for t in tr.opt.state[m.params][:3]]
        out += [tr.opt_rest._m.clone(), tr.opt_rest._v.clone(), tr.opt._step_dev.clone(), tr.opt_rest._dev_state[0].clone()]
        return out

    assert tr._pending is not None  # (step 3's rays were prepared ahead: the test reaches them here)
    torch.cuda.synchronize()
    tr._pending[1][:4] = float("inf")
    before = snapshot()
    tr.train_step()
    after = snapshot()
    for k, (a, b) in enumerate(zip(before, after)):
        assert torch.equal(a, b), (k, float((a.float() - b.float()).abs().max()))
    st = tr.overflow_guard_stats()
    assert st == {"scale": 32768.0, "clean_steps": 0, "skipped_steps": 1, "optimizer_steps": 3}, st
    for _ in range(4):
        tr.train_step()
    torch.cuda.synchronize()
    st = tr.overflow_guard_stats()
    assert st == {"scale": 32768.0, "clean_steps": 4, "skipped_steps": 1, "optimizer_steps": 7}, st
    assert tr.fused.grad_scale == 32768.0
    moved = snapshot()
    assert all(bool(torch.isfinite(t.float()).all()) for t in moved)
    assert not torch.equal(moved[0], after[0])
    # growth: after `overflow_growth_interval` clean steps the scale doubles again
    tr.overflow_growth_interval = 0  # (read when the state is created: set the device word directly)
    tr._guard[5] = 6
    for _ in range(3):
        tr.train_step()
    st = tr.overflow_guard_stats()
    assert st["scale"] == 65536.0 and st["clean_steps"] <= 1 and st["skipped_steps"] == 1, st
