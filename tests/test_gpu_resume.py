"""Checkpoint = model + training state (reference launch.py --resume / Lightning's checkpoint): a trainer rebuilt
from ``save()`` continues exactly where the saved one stood -- optimizer moments, step counts (host and device side), dynamic
ray count -- and ``model.load_state_dict()`` on a model a trainer was already built around is not silently reverted by the
optimizer's fp16 images."""
import io

import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(async_mode, seed=42):
    import nsr
    from nsr.scene import SyntheticBlender
    from nsr.trainer import Trainer
    torch.manual_seed(0)
    cfg = nsr.configs.get("nerf-blender")
    model = nsr.build(cfg).cuda().train()
    data = SyntheticBlender(n_images=6, w=64, h=64, device="cuda", seed=0)
    return Trainer(model, data, cfg, seed=seed, async_mode=async_mode), model


@pytest.mark.parametrize("async_mode", [False, True], ids=["sync", "async"])
def test_trainer_resumes_from_its_checkpoint(async_mode):
    tr, model = _trainer(async_mode)
    for _ in range(20):
        tr.train_step()
    torch.cuda.synchronize()
    buf = io.BytesIO()
    tr.save(buf)
    ck = torch.load(io.BytesIO(buf.getvalue()), map_location="cuda")
    ts = ck["training_state"]
    assert ts["global_step"] == 20 and ts["optimizer"]["step_count"] == 20
    ewn = model.geometry.encoding_with_network
    want_m = tr.opt.state[ewn.params][0].clone()
    want_p = ewn.params.detach().clone()
    assert float(want_m.abs().max()) > 0  # the table's first moment (also when AdamW runs inside the table backward)
    # a fresh trainer (different initial weights, no optimizer history) resumes from the file
    tr2, model2 = _trainer(async_mode, seed=7)
    with torch.no_grad():
        model2.geometry.encoding_with_network.params.mul_(0.5)
    tr2.load(ck)
    ewn2 = model2.geometry.encoding_with_network
    assert tr2.global_step == 20 and tr2.opt.step_count == 20 and tr2.train_num_rays == ts["train_num_rays"]
    assert torch.equal(ewn2.params.detach(), want_p) and torch.equal(tr2.opt.state[ewn2.params][0], want_m)
    assert torch.equal(ewn2.half_params(ewn2.params), want_p.half())  # the fp16 image the kernels read follows the loaded weights
    # ... and trains on: the optimizer continues at step 21 (bias corrections of step 21, not of step 1)
    tr2.train_step()
    torch.cuda.synchronize()
    assert tr2.global_step == 21 and tr2.opt.step_count == 21
    step1 = (ewn2.params.detach() - want_p).abs().max()
    assert 0 < float(step1) < 0.05  # (a first-ever AdamW step would move every touched entry by lr = 0.01 at once: ~5x the resumed one)
    if async_mode and getattr(tr2.opt, "_step_dev", None) is not None:
        assert int(tr2.opt._step_dev.item()) == 21  # the device-side counter the fused table update reads


def test_load_state_dict_after_the_trainer_was_built_is_not_reverted():
    tr, model = _trainer(True)
    for _ in range(3):
        tr.train_step()
    torch.cuda.synchronize()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    key = "geometry.encoding_with_network.params"
    sd[key] = torch.full_like(sd[key], 0.25)
    model.load_state_dict(sd)
    ewn = model.geometry.encoding_with_network
    assert torch.equal(ewn.half_params(ewn.params), torch.full_like(sd[key], 0.25).half())
    tr.train_step()
    torch.cuda.synchronize()
    # one AdamW step away from the loaded value (lr 0.01), not back at the pre-load weights (|w| ~ 1e-4)
    assert float((ewn.params.detach() - 0.25).abs().max()) < 0.03
