"""oracle/vanilla_ref.py (the CPU baseline of bench.py: the reference's pure-PyTorch encoding + MLP path) against the
fixture minted from the reference modules, and against the reference modules themselves when the tree is present."""
import os

import pytest
import torch

from oracle import vanilla_ref
from test_golden_glue import load


def test_vanilla_frequency_matches_reference_fixture():
    fx = load("vanilla_frequency.npz")
    x = fx["x"]
    assert torch.equal(vanilla_ref.VanillaFrequency(3, {"n_frequencies": 6})(x), fx["plain"])
    enc = vanilla_ref.VanillaFrequency(3, {"n_frequencies": 6, "n_masking_step": 1000})
    for step in (0, 250, 700, 5000):
        enc.update_step(0, step)
        assert torch.equal(enc(x), fx[f"masked_{step}"]), step
    comp = vanilla_ref.VanillaFrequency(3, {"n_frequencies": 4})
    assert torch.equal(vanilla_ref.include_xyz(comp, x), fx["composite"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference tree only exists in the build container")
def test_vanilla_mlp_matches_reference_module():
    import refshim
    from oracle import nerfacc_ref, tcnn_ref
    refshim.install(tcnn_ref, nerfacc_ref)
    try:
        from models.network_utils import VanillaMLP
        cfg = {"n_neurons": 64, "n_hidden_layers": 2, "output_activation": "none"}
        torch.manual_seed(0)
        ref = VanillaMLP(60, 4, cfg)
        mine = vanilla_ref.VanillaMLP(60, 4, 64, 2)
        mine.layers.load_state_dict(ref.layers.state_dict())
        x = torch.randn(100, 60)
        assert torch.equal(mine(x), ref(x))
    finally:
        refshim.uninstall()
