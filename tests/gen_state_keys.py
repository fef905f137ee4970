"""Mint tests/golden/state_dict_keys.json: the state-dict key set and tensor shapes of the REFERENCE's models
(``models.make`` of /root/reference, imported unchanged through tests/refshim.py on the CPU oracle's tinycudann / nerfacc
stand-ins) built from the real YAMLs.  tests/test_gpu_export.py checks ``nsr.state.HotPathState`` against it.

    PYTHONDONTWRITEBYTECODE=1 python tests/gen_state_keys.py          (build container only: needs /root/reference)
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import refshim  # noqa: E402
from oracle import nerfacc_ref, tcnn_ref  # noqa: E402

CASES = (("nerf-blender", "nerf", "nerf-blender.yaml", ["dataset.scene=lego"]),
         ("neus-blender", "neus", "neus-blender.yaml", ["dataset.scene=lego"]),
         ("neus-dtu", "neus", "neus-dtu.yaml", ["dataset.scene=scan24"]),
         ("neuralangelo", "neus", "neuralangelo-dtu-wmask.yaml", ["dataset.scene=scan24"]))


def main():
    models = refshim.install(tcnn_ref, nerfacc_ref)
    out = {}
    try:
        for key, name, yaml_name, cli in CASES:
            cfg = refshim.load_config(yaml_name, cli)
            torch.manual_seed(0)
            m = models.make(name, cfg.model)
            out[key] = {k: list(v.shape) for k, v in sorted(m.state_dict().items())}
            print(key, len(out[key]))
    finally:
        refshim.uninstall()
    json.dump(out, open(os.path.join(HERE, "golden", "state_dict_keys.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
