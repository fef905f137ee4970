"""Mint instant-nsr-pl_amd/nsr/configs.json = the resolved ``model:`` sections of the reference's YAMLs (build container
only; the reference tree is absent on the GPU box).  tests/test_configs.py re-checks the committed file here."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import refshim  # noqa: E402
from oracle import nerfacc_ref, tcnn_ref  # noqa: E402

YAMLS = {"nerf-blender": ("nerf-blender.yaml", ["dataset.scene=lego"]),
         "neus-blender": ("neus-blender.yaml", ["dataset.scene=lego"]),
         "neus-dtu": ("neus-dtu.yaml", ["dataset.root_dir=unused"]),
         "neuralangelo": ("neuralangelo-dtu-wmask.yaml", ["dataset.root_dir=unused"])}


def resolved():
    refshim.install(tcnn_ref, nerfacc_ref)
    try:
        return {k: refshim._unwrap(refshim.load_config(y, cli).model) for k, (y, cli) in YAMLS.items()}
    finally:
        refshim.uninstall()


if __name__ == "__main__":
    out = os.path.join(ROOT, "instant-nsr-pl_amd", "nsr", "configs.json")
    json.dump(resolved(), open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)
