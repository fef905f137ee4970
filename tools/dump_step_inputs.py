"""Capture what the table backward of a REAL training step sees -- the kept samples' unit-cube positions and the level-major
gradient d L / d enc the density MLP's dgrad hands over -- at the two operating points of bench.py (the start-up transient and
the steady state), for tools/table_backward_variants.py (NSR_VARIANT_DATA).  Synthetic ray-coherent positions spread over the
whole volume; a trained scene concentrates its samples on the surface (hot cells, ragged dense-level slices).

    python tools/dump_step_inputs.py gpurun_out/step_inputs.pt [steady_step [transient_step]]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer

if __name__ == "__main__":
    out = sys.argv[1]
    steady = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    transient = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    dev = torch.device("cuda", 0)
    torch.manual_seed(42)
    cfg = nsr.configs.get("nerf-blender")
    model = nsr.build(cfg).to(dev).train()
    data = SyntheticBlender(n_images=100, w=800, h=800, device=dev, seed=0)
    tr = Trainer(model, data, cfg, seed=42, async_mode=True)
    tr.fuse_table_update = False  # (keeps d_enc / x01 of the last step untouched in the workspace either way)
    res = {}
    for step in range(steady + 1):
        tr.train_step()
        if step in (transient, steady):
            torch.cuda.synchronize()
            ab = tr.fused._ab
            L, ws = ab["ML"], ab["ws"]
            s_cap = ab["key"][2]
            S = int(tr._as["total_kept"].item())
            x01 = ws[L.x01:L.x01 + s_cap * 12].view(torch.float32).view(s_cap, 3)[:S].clone()
            d_enc = ws[L.d_enc:L.d_enc + 16 * s_cap * 8].view(torch.float32).view(16, s_cap, 2)[:, :S].clone()
            res["transient" if step == transient else "steady"] = {"x": x01.cpu(), "dy": d_enc.cpu(), "step": step, "n": S}
            print(step, "kept", S, "s_cap", s_cap, "x range", float(x01.min()), float(x01.max()), "dy norm", float(d_enc.norm()))
    torch.save(res, out)
