"""The sigma pass of the NeRF step on the ray sets of a trained model: the ray-ordered kernel that stops at the transmittance
cut (nsr_sigma_rays) against encode + MLP + visibility prefix over every marched sample, isolated, same inputs.
    python tools/sigma_rays_bench.py [train_steps]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
import torch
import nsr
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer
from nsr_hip import check, lib, ptr, stream_ptr
from kernel_microbench import median_us

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 700
dev = torch.device("cuda", 0)
torch.manual_seed(42)
cfg = nsr.configs.get("nerf-blender")
model = nsr.build(cfg).to(dev).train()
data = SyntheticBlender(n_images=24, w=400, h=400, device=dev, seed=0)
tr = Trainer(model, data, cfg, seed=42, async_mode=True)
for _ in range(steps):
    tr.train_step()
torch.cuda.synchronize()
a = tr._async_state()
fused = tr.fused
ewn, d = fused.ewn, fused.desc
half = ewn.half_params(ewn.params)
table, w1 = half[ewn.n_network_params:].clone(), half[:ewn.n_network_params].clone()
res = {"steps": steps, "cases": []}
for back in (1, 2):
    rs = a["sets3"][(tr.global_step - back) % a["window"]]
    # re-write the sample arrays of this ring slot from its marching scratch (they were consumed by the step)
    fused.write_async(rs)
    torch.cuda.synchronize()
    mb = rs["marched"]
    m_cap, slots = rs["m_cap"], rs["slots"]
    M = int(rs["total"].item())
    x01, t0, t1, packed = mb["x01"], mb["t0"], mb["t1"], rs["packed"]
    L, F = int(d.grid.n_levels), int(d.grid.n_features)
    C = L * F
    enc = torch.empty(L * m_cap * F, dtype=torch.float16, device=dev)
    out1 = torch.empty(m_cap * 16, dtype=torch.float16, device=dev)
    acts = torch.empty(m_cap * 64, dtype=torch.float16, device=dev)
    kept = torch.empty(slots, dtype=torch.int32, device=dev)
    kept2 = torch.empty(slots, dtype=torch.int32, device=dev)
    gd, md = ctypes.byref(d.grid), ctypes.byref(d.mlp_density)

    def trio():
        s = stream_ptr()
        check(lib.nsr_hashgrid_forward_ex(ptr(x01), ptr(table), ptr(enc), m_cap, C, 1, L, gd, ptr(rs["total"]), s), "enc")
        check(lib.nsr_mlp_forward_ex(ptr(enc), 0, C, F, ptr(w1), ptr(out1), ptr(acts), m_cap, md, ptr(rs["total"]), s), "mlp")
        check(lib.nsr_visibility_prefix(ptr(out1), 16, fused.bias, ptr(t0), ptr(t1), ptr(packed), fused.eps, ptr(kept), slots, s), "vis")

    def rays():
        check(lib.nsr_sigma_rays(ptr(x01), ptr(table), ptr(w1), ptr(out1), ptr(acts), ptr(enc), m_cap, ptr(packed), ptr(t0),
                                 ptr(t1), fused.bias, fused.eps, ptr(kept2), slots, gd, md, stream_ptr()), "sigma_rays")
    trio(); rays(); torch.cuda.synchronize()
    case = {"marched": M, "kept": int(kept.sum()), "same_kept_counts": bool(torch.equal(kept, kept2)),
            "rays_with_samples": int((packed[:, 1] > 0).sum()), "trio_us": round(median_us(trio, 10, 40), 1), "rays_us": {}}
    for blocks in (256, 512, 768, 1024, 2048, 8192):
        lib.nsr_sigma_rays_blocks(blocks)
        case["rays_us"][blocks] = round(median_us(rays, 10, 40), 1)
    lib.nsr_sigma_rays_blocks(768)
    res["cases"].append(case)
print(json.dumps(res))
