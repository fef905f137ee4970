import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "instant-nsr-pl_amd"))
import torch
import tinycudann as tcnn
cfg = dict(otype="HashGrid", n_levels=4, n_features_per_level=2, log2_hashmap_size=12, base_resolution=8, per_level_scale=2.0)
enc = tcnn.Encoding(3, cfg)
with torch.no_grad():
    enc.params.normal_(0, 0.05)
x = torch.rand(500, 3, device="cuda")
y = enc(x)
y.float().sum().backward()
print("A plain backward: grad abs sum", float(enc.params.grad.abs().sum()))
enc.params.grad = None
x = torch.rand(500, 3, device="cuda").requires_grad_(True)
y = enc(x).float()
s = y.sum(-1)
(g,) = torch.autograd.grad(s, x, torch.ones_like(s), create_graph=True)
print("B after autograd.grad: params.grad", enc.params.grad)
loss = s.mean()
loss.backward(retain_graph=True)
print("C direct path only: grad abs sum", float(enc.params.grad.abs().sum()))
enc.params.grad = None
(g.norm(dim=-1) ** 2).mean().backward(retain_graph=True)
print("D eikonal path only: grad abs sum", float(enc.params.grad.abs().sum()))
enc.params.grad = None
(s.mean() + (g.norm(dim=-1) ** 2).mean()).backward()
print("E both: grad abs sum", float(enc.params.grad.abs().sum()))
