"""Record / replay of everything that crosses the drop-in boundary.  TEST INFRASTRUCTURE.

RECORD (build container, ``tests/gen_golden.py:gen_boundary_traces``): the reference's OWN ``models.make('nerf'|'neus',
cfg)`` built from its real YAMLs runs one training forward + backward on recording wrappers around the CPU oracle's
``tinycudann`` / ``nerfacc``; every call that crosses the package boundary is logged in order -- constructor arguments,
input tensors, output tensors, every gradient that autograd delivers to an output (``gy``), every gradient it asks back
for an input (``gx``), and, for double backward (models/geometry.py:177-180 with ``create_graph=True``), the gradient
that later arrives AT a first-order input gradient (``ggx``).

REPLAY (GPU box, ``tests/test_gpu_boundary_trace.py``): the same constructor calls on the HIP packages, parameters
re-generated from the recorded seeds, then call by call: recorded inputs in, outputs compared; recorded ``gy`` pushed
back, ``gx`` compared; parameter gradients accumulated over the whole trace and compared with the summary of what the
reference run left in ``.grad``.
"""
import json

import numpy as np
import torch

import fixture_utils as fu


# ------------------------------------------------------------------------------------------------------------------
# recording
# ------------------------------------------------------------------------------------------------------------------
class Recorder:
    def __init__(self):
        self.calls, self.modules, self.tensors = [], [], {}

    def _put(self, key, t):
        self.tensors[key] = t.detach().cpu().clone()

    # ---- tcnn ------------------------------------------------------------------------------------------------
    def wrap_tcnn(self, T):
        rec = self

        def make(base):
            class Rec(base):
                def __init__(self, *a, **k):
                    super().__init__(*a, **k)
                    self._tid = len(rec.modules)
                    rec.modules.append({"cls": base.__name__, "args": list(a), "kwargs": k,
                                        "n_params": int(self.params.numel()), "segments": []})

                def forward(self, x):
                    c = len(rec.calls)
                    info = {"kind": "tcnn", "mod": self._tid, "grad": bool(torch.is_grad_enabled()),
                            "x_req": bool(x.requires_grad and torch.is_grad_enabled()), "n_gy": 0, "n_gx": 0, "n_ggx": 0,
                            "n_ggy": 0}
                    rec.calls.append(info)
                    xa = x.clone() if info["x_req"] else x
                    y = super().forward(xa)
                    rec._put(f"c{c}/x", x)
                    rec._put(f"c{c}/y", y)
                    if y.requires_grad:
                        def hy(g, info=info, c=c):
                            k = info["n_gy"]
                            rec._put(f"c{c}/gy{k}", g)
                            info["n_gy"] += 1
                            if g.requires_grad:  # d(sdf)/d(encoding): differentiated again by the eikonal term
                                def hg(gg, info=info, c=c, k=k):
                                    rec._put(f"c{c}/ggy{k}", gg)
                                    info["n_ggy"] += 1
                                g.register_hook(hg)
                        y.register_hook(hy)
                    if info["x_req"]:
                        def hx(g, info=info, c=c):
                            k = info["n_gx"]
                            rec._put(f"c{c}/gx{k}", g)
                            info["n_gx"] += 1
                            if g.requires_grad:  # first-order input gradient that is differentiated again
                                def hg(gg, info=info, c=c, k=k):
                                    rec._put(f"c{c}/ggx{k}", gg)
                                    info["n_ggx"] += 1
                                g.register_hook(hg)
                        xa.register_hook(hx)
                    return y
            Rec.__name__ = base.__name__
            return Rec

        import types
        m = types.ModuleType("tinycudann")
        m.Encoding, m.Network, m.NetworkWithInputEncoding = make(T.Encoding), make(T.Network), make(T.NetworkWithInputEncoding)
        m.free_temporary_memory = T.free_temporary_memory
        return m

    # ---- nerfacc ---------------------------------------------------------------------------------------------
    def wrap_nerfacc(self, N):
        rec = self
        import types
        m = types.ModuleType("nerfacc")
        for name in dir(N):
            if not name.startswith("_"):
                setattr(m, name, getattr(N, name))

        def fn_wrapper(name, tensor_kw):
            fn = getattr(N, name)

            def wrapped(*args, **kw):
                c = len(rec.calls)
                info = {"kind": "nerfacc", "fn": name, "args": [], "kw": {}, "n_gy": 0, "gx": []}
                rec.calls.append(info)

                def enc(v, slot):
                    if isinstance(v, torch.Tensor):
                        rec._put(f"c{c}/{slot}", v)
                        req = bool(v.requires_grad and torch.is_grad_enabled())
                        va = v.clone() if req else v
                        if req:
                            def h(g, slot=slot):
                                rec._put(f"c{c}/gx_{slot}", g)
                                info["gx"].append(slot)
                            va.register_hook(h)
                        return va, {"t": slot, "req": req}
                    if isinstance(v, N.OccupancyGrid):
                        rec.tensors[f"c{c}/{slot}_binary"] = torch.from_numpy(np.packbits(v._binary.numpy().reshape(-1)))
                        rec._put(f"c{c}/{slot}_roi", v._roi_aabb)
                        return v, {"grid": slot, "res": [int(s) for s in v._binary.shape],
                                   "contraction": v._contraction_type.name}
                    if callable(v):
                        def cb(*a, **k):
                            r = v(*a, **k)
                            rec._put(f"c{c}/{slot}_ret", r)
                            return r
                        return cb, {"callback": slot}
                    return v, {"v": v}

                a2, k2 = [], {}
                for i, v in enumerate(args):
                    vv, d = enc(v, f"a{i}")
                    a2.append(vv)
                    info["args"].append(d)
                for k, v in kw.items():
                    vv, d = enc(v, f"k_{k}")
                    k2[k] = vv
                    info["kw"][k] = d
                out = fn(*a2, **k2)
                outs = out if isinstance(out, (tuple, list)) else (out,)
                info["n_out"] = len(outs)
                for j, o in enumerate(outs):
                    rec._put(f"c{c}/o{j}", o)
                    if o.requires_grad:
                        def hy(g, j=j):
                            rec._put(f"c{c}/gy_o{j}", g)
                            info["n_gy"] += 1
                        o.register_hook(hy)
                return out
            return wrapped

        for name in ("ray_marching", "render_weight_from_density", "render_weight_from_alpha", "accumulate_along_rays"):
            setattr(m, name, fn_wrapper(name, None))
        inter = types.ModuleType("nerfacc.intersection")
        inter.ray_aabb_intersect = fn_wrapper("ray_aabb_intersect", None)
        m.intersection = inter
        m.ray_aabb_intersect = inter.ray_aabb_intersect
        return m

    def seed_parameters(self, model, plan):
        """overwrite every large tcnn parameter from seeds: ``plan(module_info, module) -> [(count, seed, std), ...]``"""
        for mod in model.modules():
            tid = getattr(mod, "_tid", None)
            if tid is None or mod.params.numel() == 0:
                continue
            segs = plan(self.modules[tid], mod)
            flat = torch.cat([fu.seeded_normal(n, seed, std) for n, seed, std in segs])
            assert flat.numel() == mod.params.numel()
            with torch.no_grad():
                mod.params.copy_(flat)
            self.modules[tid]["segments"] = [[int(n), int(seed), float(std)] for n, seed, std in segs]

    def finish(self, model, extra=None):
        """-> dict for np.savez: tensors + JSON meta; parameter gradients of the tcnn modules as summaries"""
        for mod in model.modules():
            tid = getattr(mod, "_tid", None)
            if tid is None or mod.params.numel() == 0 or mod.params.grad is None:
                continue
            self.tensors.update(fu.pack_summary(f"m{tid}/gradsum", fu.grad_summary(mod.params.grad)))
        meta = {"modules": self.modules, "calls": self.calls, "extra": extra or {}}

        def clean(o):
            if isinstance(o, dict):
                return {k: clean(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [clean(v) for v in o]
            if isinstance(o, (int, float, str, bool)) or o is None:
                return o
            if hasattr(o, "name") and hasattr(o, "value"):  # enums
                return {"enum": o.name}
            return str(o)
        out = {k: v.numpy() for k, v in self.tensors.items()}
        out["meta"] = np.frombuffer(json.dumps(clean(meta)).encode(), dtype=np.uint8)
        return out


# ------------------------------------------------------------------------------------------------------------------
# replay
# ------------------------------------------------------------------------------------------------------------------
def load_trace(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}


def build_modules(meta, tcnn, device):
    """the recorded constructor calls on the given tinycudann package, parameters from the recorded seeds"""
    mods = []
    for info in meta["modules"]:
        with torch.cuda.device(device):
            m = getattr(tcnn, info["cls"])(*info["args"], **info["kwargs"])
        assert int(m.params.numel()) == info["n_params"], (info["cls"], m.params.numel(), info["n_params"])
        if info["segments"]:
            flat = torch.cat([fu.seeded_normal(n, seed, std) for n, seed, std in info["segments"]])
            with torch.no_grad():
                m.params.copy_(flat.to(m.params.device))
            if hasattr(m, "invalidate"):
                m.invalidate()
        mods.append(m.train())
    return mods
