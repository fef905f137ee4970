"""Fused NeRF training step: the same computation as ``NeRFModel.forward_`` + smooth-L1 loss + backward
(reference models/nerf.py:61-127, systems/nerf.py:87-99), issued as ~45 kernel launches with hand-chained
backward instead of ~250 launches through autograd.  It reads the parameters of an ordinary ``NeRFModel``
(same ``state_dict``) and writes the gradients into their ``.grad``; parity with the modular path is tested in
``tests/test_gpu_fused.py``.

Three entry points: ``forward_backward`` (reads its two sample counts back to the host), ``forward_backward_async`` +
``march_async`` / ``pack_async`` (every count stays on the device, fixed-capacity buffers: no host sync) and
``refresh_occupancy_async`` (the occupancy refresh of models/nerf.py:45-55 without ``torch.nonzero``).
"""
import ctypes

import torch

from nerfacc import ContractionType
from nsr_hip import check, device_guard, lib, ptr, stream_ptr
from nsr_hip import ops as _ops

F32, F16 = torch.float32, torch.float16
_byref = ctypes.byref


class _PendingCount:
    """the (marched, kept) sample counts of a lazy ``render_forward``: ``previous()`` -- the last call's, already on the host (None
    before the second call); ``current()`` -- this call's, waiting for its pinned copy"""

    def __init__(self, bb):
        self._bb, self._pending, self._prev = bb, bb["pending"], bb["prev_counts"]

    def previous(self):
        return self._prev

    def current(self):
        host, ev = self._pending[:2]
        ev.synchronize()
        return int(host[0]), int(host[8])


class FusedNeRFStep:
    def __init__(self, model, early_stop_eps=1e-4, grad_scale=None, native=True):
        # the fp16 MLP backward scales dL/dy before it rounds it to fp16 (tcnn: loss_scale 128 ON TOP of Lightning's
        # GradScaler(65536), configs/nerf-blender.yaml:103 -- 2^23 together).  The fused step has no GradScaler; with 128 alone
        # the late-training gradients (loss ~1e-4 over ~25,000 colour values) sit at the fp16 subnormal edge
        if grad_scale is None:
            import os
            grad_scale = float(os.environ.get("NSR_GRAD_SCALE", "65536"))
        self.native = native  # True: one C call per phase (csrc/step.hip); False: every launch issued from Python
        cfg = model.config
        if cfg["learned_background"] or not cfg["grid_prune"]:
            raise NotImplementedError("FusedNeRFStep covers the bounded (AABB + occupancy grid) nerf-blender path")
        self.model = model
        self.ewn = model.geometry.encoding_with_network        # tcnn.NetworkWithInputEncoding
        self.tex = model.texture.network                       # tcnn.Network (sigmoid output)
        if not hasattr(self.ewn, "grid_desc") or not hasattr(self.tex, "mlp_desc"):
            raise NotImplementedError("FusedNeRFStep needs the fused tcnn geometry / texture modules")
        self.radius = float(cfg["radius"])
        self.bias = float(cfg["geometry"].get("density_bias", 0.0))
        assert cfg["geometry"].get("density_activation") == "trunc_exp"
        self.eps = float(early_stop_eps)
        self.grad_scale = float(grad_scale)
        import nsr_hip
        self.desc = nsr_hip.NsrNerfStepDesc(self.ewn.grid_desc, self.ewn.mlp_desc, self.tex.mlp_desc, self.radius,
                                            ContractionType.AABB.value, self.bias, self.eps, self.grad_scale, 1.0)
        self._PL, self._ML = nsr_hip.NsrNerfPruneLayout(), nsr_hip.NsrNerfMainLayout()
        import os
        self.kept_rows_event = not os.environ.get("NSR_PRUNED_EVENT")  # A/B switch: a torch event behind the pruning pass instead
        # packing of the kept counts folded into the main pass's kept-row copy (nsr_nerf_prune_pass_deferred): used wherever a
        # main pass follows the pruning pass on the same stream with nothing reading packed_kept / total in between
        self.defer_pack = not os.environ.get("NSR_PACK_SEPARATE")

    # ---- small launch helpers (all on torch's current stream) -------------------------------------------------
    def _positions(self, rays_o, rays_d, ri, t0, t1, want_dirs):
        n = ri.shape[0]
        x01 = torch.empty((n, 3), dtype=F32, device=ri.device)
        dirs = torch.empty((n, 3), dtype=F32, device=ri.device) if want_dirs else None
        check(lib.nsr_sample_positions_unit(ptr(rays_o), ptr(rays_d), ptr(ri), ptr(t0), ptr(t1), self.radius,
                                            ContractionType.AABB.value, ptr(x01), ptr(dirs), n, None, stream_ptr()),
              "nsr_sample_positions_unit")
        return x01, dirs

    def march_begin(self, rays_o, rays_d, t_min=None, t_max=None):
        """slab test + (stratified jitter) + marching pass, enqueued on the current stream with no host sync"""
        m = self.model
        grid = m.occupancy_grid
        if t_min is None:
            t_min, t_max = _ops.ray_aabb_intersect(rays_o, rays_d, m.scene_aabb)
            if m.randomized:
                t_min = t_min + torch.rand_like(t_min) * m.render_step_size
        return _ops.ray_march_begin(rays_o, rays_d, t_min, t_max, grid.roi_aabb, grid.binary,
                                    ContractionType.AABB.value, m.render_step_size, 0.0, roi_host=grid._roi_host)

    def march_and_prune(self, rays_o, rays_d, keep_rows, handle=None, after_prune=None, before_sync=None):
        """ray_marching(..., sigma_fn) of models/nerf.py:82-93.  With ``keep_rows`` the sigma pass saves encodings and
        activations of ALL marched samples and the pruning copies the kept rows: the main pass re-encodes nothing.
        -> dict(packed, ri, t0, t1, M, [x01, dirs, enc, out1, acts1])"""
        if handle is None:
            handle = self.march_begin(rays_o, rays_d)
        packed, ri, t0, t1 = _ops.ray_march_finish(handle)
        n_rays, M = rays_o.shape[0], ri.shape[0]
        if M == 0:
            if before_sync is not None:
                before_sync(None)
            if after_prune is not None:
                after_prune(0)
            return dict(packed=packed, ri=ri, t0=t0, t1=t1, M=0)
        ewn = self.ewn
        table, w = ewn.table_half(ewn.params), ewn.weights_half(ewn.params)
        x01, _ = self._positions(rays_o, rays_d, ri, t0, t1, False)
        enc = _ops.hashgrid_forward(x01, table, ewn.grid_desc)
        out, acts = _ops.mlp_forward(enc, w, ewn.mlp_desc, save_acts=keep_rows)
        dev = ri.device
        kept = torch.empty(n_rays, dtype=torch.int32, device=dev)
        packed2 = torch.empty((n_rays, 2), dtype=torch.int32, device=dev)
        total = torch.zeros(1, dtype=torch.int32, device=dev)
        s = stream_ptr()
        check(lib.nsr_visibility_prefix(ptr(out), out.stride(0), self.bias, ptr(t0), ptr(t1), ptr(packed), self.eps,
                                        ptr(kept), n_rays, s), "nsr_visibility_prefix")
        check(lib.nsr_pack_from_counts(ptr(kept), ptr(packed2), ptr(total), n_rays, s), "nsr_pack_from_counts")
        if before_sync is not None:
            before_sync(total)
        S = int(total.item())  # second (and last) host sync of the step
        if after_prune is not None:
            after_prune(S)  # e.g. the trainer launches the NEXT step's marching on a side stream right here
        ri2 = torch.empty(S, dtype=torch.int64, device=dev)
        t0b, t1b = torch.empty((S, 1), dtype=F32, device=dev), torch.empty((S, 1), dtype=F32, device=dev)
        res = dict(packed=packed2, ri=ri2, t0=t0b, t1=t1b, M=M)
        srcs, dsts = [t0, t1], [t0b, t1b]
        dirs = None
        if keep_rows:
            res["x01"] = torch.empty((S, 3), dtype=F32, device=dev)
            res["enc"] = torch.empty((S, enc.shape[1]), dtype=F16, device=dev)
            res["out1"] = torch.empty((S, out.shape[1]), dtype=F16, device=dev)
            res["acts1"] = torch.empty((acts.shape[0], S, 64), dtype=F16, device=dev)
            dirs = res["dirs"] = torch.empty((S, 3), dtype=F32, device=dev)
            srcs += [x01, enc, out] + [acts[h] for h in range(acts.shape[0])]
            dsts += [res["x01"], res["enc"], res["out1"]] + [res["acts1"][h] for h in range(acts.shape[0])]
        n = len(srcs)
        sp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
        dp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dsts])
        rb = (ctypes.c_uint32 * n)(*[t.stride(0) * t.element_size() for t in srcs])
        check(lib.nsr_copy_ray_prefix_rows(ptr(packed), ptr(packed2), n, sp, dp, rb, ptr(rays_d), ptr(dirs), ptr(ri2),
                                           n_rays, s), "nsr_copy_ray_prefix_rows")
        return res

    def forward_backward(self, rays, gt_rgb, background, compute_grads=True, loss_scale=1.0, march_handle=None,
                         after_prune=None, after_enqueue=None, before_sync=None):
        """hooks: ``before_sync(total)`` runs right after the pruning pass is QUEUED (``total``: int32[1] device tensor
        that will hold the kept-sample count, or None when nothing was marched) and before the host waits for it;
        ``after_prune(S)`` runs once the host knows the count."""
        if self.native:
            return self._forward_backward_native(rays, gt_rgb, background, compute_grads, loss_scale, march_handle,
                                                 after_prune, before_sync)
        return self._forward_backward_python(rays, gt_rgb, background, compute_grads, loss_scale, march_handle,
                                             after_prune, after_enqueue, before_sync)

    def _forward_backward_native(self, rays, gt_rgb, background, compute_grads, loss_scale, march_handle, after_prune,
                                 before_sync=None):
        """the same step with ONE C call per phase (nsr_nerf_prune_pass / nsr_nerf_main_pass, csrc/step.hip)"""
        m, ewn, tex, d = self.model, self.ewn, self.tex, self.desc
        dev = rays.device
        n_rays = rays.shape[0]
        d.loss_scale = float(loss_scale)
        with torch.no_grad(), torch.cuda.device(dev):
            with _ops.timed("fused:march_prune"):
                if march_handle is None:
                    rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()
                    march_handle = self.march_begin(rays_o, rays_d)
                rays_o, rays_d = march_handle.args[0], march_handle.args[1]
                packed, ri, t0, t1 = _ops.ray_march_finish(march_handle)
                M = ri.shape[0]
                s = stream_ptr()
                half = ewn.half_params(ewn.params)
                table, w1, w2 = half[ewn.n_network_params:], half[:ewn.n_network_params], tex.half_params(tex.params)
                check(lib.nsr_nerf_prune_layout(_byref(d), M, _byref(self._PL)), "nsr_nerf_prune_layout")
                pws = torch.empty(max(int(self._PL.total_bytes), 256), dtype=torch.uint8, device=dev)
                meta = torch.empty(3 * n_rays + 1, dtype=torch.int32, device=dev)  # kept | packed_kept | total
                kept, packed2, total = meta[:n_rays], meta[n_rays:3 * n_rays].view(n_rays, 2), meta[3 * n_rays:]
                if M > 0:
                    check(lib.nsr_nerf_prune_pass(_byref(d), ptr(rays_o), ptr(rays_d), ptr(ri), ptr(t0), ptr(t1),
                                                  ptr(packed), ptr(table), ptr(w1), ptr(pws), ptr(kept), ptr(packed2),
                                                  ptr(total), M, n_rays, None, 0, None, None, s), "nsr_nerf_prune_pass")
                    if before_sync is not None:
                        before_sync(total)
                    S = _ops.read_count_when_ready(total)  # second (and last) host sync of the step
                else:
                    meta.zero_()
                    if before_sync is not None:
                        before_sync(None)
                    S = 0
                if after_prune is not None:
                    after_prune(S)
            with _ops.timed("fused:main_pass"):
                check(lib.nsr_nerf_main_layout(_byref(d), S, n_rays, _byref(self._ML)), "nsr_nerf_main_layout")
                L = self._ML
                ws = torch.empty(int(L.total_bytes), dtype=torch.uint8, device=dev)
                bg = background.to(F32).contiguous()
                gt = gt_rgb.to(F32).contiguous()
                if compute_grads:
                    for p in (ewn.params, tex.params):
                        if p.grad is None:
                            p.grad = torch.zeros_like(p)
                g1 = ewn.params.grad if compute_grads else None
                g2 = tex.params.grad if compute_grads else None
                check(lib.nsr_nerf_main_pass(_byref(d), ptr(pws), M, ptr(packed), ptr(packed2), ptr(t0), ptr(t1),
                                             ptr(rays_d), ptr(bg), ptr(gt), ptr(w1), ptr(w2),
                                             ptr(ewn.mlp_slice(g1)) if compute_grads else None,
                                             ptr(ewn.grid_slice(g1)) if compute_grads else None,
                                             ptr(g2) if compute_grads else None, ptr(ws), S, n_rays,
                                             int(bool(compute_grads)), None, None, None, s), "nsr_nerf_main_pass")
            def view(off, n, dtype, shape):
                return ws[off:off + n * dtype.itemsize].view(dtype).view(shape)

            opacity = view(L.opacity, n_rays, F32, (n_rays, 1))
            return {"comp_rgb": view(L.comp_rgb, n_rays * 3, F32, (n_rays, 3)), "opacity": opacity,
                    "depth": view(L.depth, n_rays, F32, (n_rays, 1)), "rays_valid": opacity > 0, "num_samples": S,
                    "num_marched": M, "weights": view(L.weights, S, F32, (S,)),
                    "ray_indices": view(L.ray_indices, S, torch.int64, (S,)),
                    "t_starts": view(L.t_starts, S, F32, (S, 1)), "t_ends": view(L.t_ends, S, F32, (S, 1)),
                    "loss_acc": view(L.loss_acc, 2, F32, (2,)), "_workspace": ws}

    # ---- the step split at the loss (nsr.models.FusedNeRFModel: the reference's system owns loss and backward()) ---------
    def render_forward(self, rays, background, prepare_backward, lazy=False):
        """march + sigma pass + main forward of ``NeRFModel.forward_`` (models/nerf.py:61-127) queued as one run of launches
        with ONE host synchronisation at its end: the marched / kept sample counts stay on the device, the buffers have
        capacities that follow the counts of the previous calls (a call whose counts exceed them is re-queued with larger
        ones from the marcher's scratch rows).  Returns the reference's output tensors plus the state ``render_backward``
        needs (everything lives in two workspaces).

        ``lazy``: NO synchronisation -- the counts are copied to pinned memory behind the pass and read by the NEXT call (which
        sizes its buffers from them, one call late, like the asynchronous trainer: a call whose counts exceed its capacities is
        truncated and reported in ``self.render_truncated``); ``out['num_samples']`` / ``['num_marched']`` are then None and
        ``out['count']`` is a handle (``.current()`` waits for this call's counts, ``.previous()`` returns the last call's), the
        per-sample outputs are capacity-sized (rows >= the live count are unspecified)."""
        m, ewn, tex, d = self.model, self.ewn, self.tex, self.desc
        grid = m.occupancy_grid
        dev = rays.device
        n_rays = rays.shape[0]
        bb = getattr(self, "_bb", None)
        if bb is None or bb["slots"] < n_rays or bb["dev"] != dev:
            cap = int(lib.nsr_ray_march_capacity((ctypes.c_float * 6)(*[float(v) for v in grid._roi_host]),
                                                 float(m.render_step_size)))
            slots = max(n_rays, 1024)
            bb = self._bb = dict(slots=slots, dev=dev, cap=cap, counts=torch.empty(slots, dtype=torch.int32, device=dev),
                                 scratch=torch.empty(slots * cap * 2, dtype=F32, device=dev),
                                 stats=torch.zeros(16, dtype=torch.int32, device=dev),
                                 host=torch.zeros(16, dtype=torch.int32).pin_memory(), m_cap=1 << 19, s_cap=1 << 18,
                                 hosts=[torch.zeros(16, dtype=torch.int32).pin_memory() for _ in range(2)], pending=None,
                                 prev_counts=None, flip=0)
            self.render_truncated = 0
        rx, ry, rz = (int(v) for v in grid.binary.shape)
        with torch.no_grad(), torch.cuda.device(dev):
            s = stream_ptr()
            rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()
            t_min, t_max = _ops.ray_aabb_intersect(rays_o, rays_d, m.scene_aabb)
            if m.randomized:
                t_min = t_min + torch.rand_like(t_min) * m.render_step_size
            bricks = _ops.grid_bricks(grid.binary)
            if bricks is None:
                raise NotImplementedError("the fused NeRF step needs a brick-able occupancy grid (resolution % 16 == 0)")
            roi, step = grid.roi_aabb, float(m.render_step_size)
            check(lib.nsr_ray_march_bricks_count(ptr(rays_o), ptr(rays_d), ptr(t_min), ptr(t_max), ptr(roi), ptr(bricks), rx, ry,
                                                 rz, ContractionType.AABB.value, step, 0.0, ptr(bb["counts"]),
                                                 ptr(bb["scratch"]), bb["cap"], n_rays, s), "nsr_ray_march_bricks_count")
            half = ewn.half_params(ewn.params)
            table, w1, w2 = half[ewn.n_network_params:], half[:ewn.n_network_params], tex.half_params(tex.params)
            bg = background.to(F32).contiguous()
            import copy
            if lazy and bb["pending"] is None:
                # the first lazy call has no counts to size its buffers from: it takes the synchronising path (which re-queues
                # until everything fits) and leaves its counts for the next one (fixed initial capacities truncated)
                lazy = False
            if lazy:
                # the counts of the PREVIOUS lazy call (its pass finished long ago: the host is one step behind at most)
                host_p, ev_p, caps_p = bb["pending"]
                ev_p.synchronize()
                Mp, Sp = int(host_p[0]), int(host_p[8])
                # (what that call's buffers held is what it rendered and what the system is told)
                bb["prev_counts"], bb["pending"] = (min(Mp, caps_p[0]), min(Sp, caps_p[1])), None
                # (capacities follow the counts one call late: more head-room than the synchronising path's 1.3 x)
                grow = lambda n: -(-int(1.6 * n) // 65536) * 65536  # noqa: E731
                if Mp > 0.75 * bb["m_cap"] or Mp < 0.3 * bb["m_cap"]:
                    bb["m_cap"] = max(grow(max(Mp, 1)), 1 << 18)
                if Sp > 0.75 * bb["s_cap"] or Sp < 0.3 * bb["s_cap"]:
                    bb["s_cap"] = max(grow(max(Sp, 1)), 1 << 17)
                if Mp > caps_p[0] or Sp > caps_p[1]:
                    # that call's samples were cut at its capacities (a jump of the dynamic ray count, a grid refresh): its
                    # rays past the cut got no samples.  Say so once, and take THIS call through the synchronising path so
                    # that two steps in a row cannot be truncated
                    self.render_truncated += 1
                    if self.render_truncated == 1:
                        import warnings
                        warnings.warn(f"FusedNeRFStep.render_forward(lazy): a call was cut at its sample capacity (marched {Mp} "
                                      f"> {caps_p[0]} or kept {Sp} > {caps_p[1]}); this call synchronises, capacities grown")
                    lazy = False
            while True:
                m_cap, s_cap = bb["m_cap"], bb["s_cap"]
                meta = torch.empty(5 * n_rays + 2, dtype=torch.int32, device=dev)  # packed | kept | packed_kept | totals
                packed, kept = meta[:2 * n_rays].view(n_rays, 2), meta[2 * n_rays:3 * n_rays]
                packed2, total_m, total_s = meta[3 * n_rays:5 * n_rays].view(n_rays, 2), meta[5 * n_rays:5 * n_rays + 1], \
                    meta[5 * n_rays + 1:]
                stats = bb["stats"]
                check(lib.nsr_pack_from_counts_capped(ptr(bb["counts"]), ptr(packed), ptr(total_m), n_rays, m_cap,
                                                      ptr(stats[0:8]), None, s), "nsr_pack_from_counts_capped")
                smp = torch.empty(m_cap * 4, dtype=torch.int32, device=dev)  # ray_indices (i64) | t_starts | t_ends
                ri, t0, t1 = smp[:2 * m_cap].view(torch.int64), smp[2 * m_cap:3 * m_cap].view(F32), smp[3 * m_cap:].view(F32)
                check(lib.nsr_ray_march_bricks_write(ptr(rays_o), ptr(rays_d), ptr(t_min), ptr(t_max), ptr(roi), None, rx, ry, rz,
                                                     ContractionType.AABB.value, step, 0.0, ptr(packed), ptr(bb["scratch"]),
                                                     bb["cap"], ptr(ri), ptr(t0), ptr(t1), n_rays, s),
                      "nsr_ray_march_bricks_write")
                check(lib.nsr_nerf_prune_layout(_byref(d), m_cap, _byref(self._PL)), "nsr_nerf_prune_layout")
                pws = torch.empty(max(int(self._PL.total_bytes), 256), dtype=torch.uint8, device=dev)
                prune = lib.nsr_nerf_prune_pass_deferred if self.defer_pack else lib.nsr_nerf_prune_pass
                check(prune(_byref(d), ptr(rays_o), ptr(rays_d), ptr(ri), ptr(t0), ptr(t1), ptr(packed),
                            ptr(table), ptr(w1), ptr(pws), ptr(kept), ptr(packed2), ptr(total_s), m_cap, n_rays,
                            ptr(total_m), s_cap, ptr(stats[8:16]), None, s), "nsr_nerf_prune_pass")
                check(lib.nsr_nerf_main_layout(_byref(d), s_cap, n_rays, _byref(self._ML)), "nsr_nerf_main_layout")
                L = copy.copy(self._ML)
                ws = torch.empty(int(L.total_bytes), dtype=torch.uint8, device=dev)
                check(lib.nsr_nerf_render_forward(_byref(d), ptr(pws), m_cap, ptr(packed), ptr(packed2), ptr(t0), ptr(t1),
                                                  ptr(rays_d), ptr(bg), ptr(w1), ptr(w2), ptr(ws), s_cap, n_rays,
                                                  int(bool(prepare_backward)), ptr(total_s), None, s), "nsr_nerf_render_forward")
                if lazy:  # no synchronisation: the counts travel to pinned memory behind the pass, the next call reads them
                    host = bb["hosts"][bb["flip"]]
                    bb["flip"] ^= 1
                    host.copy_(stats, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                    bb["pending"] = (host, ev, (m_cap, s_cap))
                    M = S = None
                    break
                # the ONE synchronisation of the forward: both counts (unclamped) in one pinned read-back
                host = bb["host"]
                host.copy_(stats, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                M, S = int(host[0]), int(host[8])
                grow = lambda n: -(-int(1.3 * n) // 65536) * 65536  # noqa: E731
                if M <= m_cap and S <= s_cap:
                    # capacities follow the counts: 1.3 x the last ones, re-sized when they leave the [40 %, 90 %] band
                    if M > 0.9 * m_cap or M < 0.4 * m_cap:
                        bb["m_cap"] = max(grow(M), 1 << 17)
                    if S > 0.9 * s_cap or S < 0.4 * s_cap:
                        bb["s_cap"] = max(grow(S), 1 << 16)
                    break
                if M > m_cap:   # (the kept count of a truncated sigma pass means nothing: it is re-evaluated next round)
                    bb["m_cap"] = grow(M)
                elif S > s_cap:
                    bb["s_cap"] = grow(S)

        if not lazy:  # a synchronising call: its counts size the next lazy call (and are what that one reports as "previous")
            hostc = bb["hosts"][bb["flip"]]
            bb["flip"] ^= 1
            hostc.copy_(bb["host"])
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            bb["pending"], bb["prev_counts"] = (hostc, ev, (bb["m_cap"], bb["s_cap"])), (M, S)

        def view(off, n, dtype, shape):
            return ws[off:off + n * dtype.itemsize].view(dtype).view(shape)

        Sv = S if S is not None else s_cap  # (lazy: capacity-sized views, the live count is not on the host yet)
        out = {"comp_rgb": view(L.comp_rgb, n_rays * 3, F32, (n_rays, 3)), "opacity": view(L.opacity, n_rays, F32, (n_rays, 1)),
               "depth": view(L.depth, n_rays, F32, (n_rays, 1)), "weights": view(L.weights, Sv, F32, (Sv,)),
               "ray_indices": view(L.ray_indices, Sv, torch.int64, (Sv,)), "t_starts": view(L.t_starts, Sv, F32, (Sv,)),
               "t_ends": view(L.t_ends, Sv, F32, (Sv,)), "num_samples": S, "num_marched": M}
        if lazy:
            out["count"] = _PendingCount(bb)
        state = dict(pws=pws, ws=ws, packed=packed, packed2=packed2, rays_d=rays_d, bg=bg, M=m_cap, S=s_cap, S_live=S,
                     n_kept_dev=total_s, n_rays=n_rays, w1=w1, w2=w2, keep=(meta, half, smp))
        return out, state

    def render_backward(self, state, g_comp_rgb, g_opacity=None, g_depth=None, g_weights=None):
        """gradients of the flat parameters (geometry.encoding_with_network.params [MLP | table], texture.network.params) from
        the upstream gradients of comp_rgb [R,3], opacity [R,1], depth [R,1], weights [S]; fresh fp32 tensors"""
        import nsr_hip
        ewn, tex, d = self.ewn, self.tex, self.desc
        dev = state["ws"].device
        g1 = torch.empty_like(ewn.params)
        g1[:ewn.n_network_params].zero_()  # the MLP slices are accumulated into, the table slice is overwritten
        g2 = torch.zeros_like(tex.params)
        if state["S_live"] is not None and state["S_live"] == 0:  # (None: a lazy forward -- the kernels read the device-side count)
            g1.zero_()
            return g1, g2
        up = nsr_hip.NsrRenderGrads()
        keep = []
        for name, g in (("comp_rgb", g_comp_rgb), ("opacity", g_opacity), ("depth", g_depth), ("weights", g_weights)):
            if g is not None:
                g = g.to(F32).contiguous()
                keep.append(g)
                setattr(up, name, g.data_ptr())
        with torch.no_grad(), torch.cuda.device(dev):
            check(lib.nsr_nerf_render_backward(_byref(d), ptr(state["pws"]), state["M"], ptr(state["packed"]),
                                               ptr(state["packed2"]), ptr(state["rays_d"]), ptr(state["bg"]), _byref(up),
                                               ptr(state["w1"]), ptr(state["w2"]), ptr(ewn.mlp_slice(g1)), ptr(ewn.grid_slice(g1)),
                                               ptr(g2), ptr(state["ws"]), state["S"], state["n_rays"], ptr(state["n_kept_dev"]),
                                               stream_ptr()), "nsr_nerf_render_backward")
        return g1, g2

    def _forward_backward_python(self, rays, gt_rgb, background, compute_grads=True, loss_scale=1.0, march_handle=None,
                                 after_prune=None, after_enqueue=None, before_sync=None):
        """-> dict(loss_acc, comp_rgb, opacity, depth, num_samples, weights, ray_indices, t_starts, t_ends).  Gradients
        of ``loss_scale * loss`` are ACCUMULATED into ``.grad`` of the MLP slices and OVERWRITE the hash-table slice."""
        m, ewn, tex = self.model, self.ewn, self.tex
        dev = rays.device
        n_rays = rays.shape[0]
        rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()
        with torch.no_grad(), torch.cuda.device(dev):
            with _ops.timed("fused:march_prune"):
                mp = self.march_and_prune(rays_o, rays_d, keep_rows=True, handle=march_handle, after_prune=after_prune,
                                          before_sync=before_sync)
            packed, ri, t0, t1, M = mp["packed"], mp["ri"], mp["t0"], mp["t1"], mp["M"]
            S = ri.shape[0]
            s = stream_ptr()
            w1, w2 = ewn.weights_half(ewn.params), tex.weights_half(tex.params)
            with _ops.timed("fused:forward"):
                if S > 0:
                    x01, dirs, enc, out1, acts1 = mp["x01"], mp["dirs"], mp["enc"], mp["out1"], mp["acts1"]
                else:
                    x01 = dirs = torch.empty((0, 3), dtype=F32, device=dev)
                    enc, out1 = torch.empty((0, 32), dtype=F16, device=dev), torch.empty((0, 16), dtype=F16, device=dev)
                    acts1 = torch.empty((1, 0, 64), dtype=F16, device=dev)
                tex_in = torch.empty((S, 32), dtype=F16, device=dev)
                check(lib.nsr_texture_input(ptr(out1), 16, ptr(dirs), ptr(tex_in), S, None, s), "nsr_texture_input")
                out2, acts2 = _ops.mlp_forward(tex_in, w2, tex.mlp_desc, save_acts=compute_grads)
                weights, trans = torch.empty(S, dtype=F32, device=dev), torch.empty(S, dtype=F32, device=dev)
                comp_rgb = torch.empty((n_rays, 3), dtype=F32, device=dev)
                opacity, depth = torch.empty((n_rays, 1), dtype=F32, device=dev), torch.empty((n_rays, 1), dtype=F32, device=dev)
                bg = background.to(F32).contiguous()
                check(lib.nsr_composite_forward(ptr(out1), 16, self.bias, ptr(t0), ptr(t1), ptr(out2), out2.stride(0),
                                                ptr(packed), ptr(bg), ptr(weights), ptr(trans), ptr(comp_rgb),
                                                ptr(opacity), ptr(depth), n_rays, s), "nsr_composite_forward")
                acc = torch.zeros(2, dtype=F32, device=dev)
                gt = gt_rgb.to(F32).contiguous()
                check(lib.nsr_smooth_l1_valid(ptr(comp_rgb), ptr(opacity), ptr(gt), ptr(acc), n_rays, s), "nsr_smooth_l1_valid")
            res = {"comp_rgb": comp_rgb, "opacity": opacity, "depth": depth, "rays_valid": opacity > 0,
                   "num_samples": S, "num_marched": M, "weights": weights, "ray_indices": ri, "t_starts": t0,
                   "t_ends": t1, "loss_acc": acc}
            if not compute_grads or S == 0:
                if after_enqueue is not None:
                    after_enqueue()
                return res
            with _ops.timed("fused:backward"):
                g_comp = torch.empty((n_rays, 3), dtype=F32, device=dev)
                check(lib.nsr_smooth_l1_valid_backward(ptr(comp_rgb), ptr(opacity), ptr(gt), ptr(acc), float(loss_scale),
                                                       ptr(g_comp), n_rays, s), "nsr_smooth_l1_valid_backward")
                d_rgb, d_logit = torch.empty((S, 3), dtype=F32, device=dev), torch.empty(S, dtype=F32, device=dev)
                check(lib.nsr_composite_backward(ptr(out1), 16, self.bias, ptr(t0), ptr(t1), ptr(out2), out2.stride(0),
                                                 ptr(packed), ptr(bg), ptr(weights), ptr(trans), ptr(g_comp), None, None,
                                                 ptr(d_rgb), ptr(d_logit), n_rays, s), "nsr_composite_backward")
                for p in (ewn.params, tex.params):
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                # colour MLP: dW2 and d(tex_in) (first 16 columns = d feature)
                d_tex = self._mlp_backward(d_rgb, 3, None, out2, tex_in, acts2, w2, tex.mlp_desc, tex.params.grad, 0)
                # density MLP: dout = d feature (+ d logit on column 0) -> dW1 and d(enc), level-major
                d_enc = self._mlp_backward(d_tex, 32, d_logit, out1, enc, acts1, w1, ewn.mlp_desc,
                                           ewn.mlp_slice(ewn.params.grad), ewn.grid_desc.n_features)
                _ops.hashgrid_backward_params(x01, d_enc, ewn.grid_slice(ewn.params.grad), ewn.grid_desc,
                                              accumulate=False, level_major=True)
            if after_enqueue is not None:
                after_enqueue()  # the whole step is queued: host time spent here overlaps GPU work
            return res

    # ---- fully asynchronous step: every sample count stays on the device ------------------------------------
    def async_ray_sets(self, k_sets, slots, dev):
        """``k_sets`` ray sets (the buffers one marching pass writes) whose marcher-facing arrays are CONTIGUOUS across
        sets -- ``ro/rd [K][slots][3]``, ``t_min/t_max/counts [K][slots]``, scratch rows ``[K][slots][cap]`` -- so that any
        run of consecutive sets is marched by ONE launch over ``n_sets * slots`` rays (``march_async_many``): the grid only
        changes every 16 steps, nothing forces one marching launch per step"""
        grid = self.model.occupancy_grid
        cap = int(lib.nsr_ray_march_capacity((ctypes.c_float * 6)(*[float(v) for v in grid._roi_host]),
                                             float(self.model.render_step_size)))
        K, n = int(k_sets), int(slots)
        big = dict(ro=torch.empty((K, n, 3), dtype=F32, device=dev), rd=torch.empty((K, n, 3), dtype=F32, device=dev),
                   t_min=torch.empty((K, n), dtype=F32, device=dev), t_max=torch.empty((K, n), dtype=F32, device=dev),
                   counts=torch.empty((K, n), dtype=torch.int32, device=dev),
                   scratch=torch.empty((K, n * cap * 2), dtype=F32, device=dev))
        sets = []
        for k in range(K):
            rs = dict(slots=n, cap=cap, index=k, big=big)
            rs["buf"] = torch.empty(n * 10, dtype=F32, device=dev)  # rays(6) rgb(3) fg(1)
            b = rs["buf"]
            rs["rays"], rs["rgb"], rs["fg"] = b[:6 * n].view(n, 6), b[6 * n:9 * n].view(n, 3), b[9 * n:10 * n]
            rs["ro"], rs["rd"], rs["t_min"], rs["t_max"] = big["ro"][k], big["rd"][k], big["t_min"][k], big["t_max"][k]
            rs["counts"], rs["scratch"] = big["counts"][k], big["scratch"][k]
            rs["u"] = torch.empty((5, max(n, 3)), dtype=F32, device=dev)
            rs["packed"] = torch.empty((n, 2), dtype=torch.int32, device=dev)
            rs["total"] = torch.zeros(1, dtype=torch.int32, device=dev)
            sets.append(rs)
        return sets

    def async_ray_set(self, slots, dataset_like_device):
        return self.async_ray_sets(1, slots, dataset_like_device)[0]

    def prepare_rays_async(self, rs, dataset, generator, n_active, background="random"):
        """pixel choice -> rays -> slab test -> jitter of ONE ray set, one launch on the current stream"""
        m = self.model
        slots = rs["slots"]
        rs["u"].uniform_(generator=generator)
        rs["bg"] = rs["u"][4, :3] if background == "random" else torch.ones(3, device=rs["u"].device)
        jitter = float(m.render_step_size) if m.randomized else 0.0
        check(lib.nsr_prepare_train_rays(ptr(dataset.all_images), ptr(dataset.all_fg_masks), ptr(dataset.directions),
                                         ptr(dataset.all_c2w), ptr(rs["u"]), ptr(rs["bg"]),
                                         dataset.all_images.shape[0], dataset.h, dataset.w, int(dataset.apply_mask),
                                         ptr(m.scene_aabb), jitter, ptr(rs["rays"]), ptr(rs["ro"]), ptr(rs["rd"]),
                                         ptr(rs["rgb"]), ptr(rs["fg"]), ptr(rs["t_min"]), ptr(rs["t_max"]), slots,
                                         ptr(n_active), stream_ptr()), "nsr_prepare_train_rays")

    def march_async_many(self, sets, dataset, generator, background="random", bricks=None):
        """ray preparation of every set in ``sets`` (consecutive sets of one ``async_ray_sets`` allocation) and ONE marching
        launch over all their rays, on the current stream; no host sync, no packing (``pack_async`` per set, later)"""
        m, grid = self.model, self.model.occupancy_grid
        k0, n_sets, slots = sets[0]["index"], len(sets), sets[0]["slots"]
        assert [rs["index"] for rs in sets] == list(range(k0, k0 + n_sets)), "sets must be consecutive"
        for rs in sets:
            self.prepare_rays_async(rs, dataset, generator, None, background)
        if bricks is None:
            bricks = _ops.grid_bricks(grid.binary)
        if bricks is None:
            raise NotImplementedError("the asynchronous step needs a brick-able occupancy grid (resolution % 16 == 0)")
        rx, ry, rz = (int(v) for v in grid.binary.shape)
        big = sets[0]["big"]
        sl = slice(k0, k0 + n_sets)
        with _ops.timed("ray_march_count", n_sets * slots):
            check(lib.nsr_ray_march_bricks_count(ptr(big["ro"][sl]), ptr(big["rd"][sl]), ptr(big["t_min"][sl]),
                                                 ptr(big["t_max"][sl]), ptr(grid.roi_aabb), ptr(bricks), rx, ry, rz,
                                                 ContractionType.AABB.value, float(m.render_step_size), 0.0,
                                                 ptr(big["counts"][sl]), ptr(big["scratch"][sl]), sets[0]["cap"],
                                                 n_sets * slots, stream_ptr()), "nsr_ray_march_bricks_count")

    def march_async(self, rs, dataset, generator, n_active, m_cap, stats, background="random", bricks=None,
                    pack_masks=False):
        """ray preparation + marching pass (+ capped packing unless ``m_cap`` is None) into ray set ``rs`` on the CURRENT
        stream; no host sync.  ``n_active`` (device int32[1] or None): dead-slot marking at ray preparation;
        ``bricks``: the packed occupancy grid to march through (default: pack / look up the model's current grid)"""
        m, grid = self.model, self.model.occupancy_grid
        slots = rs["slots"]
        self.prepare_rays_async(rs, dataset, generator, n_active, background)
        if bricks is None:
            bricks = _ops.grid_bricks(grid.binary)
        if bricks is None:
            raise NotImplementedError("the asynchronous step needs a brick-able occupancy grid (resolution % 16 == 0)")
        rx, ry, rz = (int(v) for v in grid.binary.shape)
        s = stream_ptr()
        with _ops.timed("ray_march_count", slots):
            check(lib.nsr_ray_march_bricks_count(ptr(rs["ro"]), ptr(rs["rd"]), ptr(rs["t_min"]), ptr(rs["t_max"]),
                                                 ptr(grid.roi_aabb), ptr(bricks), rx, ry, rz,
                                                 ContractionType.AABB.value, float(m.render_step_size), 0.0,
                                                 ptr(rs["counts"]), ptr(rs["scratch"]), rs["cap"], slots, s),
                  "nsr_ray_march_bricks_count")
        if m_cap is not None:
            self.pack_async(rs, n_active if pack_masks else None, m_cap, stats)

    def pack_async(self, rs, n_active, m_cap, stats, stream=None):
        """packed_info / total of ray set ``rs`` from its marched counts, clamped to ``m_cap``; slots >= n_active[0]
        (device) keep nothing.  Separate from the marching pass so that the pass can run before the ray count exists.
        ``stream``: raw stream pointer (default: the current stream)"""
        check(lib.nsr_pack_from_counts_capped(ptr(rs["counts"]), ptr(rs["packed"]), ptr(rs["total"]), rs["slots"],
                                              int(m_cap), ptr(stats), ptr(n_active),
                                              stream if stream is not None else stream_ptr()),
              "nsr_pack_from_counts_capped")
        rs["m_cap"] = int(m_cap)
        if rs.get("marched") is not None:
            rs["marched"]["valid"] = False  # sample arrays of an earlier packing of this ring slot

    def write_async(self, rs, consumer_stream=None, stream=None, writer_stream=None):
        """sample arrays of ray set ``rs`` (ray index, t_starts, t_ends, unit-cube positions of every marched sample) from
        its marching scratch + packed_info, into buffers that belong to the ring slot -- queued on the CURRENT stream right
        behind ``pack_async`` (the marching side stream), so the step itself starts at the hash encode.
        ``consumer_stream``: the stream the step runs on (allocator bookkeeping for buffers born here)."""
        m_cap, slots = rs["m_cap"], rs["slots"]
        dev = rs["buf"].device
        mb = rs.get("marched")
        if mb is None or mb["m_cap"] != m_cap:
            mb = rs["marched"] = dict(m_cap=m_cap, ri=torch.empty(m_cap, dtype=torch.int64, device=dev),
                                      t0=torch.empty((m_cap, 1), dtype=F32, device=dev),
                                      t1=torch.empty((m_cap, 1), dtype=F32, device=dev),
                                      x01=torch.empty((m_cap, 3), dtype=F32, device=dev), valid=False)
            if consumer_stream is not None:
                for k in ("ri", "t0", "t1", "x01"):
                    mb[k].record_stream(consumer_stream)
            if writer_stream is not None:
                #  the buffers were just carved from the CURRENT stream's allocator pool but are written through a raw
                # pointer of ``writer_stream``: tell the allocator, and order the writer behind whatever the current stream still
                # has queued on a recycled block
                cur = torch.cuda.current_stream(dev)
                for k in ("ri", "t0", "t1", "x01"):
                    mb[k].record_stream(writer_stream)
                if writer_stream != cur:
                    ev = torch.cuda.Event()
                    ev.record(cur)
                    writer_stream.wait_event(ev)
        grid, d = self.model.occupancy_grid, self.desc
        rx, ry, rz = (int(v) for v in grid.binary.shape)
        with device_guard(dev):
            s = stream if stream is not None else stream_ptr()
            check(lib.nsr_ray_march_bricks_write(ptr(rs["ro"]), ptr(rs["rd"]), ptr(rs["t_min"]), ptr(rs["t_max"]),
                                                 ptr(grid.roi_aabb), None, rx, ry, rz, ContractionType.AABB.value,
                                                 float(self.model.render_step_size), 0.0, ptr(rs["packed"]),
                                                 ptr(rs["scratch"]), rs["cap"], ptr(mb["ri"]), ptr(mb["t0"]),
                                                 ptr(mb["t1"]), slots, s), "nsr_ray_march_bricks_write")
            check(lib.nsr_sample_positions_unit(ptr(rs["ro"]), ptr(rs["rd"]), ptr(mb["ri"]), ptr(mb["t0"]), ptr(mb["t1"]),
                                                float(d.radius), int(d.contraction), ptr(mb["x01"]), None, m_cap,
                                                ptr(rs["total"]), s), "nsr_sample_positions_unit")
        mb["valid"] = True

    def _async_buffers(self, slots, m_cap, s_cap, dev):
        key = (slots, m_cap, s_cap)
        ab = getattr(self, "_ab", None)
        if ab is not None and ab["key"] == key:
            return ab
        d = self.desc
        check(lib.nsr_nerf_prune_layout(_byref(d), m_cap, _byref(self._PL)), "nsr_nerf_prune_layout")
        check(lib.nsr_nerf_main_layout(_byref(d), s_cap, slots, _byref(self._ML)), "nsr_nerf_main_layout")
        ab = dict(key=key, prev=ab)  # the previous set stays referenced until the next resize: queued work may use it
        if ab["prev"] is not None:
            ab["prev"]["prev"] = None
        ab["pws"] = torch.empty(max(int(self._PL.total_bytes), 256), dtype=torch.uint8, device=dev)
        ab["ws"] = torch.empty(int(self._ML.total_bytes), dtype=torch.uint8, device=dev)
        ab["ri"] = torch.empty(m_cap, dtype=torch.int64, device=dev)
        ab["t0"] = torch.empty((m_cap, 1), dtype=F32, device=dev)
        ab["t1"] = torch.empty((m_cap, 1), dtype=F32, device=dev)
        ab["meta"] = torch.zeros(3 * slots + 1, dtype=torch.int32, device=dev)  # kept | packed_kept | total
        import copy
        ab["ML"] = copy.copy(self._ML)
        self._ab = ab
        return ab

    def forward_backward_async(self, rs, s_cap, kept_stats, loss_scale=1.0, compute_grads=True, after_prune_queued=None,
                               table_adam=None, exchange=None, defer_wgrad_join=False):
        """the training step on ray set ``rs`` (filled by march_async, possibly on another stream -- the caller orders
        the streams) with NO host synchronisation: the marched / kept sample counts stay on the device, all buffers
        have fixed capacities (rs['m_cap'], s_cap) and every kernel is launched for the capacity.
        ``after_prune_queued(total_kept, pruned_event)`` runs once the whole step is queued; ``pruned_event`` was recorded right
        behind the pruning pass (``total_kept``: device int32[1] tensor).
        ``exchange`` = (NsrTableExchange, grad_density_mlp, grad_color_mlp): the ray-sharded form of the main pass -- the table
        gradient leaves as bf16 in level groups with an event behind each, the MLP gradients go into the given fp32 views
        (nsr/parallel.py ShardedAdamW); ``.grad`` of the parameters is not touched."""
        ewn, tex, d = self.ewn, self.tex, self.desc
        dev = rs["buf"].device
        slots, m_cap = rs["slots"], rs["m_cap"]
        d.loss_scale = float(loss_scale)
        ab = self._async_buffers(slots, m_cap, int(s_cap), dev)
        fast = self._forward_backward_async_cached(rs, ab, int(s_cap), kept_stats, compute_grads, after_prune_queued, table_adam,
                                                   exchange, defer_wgrad_join)
        if fast is not None:
            return fast
        meta = ab["meta"]
        kept, packed2, total = meta[:slots], meta[slots:3 * slots].view(slots, 2), meta[3 * slots:]
        with torch.no_grad(), device_guard(dev):
            s = stream_ptr()
            grid = self.model.occupancy_grid
            rx, ry, rz = (int(v) for v in grid.binary.shape)
            half = ewn.half_params(ewn.params)
            table, w1, w2 = half[ewn.n_network_params:], half[:ewn.n_network_params], tex.half_params(tex.params)
            mb = rs.get("marched")  # sample arrays already written behind the packing kernel (write_async)?
            if mb is None or mb["m_cap"] != m_cap or not mb["valid"]:
                mb, x01m = dict(ri=ab["ri"], t0=ab["t0"], t1=ab["t1"]), None
                with _ops.timed("fused:march_prune"):
                    check(lib.nsr_ray_march_bricks_write(ptr(rs["ro"]), ptr(rs["rd"]), ptr(rs["t_min"]), ptr(rs["t_max"]),
                                                         ptr(grid.roi_aabb), None, rx, ry, rz, ContractionType.AABB.value,
                                                         float(self.model.render_step_size), 0.0, ptr(rs["packed"]),
                                                         ptr(rs["scratch"]), rs["cap"], ptr(mb["ri"]), ptr(mb["t0"]),
                                                         ptr(mb["t1"]), slots, s), "nsr_ray_march_bricks_write")
            else:
                x01m = mb["x01"]
                mb["valid"] = False  # consumed: the ring slot is re-marched before its next use
            # (the kept counts are packed by the main pass's first kernel unless something is queued behind the pruning pass
            # that reads the count before it: the torch-event variants below)
            defer = self.defer_pack and (after_prune_queued is None or (compute_grads and self.kept_rows_event))
            prune = lib.nsr_nerf_prune_pass_deferred if defer else lib.nsr_nerf_prune_pass
            with _ops.timed("fused:march_prune"):
                check(prune(_byref(d), ptr(rs["ro"]), ptr(rs["rd"]), ptr(mb["ri"]), ptr(mb["t0"]),
                            ptr(mb["t1"]), ptr(rs["packed"]), ptr(table), ptr(w1), ptr(ab["pws"]),
                            ptr(kept), ptr(packed2), ptr(total), m_cap, slots, ptr(rs["total"]),
                            int(s_cap), ptr(kept_stats), ptr(x01m), s), "nsr_nerf_prune_pass")
            # what the caller queues behind the pruning pass (the next step's ray count / packing, on a side stream) waits for
            # THIS event; the call itself comes after the main pass is queued -- the main stream must not sit idle behind the
            # pruning pass while the host issues side-stream launches (rocprofv3 timeline, round 3: pack ... 72 us ... copy_kept_rows)
            pruned = None
            if after_prune_queued is not None and not (compute_grads and self.kept_rows_event):
                pruned = self._pruned_events[self._pruned_next] if hasattr(self, "_pruned_events") else None
                if pruned is None:
                    self._pruned_events = [torch.cuda.Event() for _ in range(4)]
                    self._pruned_next = 0
                    pruned = self._pruned_events[0]
                self._pruned_next = (self._pruned_next + 1) % 4
                pruned.record(torch.cuda.current_stream())
            # (else: the main pass records an event of its own one kernel later, behind the kept-row copy -- the caller's side
            # stream waits for that one, lib.nsr_nerf_wait_kept_rows: one event record less on the step's stream)
            if exchange is not None:
                xd, g_density, g_color = exchange
                with _ops.timed("fused:main_pass"):
                    check(lib.nsr_nerf_main_pass_exchange(_byref(d), ptr(ab["pws"]), m_cap, ptr(rs["packed"]), ptr(packed2),
                                                          ptr(mb["t0"]), ptr(mb["t1"]), ptr(rs["rd"]), ptr(rs["bg"]),
                                                          ptr(rs["rgb"]), ptr(w1), ptr(w2), ptr(g_density), ptr(g_color),
                                                          ptr(ab["ws"]), int(s_cap), slots, ptr(total), ptr(x01m),
                                                          _byref(xd), s), "nsr_nerf_main_pass_exchange")
            else:
              with _ops.timed("fused:main_pass"):
                if compute_grads:
                    for p in (ewn.params, tex.params):
                        if p.grad is None:
                            p.grad = torch.zeros_like(p)
                g1 = ewn.params.grad if compute_grads else None
                g2 = tex.params.grad if compute_grads else None
                if defer_wgrad_join:  # the caller joins the helper stream itself (behind more work it queues there)
                    lib.nsr_nerf_defer_wgrad_join(1)
                try:
                    check(lib.nsr_nerf_main_pass(_byref(d), ptr(ab["pws"]), m_cap, ptr(rs["packed"]), ptr(packed2),
                                                 ptr(mb["t0"]), ptr(mb["t1"]), ptr(rs["rd"]), ptr(rs["bg"]), ptr(rs["rgb"]),
                                                 ptr(w1), ptr(w2),
                                                 ptr(ewn.mlp_slice(g1)) if compute_grads else None,
                                                 ptr(ewn.grid_slice(g1)) if compute_grads else None,
                                                 ptr(g2) if compute_grads else None, ptr(ab["ws"]), int(s_cap), slots,
                                                 int(bool(compute_grads)), ptr(total), ptr(x01m),
                                                 _byref(table_adam) if (table_adam is not None and compute_grads) else None,
                                                 s), "nsr_nerf_main_pass")
                finally:
                    if defer_wgrad_join:
                        lib.nsr_nerf_defer_wgrad_join(0)
            if after_prune_queued is not None:
                after_prune_queued(total, pruned)
            L, ws = ab["ML"], ab["ws"]

            def view(off, n, dtype, shape):
                return ws[off:off + n * dtype.itemsize].view(dtype).view(shape)

            return {"comp_rgb": view(L.comp_rgb, slots * 3, F32, (slots, 3)),
                    "opacity": view(L.opacity, slots, F32, (slots, 1)), "depth": view(L.depth, slots, F32, (slots, 1)),
                    "loss_acc": view(L.loss_acc, 2, F32, (2,)), "num_samples": total, "num_marched": rs["total"],
                    "packed_kept": packed2, "weights": view(L.weights, int(s_cap), F32, (int(s_cap),)),
                    "ray_indices": view(L.ray_indices, int(s_cap), torch.int64, (int(s_cap),)), "_workspace": ws}

    def _forward_backward_async_cached(self, rs, ab, s_cap, kept_stats, compute_grads, after_prune_queued, table_adam, exchange,
                                       defer_wgrad_join):
        """the COMMON case of ``forward_backward_async`` -- one GPU, gradients, the sample arrays already written on the marching
        stream, the kept-rows event -- with every device pointer of the two C calls resolved ONCE per (buffer set, ring slot):
        the general path converts ~90 tensors to pointers per step (slices, contiguity checks, ctypes objects: ~40 us of a
        ~300 us host step, measured with cProfile), and the host's time to queue a step is within 15 % of the GPU's time to
        run it.  Returns None when the step is not the common case (the caller goes on with the general path)."""
        mb = rs.get("marched")
        if (exchange is not None or not compute_grads or after_prune_queued is None or not self.kept_rows_event
                or not self.defer_pack or mb is None or mb["m_cap"] != rs["m_cap"] or not mb["valid"] or _ops.profiling()
                or torch._C._cuda_getDevice() != rs["buf"].device.index):
            return None
        ewn, tex, d = self.ewn, self.tex, self.desc
        half, thalf = ewn.half_params(ewn.params), tex.half_params(tex.params)
        g1, g2 = ewn.params.grad, tex.params.grad
        if g1 is None or g2 is None:
            return None
        key = (id(ab), id(mb), half.data_ptr(), thalf.data_ptr(), g1.data_ptr(), g2.data_ptr(), kept_stats.data_ptr(), s_cap)
        c = rs.get("_fbc")
        if c is None or c["key"] != key:
            slots = rs["slots"]
            meta = ab["meta"]
            kept, packed2, total = meta[:slots], meta[slots:3 * slots].view(slots, 2), meta[3 * slots:]
            n0 = ewn.n_network_params
            P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731  (every tensor below is a contiguous GPU tensor by construction)
            L, ws = ab["ML"], ab["ws"]

            def view(off, n, dtype, shape):
                return ws[off:off + n * dtype.itemsize].view(dtype).view(shape)

            c = rs["_fbc"] = dict(
                key=key, total=total,
                prune=(_byref(d), P(rs["ro"]), P(rs["rd"]), P(mb["ri"]), P(mb["t0"]), P(mb["t1"]), P(rs["packed"]),
                       ctypes.c_void_p(half.data_ptr() + 2 * n0), P(half), P(ab["pws"]), P(kept), P(packed2), P(total),
                       rs["m_cap"], slots, P(rs["total"]), s_cap, P(kept_stats), P(mb["x01"])),
                main_a=(_byref(d), P(ab["pws"]), rs["m_cap"], P(rs["packed"]), P(packed2), P(mb["t0"]), P(mb["t1"]), P(rs["rd"])),
                # (rs["bg"] is re-made by every ray preparation: resolved per step)
                main_b=(P(rs["rgb"]), P(half), P(thalf), P(ewn.mlp_slice(g1)), P(ewn.grid_slice(g1)), P(g2), P(ws), s_cap, slots, 1,
                        P(total), P(mb["x01"])),
                keep=(half, thalf, g1, g2, mb, ab),
                result={"comp_rgb": view(L.comp_rgb, slots * 3, F32, (slots, 3)),
                        "opacity": view(L.opacity, slots, F32, (slots, 1)), "depth": view(L.depth, slots, F32, (slots, 1)),
                        "loss_acc": view(L.loss_acc, 2, F32, (2,)), "num_samples": total, "num_marched": rs["total"],
                        "packed_kept": packed2, "weights": view(L.weights, s_cap, F32, (s_cap,)),
                        "ray_indices": view(L.ray_indices, s_cap, torch.int64, (s_cap,)), "_workspace": ws})
        mb["valid"] = False  # consumed: the ring slot is re-marched before its next use
        s = stream_ptr()
        check(lib.nsr_nerf_prune_pass_deferred(*c["prune"], s), "nsr_nerf_prune_pass")
        if defer_wgrad_join:
            lib.nsr_nerf_defer_wgrad_join(1)
        try:
            check(lib.nsr_nerf_main_pass(*c["main_a"], ctypes.c_void_p(rs["bg"].data_ptr()), *c["main_b"],
                                         _byref(table_adam) if table_adam is not None else None, s), "nsr_nerf_main_pass")
        finally:
            if defer_wgrad_join:
                lib.nsr_nerf_defer_wgrad_join(0)
        after_prune_queued(c["total"], None)
        return dict(c["result"])

    # ---- occupancy refresh without a host sync --------------------------------------------------------------------
    def refresh_occupancy_async(self, step, bricks, occ_thre=0.01, ema_decay=0.95, warmup_steps=256):
        """``OccupancyGrid._update`` (nerfacc 0.3.3, reference models/nerf.py:45-55) queued on the current stream with the
        selected-cell count kept on the device (csrc/occupancy.hip): no ``torch.nonzero``, ~12 launches.  Updates
        ``grid.occs`` and ``grid.binary`` IN PLACE and re-packs ``bricks`` (the persistent 4^3-brick bitfield)."""
        m, grid, ewn, d = self.model, self.model.occupancy_grid, self.ewn, self.desc
        dev = grid.occs.device
        rx, ry, rz = grid._res
        N = grid.num_cells
        all_cells = step < warmup_steps
        n_uniform = N // 4
        cap = N if all_cells else 2 * n_uniform
        ob = getattr(self, "_occ_buf", None)
        if ob is None:
            C = d.grid.n_levels * d.grid.n_features
            ob = self._occ_buf = dict(
                cells=torch.empty(N, dtype=torch.int32, device=dev), x_unit=torch.empty(N * 3, dtype=F32, device=dev),
                world=torch.empty(N * 3, dtype=F32, device=dev), x01=torch.empty(N * 3, dtype=F32, device=dev),
                enc=torch.empty(N * C, dtype=F16, device=dev), out=torch.empty(N * 16, dtype=F16, device=dev),
                u=torch.empty(2 * n_uniform, dtype=F32, device=dev), jitter=torch.empty(N * 3, dtype=F32, device=dev),
                brick_offset=torch.empty(max(bricks.numel(), 1), dtype=torch.int32, device=dev),
                occupied=torch.empty(N, dtype=torch.int32, device=dev),
                counts=torch.zeros(4, dtype=torch.int32, device=dev), occs=torch.empty(N, dtype=F32, device=dev),
                thr=torch.empty(2 + 2 * 256, dtype=F32, device=dev))
        binary = grid._binary
        assert binary.is_contiguous() and binary.dtype == torch.bool
        n_occ, n_cells = ob["counts"][0:1], ob["counts"][1:2]
        half = ewn.half_params(ewn.params)
        table, w1 = half[ewn.n_network_params:], half[:ewn.n_network_params]
        with torch.no_grad(), torch.cuda.device(dev):
            s = stream_ptr()
            ob["jitter"][:cap * 3].uniform_()
            if not all_cells:
                ob["u"].uniform_()
            check(lib.nsr_occupancy_select_cells(ptr(bricks), rx, ry, rz, ptr(ob["u"][:n_uniform]),
                                                 ptr(ob["u"][n_uniform:]), ptr(ob["jitter"]), n_uniform, int(all_cells),
                                                 cap, ptr(ob["brick_offset"]), ptr(ob["occupied"]), ptr(n_occ),
                                                 ptr(ob["cells"]), ptr(ob["x_unit"]), ptr(n_cells), s),
                  "nsr_occupancy_select_cells")
            # positions exactly as the torch path forms them: grid-unit -> world (contract_inv) -> the geometry's own
            # contract_to_unisphere (models/geometry.py:122-124)
            check(lib.nsr_contract_inv(ptr(ob["x_unit"]), ptr(grid.roi_aabb), ContractionType.AABB.value, ptr(ob["world"]),
                                       cap, s), "nsr_contract_inv")
            check(lib.nsr_contract_to_unisphere(ptr(ob["world"]), self.radius, ContractionType.AABB.value, ptr(ob["x01"]),
                                                cap, s), "nsr_contract_to_unisphere")
            C = d.grid.n_levels * d.grid.n_features
            check(lib.nsr_hashgrid_forward_ex(ptr(ob["x01"]), ptr(table), ptr(ob["enc"]), cap, C, 1, d.grid.n_levels,
                                              _byref(d.grid), ptr(n_cells), s), "nsr_hashgrid_forward_ex")
            check(lib.nsr_mlp_forward_ex(ptr(ob["enc"]), 0, C, d.grid.n_features, ptr(w1), ptr(ob["out"]), None, cap,
                                         _byref(d.mlp_density), ptr(n_cells), s), "nsr_mlp_forward_ex")
            check(lib.nsr_occupancy_update(ptr(ob["out"]), 16, self.bias, float(m.render_step_size), float(ema_decay),
                                           float(occ_thre), ptr(ob["cells"]), ptr(grid.occs), ptr(ob["occs"]),
                                           ptr(binary.view(torch.uint8)), ptr(ob["thr"]), N, cap, ptr(n_cells), s),
                  "nsr_occupancy_update")
            grid.occs.copy_(ob["occs"])
            check(lib.nsr_grid_pack_bricks(ptr(binary.view(torch.uint8)), rx, ry, rz, ptr(bricks), s),
                  "nsr_grid_pack_bricks")
        try:  # the cache of ops.grid_bricks keys on the tensor version, which a raw-pointer write does not bump
            binary._nsr_bricks = (binary._version, binary.data_ptr(), bricks)
        except Exception:  # noqa: BLE001
            pass

    def _mlp_backward(self, dout, dout_stride, extra, out, x, acts, w, desc, grad_w, dx_lm_f):
        n = x.shape[0]
        dx = torch.empty(n * desc.n_in, dtype=F32, device=x.device)
        nws = lib.nsr_mlp_backward_workspace_floats(_byref(desc), n)
        partials = torch.empty(int(nws), dtype=F32, device=x.device)
        with _ops.timed(f"mlp_backward_h{desc.n_hidden}", n):
            check(lib.nsr_mlp_backward_ex(ptr(dout), 1, dout_stride, ptr(extra), ptr(out), ptr(x), 0, x.stride(0), 0,
                                          ptr(acts), ptr(w), ptr(grad_w), ptr(dx), desc.n_in, dx_lm_f, ptr(partials), n,
                                          self.grad_scale, _byref(desc), None, stream_ptr()), "nsr_mlp_backward_ex")
        return dx if dx_lm_f else dx.view(n, desc.n_in)

    @staticmethod
    def loss_value(res):
        acc = res["loss_acc"]
        return acc[0] / torch.clamp(3.0 * acc[1], min=1.0)


def gather_train_rays(dataset, n_rays, generator, background="random"):
    """one RNG call + one gather kernel instead of ~15 indexing kernels (reference systems/nerf.py:38-79)"""
    dev = dataset.all_images.device
    n_img, H, W = dataset.all_images.shape[0], dataset.h, dataset.w
    r = torch.rand((4, max(n_rays, 1)), device=dev, generator=generator)
    index = (r[0] * n_img).long().clamp_(max=n_img - 1)
    px = (r[1] * W).long().clamp_(max=W - 1)
    py = (r[2] * H).long().clamp_(max=H - 1)
    bg = r[3, :3].contiguous() if background == "random" else torch.ones(3, device=dev)
    rays = torch.empty((n_rays, 6), dtype=F32, device=dev)
    rgb, fg = torch.empty((n_rays, 3), dtype=F32, device=dev), torch.empty(n_rays, dtype=F32, device=dev)
    with torch.cuda.device(dev):
        check(lib.nsr_gather_train_rays(ptr(dataset.all_images), ptr(dataset.all_fg_masks), ptr(dataset.directions),
                                        ptr(dataset.all_c2w), ptr(index), ptr(px), ptr(py), ptr(bg), H, W,
                                        int(dataset.apply_mask), ptr(rays), ptr(rgb), ptr(fg), n_rays, stream_ptr()),
              "nsr_gather_train_rays")
    return rays, rgb, fg, bg


def prepare_train_rays(dataset, n_rays, generator, model, background="random", n_active=None):
    """ONE RNG call + ONE kernel: pixel choice, gather, get_rays, background blend, slab test, stratified jitter.
    -> rays[n,6], rays_o, rays_d, rgb, fg, bg, t_min, t_max

    ``n_active`` (int32[1] on the device): only the first ``n_active[0]`` of the ``n_rays`` slots are live rays, the
    rest are dead (no samples, outside the loss) -- the batch size is then a device-side value and this call never
    needs the host to know it."""
    dev = dataset.all_images.device
    n_img, H, W = dataset.all_images.shape[0], dataset.h, dataset.w
    u = torch.rand((5, max(n_rays, 3)), device=dev, generator=generator)
    bg = u[4, :3].contiguous() if background == "random" else torch.ones(3, device=dev)
    buf = torch.empty(n_rays * 18, dtype=F32, device=dev)  # rays(6) o(3) d(3) rgb(3) fg(1) tmin(1) tmax(1)
    rays, ro, rd = buf[:6 * n_rays].view(n_rays, 6), buf[6 * n_rays:9 * n_rays].view(n_rays, 3), buf[9 * n_rays:12 * n_rays].view(n_rays, 3)
    rgb, fg = buf[12 * n_rays:15 * n_rays].view(n_rays, 3), buf[15 * n_rays:16 * n_rays]
    t_min, t_max = buf[16 * n_rays:17 * n_rays], buf[17 * n_rays:18 * n_rays]
    u4 = u if u.shape[1] == n_rays else u[:, :n_rays].contiguous()
    jitter = float(model.render_step_size) if model.randomized else 0.0
    with torch.cuda.device(dev):
        check(lib.nsr_prepare_train_rays(ptr(dataset.all_images), ptr(dataset.all_fg_masks), ptr(dataset.directions),
                                         ptr(dataset.all_c2w), ptr(u4), ptr(bg), n_img, H, W, int(dataset.apply_mask),
                                         ptr(model.scene_aabb), jitter, ptr(rays), ptr(ro), ptr(rd), ptr(rgb), ptr(fg),
                                         ptr(t_min), ptr(t_max), n_rays, ptr(n_active), stream_ptr()),
              "nsr_prepare_train_rays")
    return rays, ro, rd, rgb, fg, bg, t_min, t_max
