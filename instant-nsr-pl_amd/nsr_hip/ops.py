"""Tensor-level wrappers and autograd Functions over the C ABI (``include/nsr_hip.h``).

Everything here is host-side plumbing: allocate outputs with torch, pass raw pointers + the current HIP
stream to ``libnsr_hip.so``.  No arithmetic of the hot path happens in Python.
"""
import ctypes
import time

import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import NsrError, check, device_guard, lib, ptr, stream_ptr

_byref = ctypes.byref
F32, F16 = torch.float32, torch.float16

# ---- optional HIP-event timing of the hot kernels (bench.py's roofline leg) -----------------------------------
# Events are recorded on torch's current stream, which is the stream every kernel here is launched on.
_PROFILE = None
_NATIVE_ON = False


_NATIVE_TAGS = {0: "hashgrid_forward", 1: "hashgrid_backward_params", 2: "mlp_forward_h1", 3: "mlp_forward_h2",
                4: "mlp_backward_h2", 5: "mlp_backward_h1", 6: "hashgrid_backward_bin", 7: "hashgrid_backward_dense"}
# (tag 4 times nsr_mlp_dgrad_pair -- both networks' data gradients in one launch -- when the step uses it; tag 5 is then absent)


def profile_begin(native_only=False):
    """start collecting HIP-event timings.  ``native_only``: only the launches timed inside the C orchestration
    (pooled events, ~2 us of host time per scope); the Python-side scopes allocate a torch event pair each and are too
    expensive to leave on inside a measured region."""
    global _PROFILE, _NATIVE_ON
    _PROFILE = None if native_only else {}
    _NATIVE_ON = True
    d, a, b = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
    lib.nsr_profile_collect(-1, _byref(d), _byref(a), _byref(b))
    lib.nsr_profile_enable(1)


def profiling():
    """True while HIP-event timing is on (events cannot be recorded into a captured graph)"""
    return _PROFILE is not None or _NATIVE_ON


def profile_end():
    """-> {name: (total_ms, launches, units)}; synchronises."""
    global _PROFILE, _NATIVE_ON
    prof, _PROFILE, _NATIVE_ON = _PROFILE or {}, None, False
    torch.cuda.synchronize()
    lib.nsr_profile_enable(0)
    out = {k: (sum(a.elapsed_time(b) for a, b, _ in v), len(v), sum(u for _, _, u in v)) for k, v in prof.items()}
    for tag, name in _NATIVE_TAGS.items():  # launches issued by the native phase orchestration (csrc/step.hip)
        d, n, u = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        check(lib.nsr_profile_collect(tag, _byref(d), _byref(n), _byref(u)), "nsr_profile_collect")
        if n.value:
            key = name + ("" if name not in out else "")
            t0, n0, u0 = out.get(key, (0.0, 0, 0))
            out[key] = (t0 + d.value, n0 + n.value, u0 + u.value)
    lib.nsr_profile_collect(-1, _byref(d), _byref(n), _byref(u))
    return out


class _timed:
    def __init__(self, name, units):
        self.name, self.units = name, units

    def __enter__(self):
        if _PROFILE is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if _PROFILE is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _PROFILE.setdefault(self.name, []).append((self.a, b, self.units))


def timed(name, units=1):
    """phase timers for bench.py / the trainer (no-ops unless profile_begin() was called)"""
    return _timed(name, units)


def _f32c(t):
    return t.detach().to(F32).contiguous()


def _is_f32(t):
    if t.dtype == F32:
        return 1
    if t.dtype == F16:
        return 0
    raise NsrError(f"expected float16/float32 tensor, got {t.dtype}")


# ------------------------------------------------------------------------------------------------
# hash grid
# ------------------------------------------------------------------------------------------------
def hashgrid_forward(x, table_half, desc, mask_count=None, out=None):
    n = x.shape[0]
    C = desc.n_levels * desc.n_features
    y = torch.empty((n, C), dtype=F16, device=x.device) if out is None else out
    mc = desc.n_levels if mask_count is None else int(mask_count)
    with device_guard(x.device), _timed("hashgrid_forward", n):
        check(lib.nsr_hashgrid_forward(ptr(x), ptr(table_half), ptr(y), n, y.stride(0), mc, _byref(desc),
                                       stream_ptr()), "nsr_hashgrid_forward")
    return y


def hashgrid_backward_params(x, dy, grad_table, desc, mask_count=None, grad_scale=1.0, accumulate=True,
                             method="owner", level_major=False):
    """grad_table (+)= scatter(dy).  method "owner": atomic-free owner-computes kernel (default);
    "atomic": one lane per (sample, level) with global fp32 atomics (always accumulates)."""
    mc = desc.n_levels if mask_count is None else int(mask_count)
    n = x.shape[0]
    with device_guard(x.device), _timed("hashgrid_backward_params", n):
        if method == "atomic":
            if not accumulate:
                grad_table.zero_()
            check(lib.nsr_hashgrid_backward_params(ptr(x), ptr(dy), _is_f32(dy), dy.stride(0), ptr(grad_table), n, mc,
                                                   float(grad_scale), _byref(desc), stream_ptr()),
                  "nsr_hashgrid_backward_params")
        else:
            layout, stride = (2, 0) if level_major else (_is_f32(dy), dy.stride(0))
            nws = lib.nsr_hashgrid_backward_params_workspace_floats(_byref(desc), n)
            ws = torch.empty(int(nws), dtype=F32, device=x.device)
            check(lib.nsr_hashgrid_backward_params_owner(ptr(x), ptr(dy), layout, stride, ptr(grad_table), ptr(ws), n, mc,
                                                         float(grad_scale), int(bool(accumulate)), _byref(desc),
                                                         None, stream_ptr()), "nsr_hashgrid_backward_params_owner")
    return grad_table


def hashgrid_backward_input(x, table_half, dy, desc, mask_count=None):
    mc = desc.n_levels if mask_count is None else int(mask_count)
    dx = torch.empty((x.shape[0], 3), dtype=F32, device=x.device)
    with device_guard(x.device):
        check(lib.nsr_hashgrid_backward_input(ptr(x), ptr(table_half), ptr(dy), _is_f32(dy), dy.stride(0), ptr(dx),
                                              x.shape[0], mc, _byref(desc), stream_ptr()),
              "nsr_hashgrid_backward_input")
    return dx


def hashgrid_backward_backward_input(x, table_half, dy, g, desc, mask_count=None, want_d_dy=True,
                                     grad_table=None, want_dx2=True):
    mc = desc.n_levels if mask_count is None else int(mask_count)
    n, C = x.shape[0], desc.n_levels * desc.n_features
    d_dy = torch.empty((n, C), dtype=F32, device=x.device) if want_d_dy else None
    dx2 = torch.empty((n, 3), dtype=F32, device=x.device) if want_dx2 else None
    with device_guard(x.device):
        ws = None
        if grad_table is not None and n > 0:  # table part through the binned owner-computes path (no global atomics)
            nws = lib.nsr_hashgrid_backward_params_workspace_floats(_byref(desc), n)
            ws = torch.empty(int(nws), dtype=F32, device=x.device)
        check(lib.nsr_hashgrid_backward_backward_input_ws(
            ptr(x), ptr(table_half), ptr(dy), _is_f32(dy), dy.stride(0), ptr(g), ptr(d_dy), C, ptr(grad_table),
            ptr(dx2), ptr(ws), n, mc, _byref(desc), stream_ptr()), "nsr_hashgrid_backward_backward_input_ws")
    return d_dy, dx2


class _GridEncode(Function):
    """y = encode(x; params).  ``owner`` supplies desc / fp16 shadow / progressive mask count."""

    @staticmethod
    def forward(ctx, x, params, owner):
        table = owner.table_half(params)
        mc = owner.level_mask_count()
        y = hashgrid_forward(x, table, owner.grid_desc, mc)
        ctx.save_for_backward(x, params)
        ctx.owner, ctx.table, ctx.mc = owner, table, mc
        # dtype=float32 modules convert INSIDE the Function so the incoming gradient stays fp32
        return y.float() if owner.dtype == torch.float32 else y

    @staticmethod
    def backward(ctx, dy):
        x, params = ctx.saved_tensors
        need_x, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx, dp = _GridEncodeBackward.apply(ctx.owner, ctx.table, ctx.mc, need_x, need_p, x, params, dy.contiguous())
        return (dx if need_x else None), (dp if need_p else None), None


class _GridEncodeBackward(Function):
    """First-order backward as a Function so that d(dx)/d{dy, params, x} exists (NeuS eikonal term)."""

    @staticmethod
    def forward(ctx, owner, table, mc, need_x, need_p, x, params, dy):
        desc = owner.grid_desc
        dx = hashgrid_backward_input(x, table, dy, desc, mc) if need_x else torch.zeros_like(x)
        if need_p:
            dp = torch.empty_like(params, dtype=F32)  # the owner kernel writes every entry: no memset
            hashgrid_backward_params(x, dy, owner.grid_slice(dp), desc, mc, accumulate=False)
        else:
            dp = torch.zeros(1, dtype=F32, device=x.device)
        ctx.save_for_backward(x, params, dy)
        ctx.owner, ctx.table, ctx.mc = owner, table, mc
        return dx, dp

    @staticmethod
    @once_differentiable
    def backward(ctx, g_dx, g_dp):
        # only the dx branch is differentiated again (tcnn: "bwd_bwd_input"); d(dp)/d* is not needed by
        # the reference (no second-order term through the parameter gradient).
        x, params, dy = ctx.saved_tensors
        owner, desc = ctx.owner, ctx.owner.grid_desc
        need_x, need_p, need_dy = ctx.needs_input_grad[5], ctx.needs_input_grad[6], ctx.needs_input_grad[7]
        if g_dx is None or not (need_x or need_p or need_dy):
            return (None,) * 8
        g = _f32c(g_dx)
        dp2 = torch.zeros_like(params, dtype=F32) if need_p else None
        d_dy, dx2 = hashgrid_backward_backward_input(
            x, ctx.table, dy, g, desc, ctx.mc, want_d_dy=need_dy,
            grad_table=owner.grid_slice(dp2) if need_p else None, want_dx2=need_x)
        if d_dy is not None:
            d_dy = d_dy.to(dy.dtype)
        return None, None, None, None, None, dx2, dp2, d_dy


def grid_encode(x, params, owner):
    return _GridEncode.apply(x, params, owner)


# ------------------------------------------------------------------------------------------------
# spherical harmonics
# ------------------------------------------------------------------------------------------------
def sh4_forward(u, out=None):
    n = u.shape[0]
    y = torch.empty((n, 16), dtype=F16, device=u.device) if out is None else out
    with device_guard(u.device):
        check(lib.nsr_sh4_forward(ptr(u), ptr(y), n, y.stride(0), stream_ptr()), "nsr_sh4_forward")
    return y


# ------------------------------------------------------------------------------------------------
# fused MLP
# ------------------------------------------------------------------------------------------------
def mlp_forward(x, weights_half, desc, save_acts):
    n = x.shape[0]
    out = torch.empty((n, desc.out_pad), dtype=F16, device=x.device)
    acts = torch.empty((desc.n_hidden, n, 64), dtype=F16, device=x.device) if save_acts else None
    with device_guard(x.device), _timed(f"mlp_forward_h{desc.n_hidden}", n):
        check(lib.nsr_mlp_forward(ptr(x), _is_f32(x), x.stride(0), ptr(weights_half), ptr(out), ptr(acts), n,
                                  _byref(desc), stream_ptr()), "nsr_mlp_forward")
    return out, acts


def mlp_backward(dout, out, x, acts, weights_half, desc, grad_weights=None, want_dx=False, grad_scale=128.0):
    n = x.shape[0]
    dx = torch.empty((n, desc.n_in), dtype=F32, device=x.device) if want_dx else None
    partials = None
    if grad_weights is not None:
        nws = lib.nsr_mlp_backward_workspace_floats(_byref(desc), n)
        partials = torch.empty(int(nws), dtype=F32, device=x.device)
    with device_guard(x.device), _timed(f"mlp_backward_h{desc.n_hidden}", n):
        check(lib.nsr_mlp_backward(ptr(dout), _is_f32(dout), dout.stride(0), ptr(out), ptr(x), _is_f32(x), x.stride(0),
                                   ptr(acts), ptr(weights_half), ptr(grad_weights), ptr(dx),
                                   desc.n_in if want_dx else 0, ptr(partials), n, float(grad_scale), _byref(desc),
                                   stream_ptr()), "nsr_mlp_backward")
    return dx


class _Mlp(Function):
    @staticmethod
    def forward(ctx, x, params, owner, train):
        w = owner.weights_half(params)
        desc = owner.mlp_desc
        out, acts = mlp_forward(x, w, desc, save_acts=train)
        ctx.save_for_backward(x, params)
        ctx.owner, ctx.w, ctx.out, ctx.acts = owner, w, out, acts
        res = out[:, :desc.n_out]
        return res.float() if owner.dtype == torch.float32 else res

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, params = ctx.saved_tensors
        owner, desc = ctx.owner, ctx.owner.mlp_desc
        if ctx.acts is None:
            raise NsrError("MLP backward without saved activations (forward ran with grad disabled)")
        need_x, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dout = dout.contiguous()
        dp = torch.zeros_like(params, dtype=F32) if need_p else None
        dx = mlp_backward(dout, ctx.out, x, ctx.acts, ctx.w, desc, grad_weights=owner.mlp_slice(dp) if need_p else None,
                          want_dx=need_x, grad_scale=owner.loss_scale)
        if dx is not None and dx.dtype != x.dtype:
            dx = dx.to(x.dtype)
        return dx, dp, None, None


def _wants_grad(x, params):
    # Function.forward runs with grad mode off, so "do we need to save activations" is decided here
    return torch.is_grad_enabled() and (x.requires_grad or params.requires_grad)


def mlp(x, params, owner):
    return _Mlp.apply(x, params, owner, _wants_grad(x, params))


# Launches of at most this many samples take the one-kernel encode -> MLP forward (csrc/gridmlp.hip); larger ones the
# XCD-placed encode + MLP pair.  Measured (tools/grid_mlp_ab.py, profiles/r03_grid_mlp_ab.json): 19 vs 27 us at 2,048
# samples, 21-25 vs 26 us at 8,192; even at 32,768 for ray-ordered positions; from there on the pair wins for unordered
# positions (262,144 uniform samples: 160 vs 183 us) and ties for ray-ordered ones.
GRID_MLP_FUSED_MAX_N = int(os.environ.get("NSR_GRID_MLP_FUSED_MAX_N", "16384"))


def grid_mlp_supported(gdesc, mdesc):
    return bool(lib.nsr_grid_mlp_supported(_byref(gdesc), _byref(mdesc)))


def grid_mlp_forward(x, table_half, weights_half, gdesc, mdesc, mask_count=None, save_acts=False, want_enc=False,
                     enc_level_major=False):
    """one kernel: out [n,16] half, acts (or None), enc (or None; row-major [n, L*F] or level-major [L, n, F])"""
    n = x.shape[0]
    mc = gdesc.n_levels if mask_count is None else int(mask_count)
    C = gdesc.n_levels * gdesc.n_features
    out = torch.empty((n, mdesc.out_pad), dtype=F16, device=x.device)
    acts = torch.empty((mdesc.n_hidden, n, 64), dtype=F16, device=x.device) if save_acts else None
    enc = None
    if want_enc:
        enc = torch.empty((gdesc.n_levels, n, gdesc.n_features) if enc_level_major else (n, C), dtype=F16, device=x.device)
    with device_guard(x.device), _timed("grid_mlp_forward", n):
        check(lib.nsr_grid_mlp_forward(ptr(x), ptr(table_half), ptr(weights_half), ptr(out), ptr(acts), ptr(enc),
                                       0 if (enc is None or enc_level_major) else enc.stride(0), int(enc_level_major), n, mc,
                                       _byref(gdesc), _byref(mdesc), None, stream_ptr()), "nsr_grid_mlp_forward")
    return out, acts, enc


def grid_mlp_backward(dout, out, x, enc, acts, weights_half, gdesc, mdesc, grad_weights, grad_table, mask_count=None,
                      grad_scale=128.0, enc_level_major=False):
    """dgrad (level-major d_enc) + weight gradients (accumulated into grad_weights) + owner-computes table backward
    (grad_table overwritten) in one call"""
    n = x.shape[0]
    mc = gdesc.n_levels if mask_count is None else int(mask_count)
    nws = lib.nsr_grid_mlp_backward_workspace_floats(_byref(gdesc), _byref(mdesc), n)
    ws = torch.empty(int(nws), dtype=F32, device=x.device)
    with device_guard(x.device), _timed("grid_mlp_backward", n):
        check(lib.nsr_grid_mlp_backward(ptr(dout), _is_f32(dout), dout.stride(0), ptr(out), ptr(x), ptr(enc),
                                        0 if enc_level_major else enc.stride(0), int(enc_level_major), ptr(acts),
                                        ptr(weights_half), ptr(grad_weights), ptr(grad_table), ptr(ws), n, mc,
                                        float(grad_scale), _byref(gdesc), _byref(mdesc), stream_ptr()),
              "nsr_grid_mlp_backward")


class _GridMlp(Function):
    """tcnn.NetworkWithInputEncoding: encode -> MLP with ONE flat parameter ([network | grid])."""

    @staticmethod
    def forward(ctx, x, params, owner, train):
        table, w = owner.table_half(params), owner.weights_half(params)
        fusable = grid_mlp_supported(owner.grid_desc, owner.mlp_desc)
        if fusable and x.shape[0] <= GRID_MLP_FUSED_MAX_N:
            out, acts, enc = grid_mlp_forward(x, table, w, owner.grid_desc, owner.mlp_desc, owner.level_mask_count(),
                                              save_acts=train, want_enc=train)
        else:
            enc = hashgrid_forward(x, table, owner.grid_desc, owner.level_mask_count())
            out, acts = mlp_forward(enc, w, owner.mlp_desc, save_acts=train)
        ctx.save_for_backward(x, params)
        ctx.owner, ctx.table, ctx.w, ctx.enc, ctx.out, ctx.acts, ctx.fusable = owner, table, w, enc, out, acts, fusable
        res = out[:, :owner.mlp_desc.n_out]
        return res.float() if owner.dtype == torch.float32 else res

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, params = ctx.saved_tensors
        owner = ctx.owner
        if ctx.acts is None:
            raise NsrError("backward without saved activations (forward ran with grad disabled)")
        need_x, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dp = None
        if need_p:  # only the (tiny) MLP slice needs zeroing; the grid slice is overwritten by the owner kernel
            dp = torch.empty_like(params, dtype=F32)
            owner.mlp_slice(dp).zero_()
        if need_p and not need_x and ctx.fusable:
            # the MLP's data gradient leaves level-major, as the table backward reads it: no [n, L*F] fp32 round trip, no
            # transpose kernel
            grid_mlp_backward(dout.contiguous(), ctx.out, x, ctx.enc, ctx.acts, ctx.w, owner.grid_desc, owner.mlp_desc,
                              owner.mlp_slice(dp), owner.grid_slice(dp), owner.level_mask_count(),
                              grad_scale=owner.loss_scale)
            return None, dp, None, None
        # grads w.r.t. the encoding leave the MLP in fp32 and feed the table backward directly
        d_enc = mlp_backward(dout.contiguous(), ctx.out, ctx.enc, ctx.acts, ctx.w, owner.mlp_desc,
                             grad_weights=owner.mlp_slice(dp) if need_p else None, want_dx=True,
                             grad_scale=owner.loss_scale)
        if need_p:
            hashgrid_backward_params(x, d_enc, owner.grid_slice(dp), owner.grid_desc, owner.level_mask_count(),
                                     accumulate=False)
        dx = hashgrid_backward_input(x, ctx.table, d_enc, owner.grid_desc, owner.level_mask_count()) if need_x else None
        return dx, dp, None, None


def grid_mlp(x, params, owner):
    return _GridMlp.apply(x, params, owner, _wants_grad(x, params))


# ------------------------------------------------------------------------------------------------
# marching / contraction / packing
# ------------------------------------------------------------------------------------------------
def ray_aabb_intersect(rays_o, rays_d, aabb):
    n = rays_o.shape[0]
    t_min = torch.empty(n, dtype=F32, device=rays_o.device)
    t_max = torch.empty(n, dtype=F32, device=rays_o.device)
    with device_guard(rays_o.device):
        check(lib.nsr_ray_aabb_intersect(ptr(rays_o), ptr(rays_d), ptr(aabb), ptr(t_min), ptr(t_max), n, stream_ptr()),
              "nsr_ray_aabb_intersect")
    return t_min, t_max


def grid_bricks(binary, out=None):
    """4x4x4-brick bit packing of a bool grid, cached on the tensor object (keyed by its version counter).
    None when the resolution is not brick-able (then the byte-grid kernels are used).  ``out``: pack into this
    persistent int64 buffer (a captured step keeps reading the same address after the grid is refreshed)."""
    rx, ry, rz = (int(s) for s in binary.shape)
    words = int(lib.nsr_grid_bricks_words64(rx, ry, rz))
    if words == 0 or ((rx >> 2) * (ry >> 2) * (rz >> 2)) % 64 != 0:
        return None
    tag = getattr(binary, "_nsr_bricks", None)
    if tag is not None and tag[0] == binary._version and tag[1] == binary.data_ptr() and (out is None or tag[2] is out):
        return tag[2]
    grid_u8 = binary.view(torch.uint8) if binary.dtype == torch.bool else binary
    bricks = out if out is not None else torch.empty(words, dtype=torch.int64, device=binary.device)
    with device_guard(binary.device):
        check(lib.nsr_grid_pack_bricks(ptr(grid_u8), rx, ry, rz, ptr(bricks), stream_ptr()), "nsr_grid_pack_bricks")
    try:
        binary._nsr_bricks = (binary._version, binary.data_ptr(), bricks)
    except Exception:  # noqa: BLE001
        pass
    return bricks


_PINNED = []


def _pinned_int32():
    """ring of pinned host words for the sample-count read-back (hipHostMalloc costs ~100 us: never per step)"""
    if not _PINNED:
        _PINNED.extend([torch.empty(2, dtype=torch.int32, pin_memory=True) for _ in range(8)] + [0])
    _PINNED[-1] = (_PINNED[-1] + 1) % 8
    return _PINNED[_PINNED[-1]]


SPIN_SECONDS = [0.0]  # total host time spent waiting for device counts (tools/neus_operating_point.py reads it)


def _spin_until_changed(host_word, sentinel, fallback_sync):
    t_begin = time.perf_counter()
    try:
        return _spin_until_changed_(host_word, sentinel, fallback_sync)
    finally:
        SPIN_SECONDS[0] += time.perf_counter() - t_begin


def _spin_until_changed_(host_word, sentinel, fallback_sync):
    spins = 0
    while int(host_word[0]) == sentinel:
        spins += 1
        if spins > 2000000:  # ~seconds: fall back to the blocking wait so errors surface
            break
    # the D2H blit may write the word BYTE-wise (a torn 0x800000DB was observed): the spin only tells us the copy has
    # started; the (now nearly free) synchronize makes it complete before the value is read
    fallback_sync()
    return int(host_word[0])


def read_count_when_ready(dev_word):
    """device int32 -> host int with a low-latency wait: async copy into a pinned word pre-set to a sentinel, spin
    until it starts changing, then finish with a stream synchronize.  A cold synchronize sleeps on an interrupt (tens of
    microseconds to wake up) and the step has two such read-backs on its critical path."""
    host = _pinned_int32()
    host[0] = -0x7fffffff
    host.copy_(dev_word, non_blocking=True)
    return _spin_until_changed(host, -0x7fffffff, torch.cuda.current_stream().synchronize)


def read_count_begin(dev_word):
    """first half of ``read_count_when_ready``: queue the copy into a pinned word on the current stream and return at once;
    ``read_count_finish`` waits.  Between the two the host may queue work that does not depend on the count."""
    host = _pinned_int32()
    host[0] = -0x7fffffff
    host[0:1].copy_(dev_word, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    return host, ev


def read_count_ready(pending):
    """has the copy queued by ``read_count_begin`` started to land?  (no wait; ``read_count_finish`` is then nearly free)"""
    return int(pending[0][0]) != -0x7fffffff


def read_count_finish(pending):
    host, ev = pending
    return _spin_until_changed(host, -0x7fffffff, ev.synchronize)


class MarchHandle:
    """an in-flight marching pass (count + per-ray scratch rows); ``ray_march_finish`` turns it into packed samples"""
    __slots__ = ("args", "counts", "packed", "total", "total_host", "scratch", "cap", "bricks", "grid_u8", "event",
                 "stream", "n", "host_words")


def ray_march_begin(rays_o, rays_d, t_min, t_max, roi, binary, contraction, step, cone_angle, roi_host=None,
                    method="bricks"):
    """enqueue the marching pass on the CURRENT stream without any host sync; the sample total is copied to pinned
    host memory behind it, so a caller can run this on a side stream underneath other work."""
    h = MarchHandle()
    n = h.n = rays_o.shape[0]
    dev = rays_o.device
    rx, ry, rz = (int(s) for s in binary.shape)
    h.args = (rays_o, rays_d, t_min, t_max, roi, (rx, ry, rz), int(contraction), float(step), float(cone_angle))
    h.counts = torch.empty(n, dtype=torch.int32, device=dev)
    h.packed = torch.empty((n, 2), dtype=torch.int32, device=dev)
    h.total = torch.zeros(1, dtype=torch.int32, device=dev)
    h.host_words = _pinned_int32()  # [0] sample total, [1] "a ray overflowed its scratch row"
    h.total_host = h.host_words[0:1]
    h.bricks = grid_bricks(binary) if method == "bricks" else None
    h.grid_u8 = binary.view(torch.uint8) if binary.dtype == torch.bool else binary
    h.cap, h.scratch = 0, None
    with device_guard(dev):
        s = stream_ptr()
        with _timed("ray_march_count", n):
            if h.bricks is None:
                check(lib.nsr_ray_march_count(ptr(rays_o), ptr(rays_d), ptr(t_min), ptr(t_max), ptr(roi), ptr(h.grid_u8),
                                              rx, ry, rz, int(contraction), float(step), float(cone_angle),
                                              ptr(h.counts), n, s), "nsr_ray_march_count")
            else:
                if roi_host is not None and int(contraction) == 0:
                    h.cap = int(lib.nsr_ray_march_capacity((ctypes.c_float * 6)(*[float(v) for v in roi_host]),
                                                           float(step)))
                h.scratch = torch.empty(n * h.cap * 2, dtype=F32, device=dev) if h.cap > 0 else None
                check(lib.nsr_ray_march_bricks_count(ptr(rays_o), ptr(rays_d), ptr(t_min), ptr(t_max), ptr(roi),
                                                     ptr(h.bricks), rx, ry, rz, int(contraction), float(step),
                                                     float(cone_angle), ptr(h.counts), ptr(h.scratch), h.cap, n, s),
                      "nsr_ray_march_bricks_count")
        check(lib.nsr_pack_from_counts(ptr(h.counts), ptr(h.packed), ptr(h.total), n, s), "nsr_pack_from_counts")
        h.total_host[0] = -0x7fffffff
        h.host_words[1] = 0
        if h.scratch is not None:
            # the overflow test of the single-pass scratch (see ray_march_finish) rides along with the count: reading it on
            # the consumer's stream would make the host wait for everything queued there
            h.host_words[1:2].copy_((h.counts > h.cap).any().to(torch.int32).reshape(1), non_blocking=True)
        h.total_host.copy_(h.total, non_blocking=True)  # last: the spin in ray_march_finish watches this word
        h.stream = torch.cuda.current_stream()
        h.event = torch.cuda.Event()
        h.event.record(h.stream)
    return h


def ray_march_finish(h):
    """wait for the marching pass (only ITS stream), then pack the samples on the current stream.
    -> packed_info, ray_indices, t_starts, t_ends"""
    rays_o, rays_d, t_min, t_max, roi, (rx, ry, rz), contraction, step, cone_angle = h.args
    dev, n = rays_o.device, h.n
    m = _spin_until_changed(h.total_host, -0x7fffffff, h.event.synchronize)  # the marcher's one intrinsic host sync
    cur = torch.cuda.current_stream()
    if cur != h.stream:  # tensors born on the marching stream are consumed here
        cur.wait_event(h.event)
        for t in (h.counts, h.packed, h.total, h.scratch, rays_o, rays_d, t_min, t_max):
            if t is not None:
                t.record_stream(cur)
    ray_indices = torch.empty(m, dtype=torch.int64, device=dev)
    t_starts = torch.empty((m, 1), dtype=F32, device=dev)
    t_ends = torch.empty((m, 1), dtype=F32, device=dev)
    scratch = h.scratch
    if m > 0 and scratch is not None and int(h.host_words[1]):
        # a ray emitted more samples than its scratch row holds (the diag/step+3 bound assumes unit-length directions
        # and the roi the capacity was derived from): the counts are still exact, so re-march in two-pass mode
        scratch = None
    if m > 0:
        with device_guard(dev), _timed("ray_march_write", n):
            s = stream_ptr()
            if h.bricks is None:
                check(lib.nsr_ray_march_write(ptr(rays_o), ptr(rays_d), ptr(t_min), ptr(t_max), ptr(roi), ptr(h.grid_u8),
                                              rx, ry, rz, contraction, step, cone_angle, ptr(h.packed), ptr(ray_indices),
                                              ptr(t_starts), ptr(t_ends), n, s), "nsr_ray_march_write")
            else:
                check(lib.nsr_ray_march_bricks_write(ptr(rays_o), ptr(rays_d), ptr(t_min), ptr(t_max), ptr(roi),
                                                     ptr(h.bricks), rx, ry, rz, contraction, step, cone_angle,
                                                     ptr(h.packed), ptr(scratch), h.cap, ptr(ray_indices),
                                                     ptr(t_starts), ptr(t_ends), n, s), "nsr_ray_march_bricks_write")
    return h.packed, ray_indices, t_starts, t_ends


def ray_march(rays_o, rays_d, t_min, t_max, roi, binary, contraction, step, cone_angle, roi_host=None,
              method="bricks"):
    """occupancy-grid marching; ONE host sync (the sample count).  -> packed_info, ray_indices, t_starts, t_ends.

    method "bricks" (default): bit-packed 4^3 bricks + LDS any-bits, single marching pass into per-ray scratch when
    the sample capacity is provable (AABB contraction with a finite roi: pass ``roi_host`` = 6 python floats);
    method "bytes": the plain two-pass byte-grid kernels.  Both are bit-exact against the oracle."""
    return ray_march_finish(ray_march_begin(rays_o, rays_d, t_min, t_max, roi, binary, contraction, step, cone_angle,
                                            roi_host=roi_host, method=method))


def pack_info(ray_indices, n_rays):
    packed = torch.empty((n_rays, 2), dtype=torch.int32, device=ray_indices.device)
    with device_guard(ray_indices.device):
        check(lib.nsr_pack_info(ptr(ray_indices), ptr(packed), ray_indices.shape[0], n_rays, stream_ptr()),
              "nsr_pack_info")
    return packed


def contract(x, roi, contraction, inverse=False):
    out = torch.empty_like(x)
    fn = lib.nsr_contract_inv if inverse else lib.nsr_contract
    with device_guard(x.device):
        check(fn(ptr(x), ptr(roi), int(contraction), ptr(out), x.shape[0], stream_ptr()), "nsr_contract")
    return out


def grid_query(x, roi, binary, contraction):
    rx, ry, rz = (int(s) for s in binary.shape)
    grid_u8 = binary.view(torch.uint8) if binary.dtype == torch.bool else binary
    out = torch.empty(x.shape[0], dtype=torch.uint8, device=x.device)
    with device_guard(x.device):
        check(lib.nsr_grid_query_u8(ptr(x), ptr(roi), ptr(grid_u8), rx, ry, rz, int(contraction), ptr(out), x.shape[0],
                                    stream_ptr()), "nsr_grid_query_u8")
    return out.bool()


def sample_positions(rays_o, rays_d, ray_indices, t_starts, t_ends, want_dirs=True):
    n = ray_indices.shape[0]
    pos = torch.empty((n, 3), dtype=F32, device=rays_o.device)
    dirs = torch.empty((n, 3), dtype=F32, device=rays_o.device) if want_dirs else None
    with device_guard(rays_o.device):
        check(lib.nsr_sample_positions(ptr(rays_o), ptr(rays_d), ptr(ray_indices), ptr(t_starts), ptr(t_ends), ptr(pos),
                                       ptr(dirs), n, stream_ptr()), "nsr_sample_positions")
    return pos, dirs


def compact_samples(mask, ray_indices, t_starts, t_ends):
    """order-preserving compaction by a bool mask; one host sync for the kept count."""
    n = ray_indices.shape[0]
    dev = ray_indices.device
    mask_u8 = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
    ri = torch.empty_like(ray_indices)
    t0, t1 = torch.empty_like(t_starts), torch.empty_like(t_ends)
    n_kept = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = torch.empty((n + 255) // 256 + 1, dtype=torch.int32, device=dev)
    with device_guard(dev):
        check(lib.nsr_compact_samples(ptr(mask_u8), ptr(ray_indices), ptr(t_starts), ptr(t_ends), ptr(ri), ptr(t0),
                                      ptr(t1), ptr(n_kept), ptr(scratch), n, stream_ptr()), "nsr_compact_samples")
    k = int(n_kept.item())
    return ri[:k], t0[:k], t1[:k]


# ------------------------------------------------------------------------------------------------
# compositing
# ------------------------------------------------------------------------------------------------
class _TransFromSigma(Function):
    @staticmethod
    def forward(ctx, sigmas, t_starts, t_ends, packed, n_rays):
        sig = _f32c(sigmas)
        T = torch.empty_like(sig)
        with device_guard(sig.device):
            check(lib.nsr_transmittance_from_sigma_forward(ptr(packed), ptr(t_starts), ptr(t_ends), ptr(sig), ptr(T),
                                                           n_rays, stream_ptr()), "transmittance_from_sigma_forward")
        ctx.save_for_backward(T, t_starts, t_ends, packed)
        ctx.n_rays = n_rays
        return T

    @staticmethod
    @once_differentiable
    def backward(ctx, gT):
        T, t_starts, t_ends, packed = ctx.saved_tensors
        g = torch.empty_like(T)
        gT = _f32c(gT)
        with device_guard(T.device):
            check(lib.nsr_transmittance_from_sigma_backward(ptr(packed), ptr(t_starts), ptr(t_ends), ptr(T), ptr(gT),
                                                            ptr(g), ctx.n_rays, stream_ptr()),
                  "transmittance_from_sigma_backward")
        return g, None, None, None, None


class _TransFromAlpha(Function):
    @staticmethod
    def forward(ctx, alphas, packed, n_rays):
        a = _f32c(alphas)
        T = torch.empty_like(a)
        with device_guard(a.device):
            check(lib.nsr_transmittance_from_alpha_forward(ptr(packed), ptr(a), ptr(T), n_rays, stream_ptr()),
                  "transmittance_from_alpha_forward")
        ctx.save_for_backward(T, a, packed)
        ctx.n_rays = n_rays
        return T

    @staticmethod
    @once_differentiable
    def backward(ctx, gT):
        T, a, packed = ctx.saved_tensors
        g = torch.empty_like(T)
        gT = _f32c(gT)
        with device_guard(T.device):
            check(lib.nsr_transmittance_from_alpha_backward(ptr(packed), ptr(a), ptr(T), ptr(gT), ptr(g), ctx.n_rays,
                                                            stream_ptr()), "transmittance_from_alpha_backward")
        return g, None, None


def transmittance_from_sigma(sigmas, t_starts, t_ends, packed, n_rays):
    return _TransFromSigma.apply(sigmas, t_starts.contiguous(), t_ends.contiguous(), packed, n_rays)


def transmittance_from_alpha(alphas, packed, n_rays):
    return _TransFromAlpha.apply(alphas, packed, n_rays)


class _Accumulate(Function):
    @staticmethod
    def forward(ctx, weights, values, ray_indices, packed, n_rays):
        w = _f32c(weights).view(-1)
        v = None if values is None else _f32c(values)
        dim = 1 if v is None else v.shape[-1]
        out = torch.empty((n_rays, dim), dtype=F32, device=w.device)
        with device_guard(w.device):
            check(lib.nsr_accumulate_along_rays_forward(ptr(packed), ptr(w), ptr(v), dim, ptr(out), n_rays,
                                                        stream_ptr()), "accumulate_along_rays_forward")
        ctx.save_for_backward(w, v, ray_indices)
        ctx.dim, ctx.wshape = dim, weights.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        w, v, ray_indices = ctx.saved_tensors
        need_w, need_v = ctx.needs_input_grad[0], (v is not None and ctx.needs_input_grad[1])
        g_out = _f32c(g_out)
        gw = torch.empty_like(w) if need_w else None
        gv = torch.empty_like(v) if need_v else None
        with device_guard(w.device):
            check(lib.nsr_accumulate_along_rays_backward(ptr(ray_indices), ptr(w), ptr(v), ctx.dim, ptr(g_out), ptr(gw),
                                                         ptr(gv), w.shape[0], stream_ptr()),
                  "accumulate_along_rays_backward")
        return (gw.view(ctx.wshape) if need_w else None), gv, None, None, None


def accumulate_along_rays(weights, values, ray_indices, packed, n_rays):
    return _Accumulate.apply(weights, values, ray_indices, packed, n_rays)


# ------------------------------------------------------------------------------------------------
# fused reference glue
# ------------------------------------------------------------------------------------------------
def contract_to_unisphere(x, radius, contraction):
    out = torch.empty_like(x)
    with device_guard(x.device):
        check(lib.nsr_contract_to_unisphere(ptr(x), float(radius), int(contraction), ptr(out), x.shape[0], stream_ptr()),
              "nsr_contract_to_unisphere")
    return out


def density_activation(mlp_out, n_feat, bias, want_feature=True):
    n = mlp_out.shape[0]
    density = torch.empty(n, dtype=F32, device=mlp_out.device)
    feature = torch.empty((n, n_feat), dtype=F32, device=mlp_out.device) if want_feature else None
    with device_guard(mlp_out.device):
        check(lib.nsr_density_activation_forward(ptr(mlp_out), mlp_out.stride(0), n_feat, float(bias), ptr(density),
                                                 ptr(feature), n, stream_ptr()), "nsr_density_activation_forward")
    return density, feature


class _NeusAlpha(Function):
    @staticmethod
    def forward(ctx, sdf, normal, dirs, dists, inv_s, anneal):
        sdf, normal, dirs, dists = _f32c(sdf).view(-1), _f32c(normal), _f32c(dirs), _f32c(dists).view(-1)
        inv_s_c = _f32c(inv_s).view(-1)[:1]
        alpha = torch.empty_like(sdf)
        with device_guard(sdf.device):
            check(lib.nsr_neus_alpha_forward(ptr(sdf), ptr(normal), ptr(dirs), ptr(dists), ptr(inv_s_c), float(anneal),
                                             ptr(alpha), sdf.shape[0], stream_ptr()), "nsr_neus_alpha_forward")
        ctx.save_for_backward(sdf, normal, dirs, dists, inv_s_c)
        ctx.anneal, ctx.inv_s_shape = float(anneal), inv_s.shape
        return alpha

    @staticmethod
    @once_differentiable
    def backward(ctx, g_alpha):
        sdf, normal, dirs, dists, inv_s_c = ctx.saved_tensors
        g_alpha = _f32c(g_alpha).view(-1)
        g_sdf, g_normal = torch.empty_like(sdf), torch.empty_like(normal)
        g_inv_s = torch.zeros(1, dtype=F32, device=sdf.device)
        with device_guard(sdf.device):
            check(lib.nsr_neus_alpha_backward(ptr(sdf), ptr(normal), ptr(dirs), ptr(dists), ptr(inv_s_c), ctx.anneal,
                                              ptr(g_alpha), ptr(g_sdf), ptr(g_normal), ptr(g_inv_s), sdf.shape[0],
                                              stream_ptr()), "nsr_neus_alpha_backward")
        return g_sdf, g_normal, None, None, g_inv_s.view(ctx.inv_s_shape), None


def neus_alpha(sdf, normal, dirs, dists, inv_s, cos_anneal_ratio):
    return _NeusAlpha.apply(sdf, normal, dirs, dists, inv_s, cos_anneal_ratio)


def adamw_step(params, grad, exp_avg, exp_avg_sq, shadow_half, lr, beta1, beta2, eps, weight_decay, step,
               grad_unscale=1.0, zero_grad=True, hyper=None, zero_first_n=0):
    """``hyper`` (device float[3] written by ``adam_tick``) overrides lr and the bias corrections on the device"""
    bc1, bc2 = 1.0 - beta1 ** max(step, 1), 1.0 - beta2 ** max(step, 1)
    with device_guard(params.device):
        check(lib.nsr_adamw_step(ptr(params), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), ptr(shadow_half),
                                 params.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                 float(bc1), float(bc2), float(grad_unscale), int(zero_grad), ptr(hyper),
                                 int(zero_first_n), stream_ptr()), "nsr_adamw_step")


def adamw_step_scheduled(tensors, step_dev, hyper12, base_lr, beta1, beta2, gamma, milestones, eps, weight_decay,
                         grad_unscale=1.0, zero_grad=True, out=None, stream=None):
    """``adam_tick`` + ``adamw_step`` over one or two tensors in ONE launch (bit-identical).  ``tensors``: list of
    (params, grad, exp_avg, exp_avg_sq, shadow_half, zero_first_n); ``hyper12``: 12 zero-initialised floats; ``out``:
    (step_dev, hyper12) that receive the advanced schedule state (default: in place)"""
    assert 1 <= len(tensors) <= 2 and hyper12.numel() >= 12
    step_out, hyper_out = out if out is not None else (step_dev, hyper12)
    ms = [int(m) for m in milestones][:3] + [0x7fffffff] * (3 - min(len(milestones), 3))
    a = tensors[0]
    b = tensors[1] if len(tensors) == 2 else (None,) * 5 + (0,)
    with device_guard(a[0].device):
        check(lib.nsr_adamw_step_scheduled_to(ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), ptr(a[4]), a[0].numel(), int(a[5]),
                                              ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(b[3]), ptr(b[4]),
                                              0 if b[0] is None else b[0].numel(), ptr(step_dev), ptr(hyper12),
                                              ptr(step_out), ptr(hyper_out),
                                              float(base_lr), float(beta1), float(beta2), float(gamma), ms[0], ms[1], ms[2],
                                              float(eps), float(weight_decay), float(grad_unscale), int(zero_grad),
                                              stream if stream is not None else stream_ptr()), "nsr_adamw_step_scheduled")


def adam_tick(step_dev, hyper_dev, base_lr, beta1, beta2, gamma, milestones):
    ms = [int(m) for m in milestones][:3] + [0x7fffffff] * (3 - min(len(milestones), 3))
    with device_guard(step_dev.device):
        check(lib.nsr_adam_tick(ptr(step_dev), ptr(hyper_dev), float(base_lr), float(beta1), float(beta2), float(gamma),
                                ms[0], ms[1], ms[2], stream_ptr()), "nsr_adam_tick")
