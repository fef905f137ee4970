"""``nerfacc.OccupancyGrid`` (reference: constructed models/nerf.py:37, models/neus.py:64,70; refreshed
models/nerf.py:55, models/neus.py:109-111).

An ``nn.Module`` so its buffers (``_roi_aabb``, ``_binary``, ``resolution``, ``occs``) ride in the reference's
``state_dict`` and follow ``.to(device)``.  The cell-sampling policy and EMA are host logic on torch tensors
(identical RNG calls to nerfacc 0.3.3); un-contraction of the cell centres and grid queries are HIP kernels.
"""
import torch

from nsr_hip import ops as _ops

from .contraction import ContractionType, contract_inv


class Grid(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("_dummy", torch.empty(0), persistent=False)

    @property
    def device(self):
        return self._dummy.device


class OccupancyGrid(Grid):
    NUM_DIM = 3

    def __init__(self, roi_aabb, resolution=128, contraction_type=ContractionType.AABB):
        super().__init__()
        if isinstance(resolution, int):
            resolution = [resolution] * self.NUM_DIM
        if isinstance(resolution, (list, tuple)):
            resolution = torch.tensor(resolution, dtype=torch.int32)
        if not isinstance(resolution, torch.Tensor) or resolution.shape != (self.NUM_DIM,):
            raise ValueError(f"Invalid resolution: {resolution}")
        if isinstance(roi_aabb, (list, tuple)):
            roi_aabb = torch.tensor(roi_aabb, dtype=torch.float32)
        if not isinstance(roi_aabb, torch.Tensor) or roi_aabb.shape != (self.NUM_DIM * 2,):
            raise ValueError(f"Invalid roi_aabb: {roi_aabb}")
        self.num_cells = int(resolution.prod().item())
        self._res = [int(v) for v in resolution.tolist()]
        self.register_buffer("_roi_aabb", roi_aabb.detach().clone().float())
        self.register_buffer("_binary", torch.zeros(self._res, dtype=torch.bool))
        self._contraction_type = contraction_type
        self.register_buffer("resolution", resolution.clone())
        self.register_buffer("occs", torch.zeros(self.num_cells))
        self._roi_host = [float(v) for v in roi_aabb.detach().cpu().tolist()]  # host copy: sample-capacity bound

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        key = prefix + "_roi_aabb"
        if key in state_dict:  # the host copy bounds the marcher's per-ray scratch rows: keep it in step with the buffer
            self._roi_host = [float(v) for v in state_dict[key].detach().cpu().tolist()]

    @property
    def roi_aabb(self):
        return self._roi_aabb

    @property
    def binary(self):
        return self._binary

    @property
    def contraction_type(self):
        return self._contraction_type

    # ---- cell selection (same torch RNG calls as nerfacc 0.3.3) -------------------------------------
    @torch.no_grad()
    def _get_all_cells(self):
        return torch.arange(self.num_cells, device=self.device)

    @torch.no_grad()
    def _sample_uniform_and_occupied_cells(self, n):
        uniform_indices = torch.randint(self.num_cells, (n,), device=self.device)
        occupied_indices = torch.nonzero(self._binary.flatten())[:, 0]
        if n < len(occupied_indices):
            selector = torch.randint(len(occupied_indices), (n,), device=self.device)
            occupied_indices = occupied_indices[selector]
        return torch.cat([uniform_indices, occupied_indices], dim=0)

    def _cell_coords(self, indices):
        ry, rz = self._res[1], self._res[2]
        return torch.stack([indices // (ry * rz), (indices // rz) % ry, indices % rz], dim=-1)

    @torch.no_grad()
    def _update_cells(self, indices, jitter, occ_eval_fn, occ_thre=0.01, ema_decay=0.95):
        """deterministic part of the refresh: ``jitter`` in [0,1)^3 per selected cell"""
        x = (self._cell_coords(indices) + jitter) / self.resolution
        if self._contraction_type == ContractionType.UN_BOUNDED_SPHERE:
            mask = (x - 0.5).norm(dim=1) < 0.5  # only points inside the sphere are valid
            x, indices = x[mask], indices[mask]
        x = contract_inv(x, roi=self._roi_aabb, type=self._contraction_type)
        occ = occ_eval_fn(x).squeeze(-1)
        self.occs[indices] = torch.maximum(self.occs[indices] * ema_decay, occ.to(self.occs.dtype))
        self._binary = (self.occs > torch.clamp(self.occs.mean(), max=occ_thre)).view(self._binary.shape)

    @torch.no_grad()
    def _update(self, step, occ_eval_fn, occ_thre=0.01, ema_decay=0.95, warmup_steps=256):
        if step < warmup_steps:
            indices = self._get_all_cells()
        else:
            indices = self._sample_uniform_and_occupied_cells(self.num_cells // 4)
        jitter = torch.rand((indices.shape[0], self.NUM_DIM), dtype=torch.float32, device=self.device)
        self._update_cells(indices, jitter, occ_eval_fn, occ_thre, ema_decay)

    @torch.no_grad()
    def every_n_step(self, step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16):
        if not self.training:
            raise RuntimeError("You should only call this function only during training. Please call _update() "
                               "directly if you want to update the field during inference.")
        if step % n == 0 and self.training:
            self._update(step=step, occ_eval_fn=occ_eval_fn, occ_thre=occ_thre, ema_decay=ema_decay,
                         warmup_steps=warmup_steps)

    @torch.no_grad()
    def query_occ(self, samples):
        return query_grid(samples, self._roi_aabb, self.binary, self.contraction_type)


@torch.no_grad()
def query_grid(samples, grid_roi, grid_values, grid_type):
    if grid_values.dtype != torch.bool:
        raise NotImplementedError("query_grid(gfx950): only boolean grids are implemented")
    return _ops.grid_query(samples.float().contiguous(), grid_roi.float().contiguous(), grid_values.contiguous(),
                           grid_type.value)
