"""``nerfacc.ContractionType`` / ``contract`` / ``contract_inv`` (reference use: models/geometry.py:14,18-20)."""
import enum

import torch

from nsr_hip import ops as _ops


class ContractionType(enum.Enum):
    """AABB: linear map of the roi to [0,1]^3.  UN_BOUNDED_SPHERE: mip-NeRF-360 contraction, roi maps to the
    ball of radius 0.25 around 0.5.  UN_BOUNDED_TANH: roi maps to [0.25,0.75]^3 through tanh."""
    AABB = 0
    UN_BOUNDED_TANH = 1
    UN_BOUNDED_SPHERE = 2

    def to_cpp_version(self):
        return self.value


@torch.no_grad()
def contract(x, roi, type=ContractionType.AABB):
    return _ops.contract(x.float().contiguous(), roi.float().contiguous(), type.value, inverse=False)


@torch.no_grad()
def contract_inv(x, roi, type=ContractionType.AABB):
    return _ops.contract(x.float().contiguous(), roi.float().contiguous(), type.value, inverse=True)
