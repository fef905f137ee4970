"""Drop-in ``nerfacc`` (0.3.3 surface) for bennyguo/instant-nsr-pl on AMD MI355X (gfx950).

Exports what the reference imports (``models/nerf.py:11``, ``models/neus.py:11-12``, ``models/geometry.py:14``):
``ContractionType, OccupancyGrid, ray_marching, render_weight_from_density, render_weight_from_alpha,
accumulate_along_rays`` and ``nerfacc.intersection.ray_aabb_intersect`` -- plus the helpers nerfacc 0.3.3
itself exposes (``render_transmittance_from_*``, ``render_visibility``, ``pack_info``, ``unpack_info``,
``contract``, ``contract_inv``, ``query_grid``).  All kernels live in ``libnsr_hip.so``.
"""
from .contraction import ContractionType, contract, contract_inv  # noqa: F401
from .grid import Grid, OccupancyGrid, query_grid  # noqa: F401
from .intersection import ray_aabb_intersect  # noqa: F401
from .pack import pack_info, unpack_info  # noqa: F401
from .ray_marching import ray_marching  # noqa: F401
from .vol_rendering import (accumulate_along_rays, render_transmittance_from_alpha,  # noqa: F401
                            render_transmittance_from_density, render_visibility, render_weight_from_alpha,
                            render_weight_from_density)

__version__ = "0.3.3+nsr.gfx950"
