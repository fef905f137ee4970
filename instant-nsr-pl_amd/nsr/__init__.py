"""Host side of the MI355X hot path above the drop-in ``tinycudann`` / ``nerfacc`` packages: fused step runners
(``nsr.fused``: NeRF; ``nsr.fused_neus``: NeuS / neuralangelo), the one-process-per-GPU trainer (``nsr.trainer``,
``nsr.parallel``), a parameter holder with the reference's state-dict keys (``nsr.state``), the reference YAMLs'
model sections as dicts (``nsr.configs``) and a procedural dataset (``nsr.scene``).

The reference's own ``models/`` run unchanged on the two drop-in packages; the runners here take either
those model objects or an ``nsr.state.HotPathState``.  A line-by-line restatement of the reference's glue exists only
as test infrastructure (``tests/refmirror``: ``/root/reference`` cannot travel to the GPU box).
"""
from . import configs  # noqa: F401
from .state import HotPathState, build  # noqa: F401
