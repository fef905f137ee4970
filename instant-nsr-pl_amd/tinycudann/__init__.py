"""Drop-in ``tinycudann`` for bennyguo/instant-nsr-pl on AMD MI355X (gfx950).

Same import surface the reference uses (``models/network_utils.py:6,47,90,181,209``,
``models/utils.py:10,119``); the arithmetic runs in hand-written HIP kernels (``libnsr_hip.so``).
Put ``instant-nsr-pl_amd/`` on ``PYTHONPATH`` ahead of site-packages and the reference's
``import tinycudann as tcnn`` resolves here -- see INTEGRATION.md.
"""
from .modules import (Encoding, Module, Network, NetworkWithInputEncoding,  # noqa: F401
                      batch_size_granularity, free_temporary_memory)

__version__ = "1.7+nsr.gfx950"
