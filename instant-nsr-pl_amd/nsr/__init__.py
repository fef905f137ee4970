"""Host-side mirror of the reference's hot-path glue (``models/{geometry,texture,nerf,neus}.py``) on top of the
drop-in ``tinycudann`` / ``nerfacc`` packages and the fused glue kernels of libnsr_hip.so.

It exists because ``/root/reference`` cannot travel to the GPU box: the ``-m gpu`` tests, ``smoke()`` and
``bench.py`` drive the kernels through this mirror, while the reference's own ``models/`` run unchanged on the
same two packages when a user installs them.  Module / parameter names follow the reference so
that ``state_dict`` keys match (``geometry.encoding_with_network.params``, ``texture.network.params`` ...).
"""
from . import configs  # noqa: F401
from .fields import VarianceNetwork, VolumeDensity, VolumeRadiance, VolumeSDF  # noqa: F401
from .renderers import NeRFModel, NeuSModel  # noqa: F401
