"""``nerfacc.intersection`` stand-in (reference import site: models/neus.py:12).  TEST INFRASTRUCTURE ONLY."""
from . import ray_aabb_intersect  # noqa: F401
