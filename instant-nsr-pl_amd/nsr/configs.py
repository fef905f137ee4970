"""``model:`` sections of the reference's YAMLs, resolved (``${...}`` interpolations evaluated), as plain dicts: the
values define the hot-path shapes.  Minted by tools/gen_configs.py from ``/root/reference/configs/*.yaml`` into
``configs.json`` and re-checked against the YAMLs by tests/test_configs.py.

  nerf-blender  = configs/nerf-blender.yaml            (BASELINE.json configs[1], "C2")
  neus-blender  = configs/neus-blender.yaml            (configs[2], "C3")
  neus-dtu      = configs/neus-dtu.yaml                (configs[3], "C4": learned NeRF++ background)
  neuralangelo  = configs/neuralangelo-dtu-wmask.yaml  (configs[4], "C5": progressive levels + finite differences)
"""
import copy
import json
import os

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs.json")) as _f:
    _ALL = json.load(_f)


def get(name):
    return copy.deepcopy(_ALL[name])


def names():
    return sorted(_ALL)
