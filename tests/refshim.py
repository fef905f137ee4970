"""Import the reference's ``models/`` UNCHANGED from /root/reference on a GPU-less box.  TEST INFRASTRUCTURE.

Only usable in the build container (``/root/reference`` does not exist on the GPU box): it is used by
``tests/gen_golden.py`` to mint the fixtures under ``tests/golden/`` and by the ``not gpu`` glue tests,
which skip when the reference tree is absent.

It provides throw-away stand-ins for the packages the reference imports but this image lacks
(``pytorch_lightning``, ``omegaconf``, ``torch_efficient_distloss``, ``cv2``, ``imageio``), binds
``tinycudann`` / ``nerfacc`` to a chosen backend (the CPU oracle here), and neutralises the two
CUDA-only idioms in the reference glue (``with torch.cuda.device(get_rank())`` and ``device=get_rank()``,
``models/network_utils.py:46,53,89,180,208``).
"""
import contextlib
import os
import re
import sys
import types

import torch
import yaml

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "models"))


# ------------------------------------------------------------------------------------------------
# mini OmegaConf
# ------------------------------------------------------------------------------------------------
class DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return _wrap(_unwrap(self))


def _wrap(o):
    if isinstance(o, dict):
        return DictConfig({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return [_wrap(v) for v in o]
    return o


def _unwrap(o):
    if isinstance(o, dict):
        return {k: _unwrap(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_unwrap(v) for v in o]
    return o


class OmegaConf:
    _resolvers = {}

    @classmethod
    def register_new_resolver(cls, name, fn, **kw):
        cls._resolvers[name] = fn

    @staticmethod
    def load(f):
        with open(f) as fp:
            return _wrap(yaml.safe_load(fp))

    @staticmethod
    def create(d=None):
        return _wrap(d or {})

    @staticmethod
    def from_cli(args):
        out = {}
        for a in args:
            k, v = a.split("=", 1)
            cur = out
            ks = k.split(".")
            for kk in ks[:-1]:
                cur = cur.setdefault(kk, {})
            cur[ks[-1]] = yaml.safe_load(v)
        return _wrap(out)

    @staticmethod
    def merge(*confs):
        def m(a, b):
            for k, v in b.items():
                if isinstance(v, dict) and isinstance(a.get(k), dict):
                    m(a[k], v)
                else:
                    a[k] = v
            return a
        out = {}
        for c in confs:
            m(out, _unwrap(c))
        return _wrap(out)

    @classmethod
    def resolve(cls, conf):
        root = conf

        def lookup(path):
            cur = root
            for p in path.split("."):
                cur = cur[p]
            return res(cur)

        def split_args(s):
            args, depth, cur = [], 0, ""
            for ch in s:
                if ch == "," and depth == 0:
                    args.append(cur)
                    cur = ""
                    continue
                depth += ch == "{"
                depth -= ch == "}"
                cur += ch
            args.append(cur)
            return args

        def res(v):
            if not isinstance(v, str) or "${" not in v:
                return v
            m = re.fullmatch(r"\$\{(.*)\}", v.strip())
            if m and _balanced(m.group(1)):
                inner = m.group(1)
                if ":" in inner and re.match(r"^[A-Za-z_]+:", inner):
                    name, rest = inner.split(":", 1)
                    args = [yaml.safe_load(str(res(a.strip()))) if isinstance(res(a.strip()), str) else res(a.strip())
                            for a in split_args(rest)]
                    return cls._resolvers[name](*args)
                return lookup(inner)
            # string interpolation
            return re.sub(r"\$\{([^${}]*)\}", lambda mm: str(lookup(mm.group(1))), v)

        def walk(o):
            if isinstance(o, dict):
                for k in list(o.keys()):
                    o[k] = walk(o[k])
                return o
            if isinstance(o, list):
                return [walk(x) for x in o]
            return res(o)

        walk(conf)

    @staticmethod
    def to_container(conf, resolve=True):
        return _unwrap(conf)

    @staticmethod
    def save(config, f):
        yaml.safe_dump(_unwrap(config), f)


def _balanced(s):
    d = 0
    for ch in s:
        d += ch == "{"
        d -= ch == "}"
        if d < 0:
            return False
    return d == 0


# ------------------------------------------------------------------------------------------------
# stand-in modules
# ------------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    noop = lambda *a, **k: None  # noqa: E731

    def rank_zero_only(fn):
        return fn

    class LightningModule(torch.nn.Module):
        def log(self, *a, **k):
            pass

    pl = _mod("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=object,
              __version__="1.9.0", seed_everything=lambda s: torch.manual_seed(s))
    _mod("pytorch_lightning.utilities")
    _mod("pytorch_lightning.utilities.rank_zero", rank_zero_info=noop, rank_zero_debug=noop,
         rank_zero_warn=noop, rank_zero_only=rank_zero_only)
    pl.Callback = object
    _mod("pytorch_lightning.callbacks")
    _mod("pytorch_lightning.callbacks.progress", TQDMProgressBar=object)
    _mod("pytorch_lightning.loggers")
    _mod("pytorch_lightning.loggers.base", LightningLoggerBase=object, rank_zero_experiment=rank_zero_only)

    def _no_distloss(*a, **k):
        raise NotImplementedError("torch_efficient_distloss is out of scope (lambda_distortion=0)")

    _mod("torch_efficient_distloss", flatten_eff_distloss=_no_distloss)
    _mod("cv2")
    _mod("mcubes", marching_cubes=_no_distloss)  # export-only (models/geometry.py:42), out of scope
    _mod("imageio")
    _mod("omegaconf", OmegaConf=OmegaConf, DictConfig=DictConfig)


_SAVED = {}


def uninstall():
    """undo install(): restore sys.modules / sys.path / torch.cuda.device so later tests see the product packages"""
    if not _SAVED:
        return
    for k in [k for k in sys.modules if k.split(".")[0] in _SAVED["names"]]:
        del sys.modules[k]
    sys.modules.update(_SAVED["modules"])
    if REF in sys.path:
        sys.path.remove(REF)
    torch.cuda.device, torch.cuda.empty_cache = _SAVED["cuda_device"], _SAVED["empty_cache"]
    sys.dont_write_bytecode = _SAVED["dwb"]
    _SAVED.clear()


def install(tcnn_module, nerfacc_module, device="cpu"):
    """Make ``import models`` resolve to the reference with the given tcnn / nerfacc backends."""
    assert available(), "/root/reference is not present on this machine"
    names = {"tinycudann", "nerfacc", "models", "systems", "utils", "datasets", "pytorch_lightning", "omegaconf",
             "torch_efficient_distloss", "cv2", "imageio", "mcubes"}
    if not _SAVED:
        _SAVED.update(names=names, modules={k: v for k, v in sys.modules.items() if k.split(".")[0] in names},
                      cuda_device=torch.cuda.device, empty_cache=torch.cuda.empty_cache, dwb=sys.dont_write_bytecode)
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.modules["tinycudann"] = tcnn_module
    sys.modules["nerfacc"] = nerfacc_module
    sys.modules["nerfacc.intersection"] = nerfacc_module.intersection
    for name in ("models", "systems", "utils", "datasets"):
        for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
            del sys.modules[k]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import utils.misc as misc
    misc.get_rank = lambda: device
    if device == "cpu":
        torch.cuda.device = lambda *_a, **_k: contextlib.nullcontext()
        torch.cuda.empty_cache = lambda: None
    import models  # noqa: F401  (fills the registry)
    return models


def load_config(name, cli=()):
    """Resolve one of the reference's YAMLs exactly as utils/misc.py:26-31 does."""
    import utils.misc as misc
    return misc.load_config(os.path.join(REF, "configs", name), cli_args=list(cli))
