"""The fused registry entries (nsr.models.FusedNeRFModel / FusedNeuSModel) under the protocol the reference actually runs its
models with: Lightning ``precision: 16`` = ``torch.autocast(float16)`` + ``GradScaler(init_scale=65536)``
(configs/nerf-blender.yaml:103, configs/neus-blender.yaml) and ``DistributedDataParallel(find_unused_parameters=False)``
(launch.py:93-107) -- two ranks over a gloo rendezvous on the box's one GPU, tests/entry_protocol_worker.py.  The entries'
autograd.Functions hand their parameter gradients to autograd (which fires DDP's reducer hooks); after the all-reduce and the
scaler's unscale ``.grad`` must equal the mean of the two ranks' fp32 single-process gradients, the replicas must stay
identical through the optimizer step, and a scale that overflows must skip the step and halve."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_fused_entries_under_autocast_gradscaler_and_ddp():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "entry_protocol_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("ENTRY_PROTOCOL_REPORT ")]
    assert line, p.stdout[-2000:]
    rep = json.loads(line[-1][len("ENTRY_PROTOCOL_REPORT "):])
    for kind in ("nerf", "neus"):
        r = rep[kind]
        assert r["samples"] > 5000, r
        assert not r["missing_grads"], r                      # every parameter the fp32 run trains got a gradient through DDP
        table = "geometry.encoding_with_network.params" if kind == "nerf" else "geometry.encoding.encoding.params"
        assert r["grad_norms"][table] > 0 and table in r["errs"], r   # ... the hash table among them
        # mean of the ranks' fp32 gradients, after all-reduce + unscale (65536 = 2^16: the scaling itself is exact; what is
        # left is the fp16 rounding autocast applies to the system's own loss arithmetic)
        assert r["max_rel_err"] < 5e-3, (kind, r["worst"], r["errs"])
        assert r["tensors_moved"] == r["tensors"] and r["finite"], r
        assert r["replica_mismatch"] == 0.0, r                # the ranks applied the same update
        assert r["overflow_step_skipped"] and r["scale_after_overflow"] == 2.0 ** 126, r
