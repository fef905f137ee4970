"""Multi-rank code path of the product trainers on the GPU box: two ranks (gloo rendezvous on 127.0.0.1, both on the one
GPU of the box) run nsr.trainer.Trainer (C2) and nsr.fused_neus.NeuSTrainer (C4: NeuS + NeRF++ background) for a few
steps -- see tests/two_rank_worker.py.  After every step all ranks must hold identical parameters (the replicas saw the
same reduced gradients), while their ray batches differ."""
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


def test_two_ranks_stay_replicas_through_the_sharded_exchange():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "two_rank_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("TWO_RANK_REPORT ")]
    assert line, p.stdout[-2000:]
    rep = json.loads(line[-1][len("TWO_RANK_REPORT "):])
    for name in ("nerf-blender", "neus-dtu"):
        r = rep[name]
        assert r["finite"], r
        assert r["replica_mismatch"] == 0.0, r          # bit-identical replicas
        assert r["tensors_moved"] == r["tensors"], r     # every tensor trained (incl. the background's)
        assert r["samples_rank0"] != r["samples_rank1"], r  # the ranks drew different ray batches
