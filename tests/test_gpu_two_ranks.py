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
    assert rep["nerf-blender-async"]["groups"] == [[11, 16], [0, 11]]
    for name in ("nerf-blender", "nerf-blender-async", "neus-dtu"):
        r = rep[name]
        assert r["finite"], r
        assert r["replica_mismatch"] == 0.0, r          # bit-identical replicas
        assert r["tensors_moved"] == r["tensors"], r     # every tensor trained (incl. the background's)
        assert r["samples_rank0"] != r["samples_rank1"], r  # the ranks drew different ray batches


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` from a bare interpreter (how the round-end driver calls it; the reference's launch.py:93-107
    spawns its DDP ranks itself as well): the script re-launches under torch.distributed.run, and on a box with fewer GPUs
    than ranks the ranks share device 0 over gloo -- the whole multi-rank bench path (ray-sharded fused step, bf16 table
    exchange in level groups, C3 / C4 / C5 at world 2) runs and prints one JSON line with `gradient_exchange`"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 2 and res["value"] > 0
    ex = res["gradient_exchange"]
    assert ex is not None and ex["ranges_per_step"] == 2 and ex["level_groups"] == [[11, 16], [0, 11]]
    assert ex["reduce_scatter_ms"] > 0 and ex["all_gather_ms"] > 0
    for name in ("neus-blender", "neus-dtu", "neuralangelo"):
        assert res["other_workloads"][name]["n_gpus"] == 2 and res["other_workloads"][name]["samples_per_sec"] > 0
