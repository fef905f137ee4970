"""The multi-GPU exchange of both trainers on REAL RCCL collectives (backend "nccl"), one rank: the gloo tests cover the
logic at world 2 / 4, this one the nccl branches of nsr/parallel.py (reduce_scatter_tensor in bf16, all_gather_into_tensor
into a view of the fp16 shadow, the communication stream behind the step's HIP events) that RCCL's one-rank-per-device rule
keeps the two-rank tests away from."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_trainers_run_the_sharded_exchange_on_an_nccl_group():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(HERE, "nccl_single_rank_worker.py")], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rep = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{"backend"')][-1])  # (RCCL prints its banner too)
    assert rep["backend"] == "nccl"
    assert rep["nerf-blender-async"]["groups"] == 2 and rep["nerf-blender-async"]["ranges_timed"] >= 2 * 5
    for name in ("nerf-blender-async", "nerf-blender", "neus-dtu", "neuralangelo"):
        r = rep[name]
        assert r["finite"] and r["ranges_timed"] > 0, (name, r)
        # same seeds, same batches: the sharded run differs from the one-GPU optimizer path by the bf16 rounding of the
        # gradient only.  AdamW (eps 1e-15) turns a gradient that is pure rounding noise into a full +-lr step, so a few
        # entries differ by 2 lr per step; in norm the tables agree to a few per cent (measured: 0.3-4.7 %)
        assert r["rel_l2_diff"] < 0.2, (name, r)
        # fp32 gradients on the wire: only the summation order differs from the one-GPU path (<= 0.05; the bf16
        # figure above is reported beside it)
        assert r["rel_l2_diff_fp32_transport"] < 0.05, (name, r)
        assert 0.0 < r["reduce_scatter_ms"] < 1000.0, (name, r)  # (the first collective of a process may carry RCCL's set-up)
