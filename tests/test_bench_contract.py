"""bench.py's host-side logic that needs no GPU: the command line the driver uses, and the PMC traffic figure the `roofline`
object attaches -- taken from this round's committed passes only when they ran in the same regime."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        return importlib.import_module("bench")
    finally:
        sys.argv = argv


def test_committed_pmc_passes_cover_both_command_lines():
    bench = _bench()
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")))
    assert {"w20_s200", "w5_s20"} <= set(pmc["regimes"])
    for key, (w, s) in {"w20_s200": (20, 200), "w5_s20": (5, 20)}.items():
        ent = pmc["regimes"][key]["hashgrid_backward_params"]
        got, src = bench.pmc_traffic("hashgrid_backward_params", ent["samples_per_launch"], w, s)
        assert got == ent["bytes_per_launch"] and "r04_pmc_traffic.json" in src
        # measured traffic can only exceed what the operation must move (140 B / sample + 26 B / table parameter)
        algorithmic = 140.0 * ent["samples_per_launch"] + 26.0 * 12599920
        assert algorithmic < got < 2.0 * algorithmic
    # a run in another regime (twice the samples per launch) gets no figure rather than a wrong one
    got, why = bench.pmc_traffic("hashgrid_backward_params", 2.2e5, 5, 20)
    assert got is None and "not comparable" in why


def test_default_command_line_times_the_operating_point():
    """`python bench.py` and the driver's `--gpus 1 --steps 20 --warmup 5` both run 300 untimed set-up steps first (the 8,192-ray
    operating point BASELINE.json quotes the metric on) behind a disclosed burn-in"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--setup-steps", type=int, default=300' in src and '"--burn-in-steps", type=int, default=1500' in src
    assert '"--gpus", type=int, default=1' in src and "burn_in_steps_of_a_throwaway_model" in src
