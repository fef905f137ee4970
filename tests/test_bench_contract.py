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
    name = next(f for f in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json")
                if os.path.exists(os.path.join(ROOT, "profiles", f)))
    pmc = json.load(open(os.path.join(ROOT, "profiles", name)))  # (bench.py takes the newest round's passes)
    assert {"w20_s200", "w5_s20"} <= set(pmc["regimes"])
    for key, (w, s) in {"w20_s200": (20, 200), "w5_s20": (5, 20)}.items():
        ent = pmc["regimes"][key]["hashgrid_backward_params"]
        got, src = bench.pmc_traffic("hashgrid_backward_params", ent["samples_per_launch"], w, s)
        assert got == ent["bytes_per_launch"] and name in src
        if name.startswith("r05"):  # the read side is calibrated by a stand-alone sweep of the same collection (VERDICT r4 #2a)
            cal = pmc["regimes"][key]["_fetch_calibration"]
            assert cal["kernel"] == "k_adamw" and abs(cal["measured_over_expected"] - 0.5) < 0.02
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


def test_round5_blocks_are_on_the_line():
    """the driver's line carries the whole 20,000-step run (seconds, samples/s, PSNR) and the late regime, and the
    same-process A/B of the step's forms; the traffic figure comes from this round's PMC passes when they exist"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"whole_run": whole', '"late_regime": late', '"step_forms_ab": forms_ab', '"--whole-run-steps", type=int, default=20000',
                '"test_psnr"', '"at_step": late_at'):
        assert key in src, key
    assert src.index('"r06_pmc_traffic.json"') < src.index('"r05_pmc_traffic.json"') < src.index('"r04_pmc_traffic.json"')
    cal = os.path.join(ROOT, "profiles", "r05_fetch_calibration.json")
    if os.path.exists(cal):  # the stand-alone k_adamw sweep: FETCH_SIZE under-reports by 2 on gfx950, WRITE_SIZE is exact
        c = json.load(open(cal))
        assert abs(c["fetch_measured_over_expected"] - 0.5) < 0.02 and abs(c["write_measured_over_expected"] - 1.0) < 0.03


def test_step_forms_are_named_in_one_place():
    """the forms bench.py's A/B switches between are the trainer module's (no second copy of the key tuples)"""
    src = open(os.path.join(ROOT, "instant-nsr-pl_amd", "nsr", "trainer.py")).read()
    assert "CURRENT_FORMS = dict(keys={0: 1, 2: 1, 5: 1}" in src and "ROUND4_FORMS = dict(keys={0: 0, 2: 0, 5: 0}" in src
    assert "from nsr.trainer import ROUND4_FORMS, CURRENT_FORMS, set_step_forms" in open(os.path.join(ROOT, "bench.py")).read()
