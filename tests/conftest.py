import os
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle (torch) is the slow side of most GPU tests: on the GPU box's 2 x 64-core host its op sequence peaks at ~16 threads and is
    # 2.5 - 9x slower with all 256 logical cores (tests/tools/cpu_threads_probe.py; bench.py's cpu_baseline uses 16 for the same reason)
    try:
        import torch
        if (os.cpu_count() or 1) > 32:
            torch.set_num_threads(16)
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load


# ---- loose fall-back tolerances report what they measure ------------------------------------------------------------
# A comparison that cannot be held to 1e-4 for a stated reason (a discrete assignment flipped, a near-tie moved a neighbour, 16 of 400
# optimiser steps) used to assert a flat 1e-3: a regression from 2e-5 to 9e-4 passed silently.  `calibrated(name, value, flat)` asserts
# value <= 3 x the calibration value committed in tests/golden/calibration.json (a measurement on an MI355X of the round that introduced
# it; never above the flat bound) and appends the measured value to gpurun_out/measured_tolerances.jsonl so that every GPU run leaves a
# record.  A name without a committed calibration value is held to the flat bound and reported, nothing else.
CALIBRATION = os.path.join(GOLDEN, "calibration.json")


def calibrated(name, value, flat):
    import json
    value = float(value)
    try:
        with open(CALIBRATION) as f:
            cal = json.load(f).get(name)
    except (OSError, ValueError):
        cal = None
    bound = flat if cal is None else min(flat, max(3.0 * float(cal), 1e-7))
    try:
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "measured_tolerances.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, "measured": value, "calibration": cal, "bound": bound, "flat": flat}) + "\n")
    except OSError:
        pass
    print(f"[calibrated] {name}: measured {value:.3e}, calibration {cal}, bound {bound:.3e}")
    assert value <= bound, f"{name}: measured {value:.3e} exceeds {bound:.3e} (3 x the committed calibration {cal}; flat bound {flat:.0e})"
    return value
