import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# a kernel gone wrong must end: cap the adaptive stepper's sub-step budget for the whole suite
# (the default, 2^24 attempts per trajectory and call, can keep a GPU busy for minutes)
os.environ.setdefault("HAMK_MAX_SUBSTEPS", "20000")

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE_SYSTEMS = ["pendulum", "doublePendulum", "room", "twoBody", "spring", "bezier"]
ALL_GOLDEN_SYSTEMS = REFERENCE_SYSTEMS + ["threeBodyPolar", "chain4", "opcodeZoo"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_golden(name):
    with open(os.path.join(GOLDEN, f"{name}.json")) as fh:
        return json.load(fh)


def fvec(xs):
    return np.array([float(x) for x in xs], dtype=np.float64)


@pytest.fixture(scope="session")
def oracle_lib():
    """Builds oracle/libhamk_oracle.so if needed (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hamk_lib():
    """Builds hamilton_amd/libhamk.so if needed and returns the ctypes handle."""
    import subprocess
    lib_path = os.path.join(ROOT, "hamilton_amd", "libhamk.so")
    if not os.path.exists(lib_path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "hamilton_amd", "csrc")], stdout=subprocess.DEVNULL)
    from hamilton_amd import _abi
    return _abi.lib()
