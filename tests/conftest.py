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
# the HAMK_* environment overrides (which mapping, which body, which sincos ...) are read by libhamk.so only in a process that
# asks for them: the test suites do; a product host does not
os.environ["HAMK_TEST_OVERRIDES"] = "1"

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE_SYSTEMS = ["pendulum", "doublePendulum", "room", "twoBody", "spring", "bezier"]
ALL_GOLDEN_SYSTEMS = REFERENCE_SYSTEMS + ["threeBodyPolar", "chain4", "opcodeZoo"]
# BASELINE config 5 (round 4): fixtures from the chains' closed-form mechanics (oracle/gen_golden.py evaluate_chain_point);
# kept apart from ALL_GOLDEN_SYSTEMS because these sizes run on other kernels (quad / wave) and carry no `jac` entry
CHAIN_GOLDEN_SYSTEMS = ["chain8", "chain16", "chain32"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_collection_finish(session):
    """A GPU run on a box whose code-object cache is cold (the in-tree `.hamk_cache/` normally travels with the
    snapshot; a fresh clone has none) would compile ~90 modules one after the other inside the tests -- minutes of
    one host core while the GPU idles.  hiprtc needs no GPU and the host has many cores: compile them first, in
    parallel.  Nothing here touches results; a failure only means the tests compile on demand as before."""
    if os.environ.get("HAMK_CACHE_DIR") or os.environ.get("HAMK_TEST_NO_PREWARM") or session.config.option.collectonly:
        return
    if not any(item.get_closest_marker("gpu") for item in session.items):
        return
    cache = os.path.join(ROOT, ".hamk_cache")
    try:
        if os.path.isdir(cache) and len(os.listdir(cache)) >= 150:
            return
        import subprocess
        lib_path = os.path.join(ROOT, "hamilton_amd", "libhamk.so")
        if not os.path.exists(lib_path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "hamilton_amd", "csrc")], stdout=subprocess.DEVNULL)
        jobs = str(max(1, min(48, (os.cpu_count() or 2) - 1)))
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "warm_cache.py"), "-j", jobs, "--tests-only"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    except Exception as e:                                  # noqa: BLE001 -- best effort by design
        print(f"[conftest] pre-compilation skipped: {e!r}")


# round 6: the reference's own systems from mechanics written out by hand (oracle/gen_golden_byhand.py -> byhand.json): blocks that share
# nothing with hamilton_amd/examples.py, whose definitions every other fixture, the oracle's tapes and the GPU's tapes all come from
BYHAND_SYSTEMS = ["pendulum", "doublePendulum", "doublePendulumReadme", "twoBody", "room", "spring"]


def load_golden(name):
    """`<name>` -> tests/golden/<name>.json; `byhand:<name>` -> that block of tests/golden/byhand.json (same point layout, no `jac`;
    K of these systems is at worst 3 x 3 and well conditioned on their boxes: cond_hint 1)."""
    if name.startswith("byhand:"):
        with open(os.path.join(GOLDEN, "byhand.json")) as fh:
            blk = json.load(fh)["blocks"][name[7:]]
        for pt in blk["points"]:
            pt.setdefault("cond_hint", "1")
        return blk
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
