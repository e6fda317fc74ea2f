"""CPU: the oracle's adaptive stepper against oracle/gsl_rkf45_check.py -- a third statement of GSL's
rkf45 + standard control + evolve loop (both bindings of hmatrix-gsl's gsl-ode.c) that shares neither
the right-hand side (symbolic Hamilton's equations instead of tape/jets/hamEqs algebra), nor the
tableau's source (literature rationals, verified by their order conditions), nor a line of code with
the oracle or the device library.  Compared per ATTEMPT: time reached, step tried, accepted/rejected.
The GPU suite compares the kernels with the same fixture (tests/test_gpu_configs.py)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from hamilton_amd import examples as E

FIX = json.load(open(os.path.join(GOLDEN, "gsl_rkf45_trace.json")))
IDS = [f"{c['system']}-{c['start']}-api{c['api']}-{'back' if c['ts'][-1] < 0 else 'fwd'}" for c in FIX["cases"]]


@pytest.mark.parametrize("case", FIX["cases"], ids=IDS)
def test_oracle_takes_the_independent_steps(oracle_lib, case):
    spec = E.get(case["system"])
    o = oracle_lib.OracleSystem(spec)
    o.gsl_api = case["api"]
    n = spec.n
    q, p, counts, trace = o.evolve_ham_trace(case["q0"], case["p0"][:n] if len(case["p0"]) > n else case["p0"], case["ts"])
    want = np.array(case["trace"])
    assert counts[3] == 0 and len(trace) == case["attempts"] == len(want), (len(trace), case["attempts"])
    assert counts[1] == case["accepted"] and counts[2] == case["attempts"] - case["accepted"]
    np.testing.assert_array_equal(trace[:, 2], want[:, 2])                         # the same accept / reject decisions
    np.testing.assert_allclose(trace[:, 0], want[:, 0], rtol=1e-6, atol=1e-12)     # the same times reached (t0 + h: inherits the sensitivity of h)
    np.testing.assert_allclose(trace[:, 1], want[:, 1], rtol=1e-5)                 # the same steps tried: a ratio^(-1/6) of an error estimate that is
    # mostly rounding on the smooth starts (spring from rest: measured 4e-8), so it moves with the evaluation order of the right-hand side
    rows = np.array(case["rows"])
    got = np.concatenate([q, p], axis=1)
    assert np.max(np.abs(got - rows) / np.maximum(1.0, np.abs(rows))) < 1e-9


def test_fixture_generator_still_agrees_with_its_fixture():
    """The committed fixture is what oracle/gsl_rkf45_check.py produces (one case re-run; the
    tableau's order conditions are re-verified on the way)."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("gsl_rkf45_check", os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "gsl_rkf45_check.py"))
    g = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(g)
    g.check_tableau()
    case = next(c for c in FIX["cases"] if c["system"] == "doublePendulum" and c["start"] == "swinging" and c["api"] == 2 and c["ts"][-1] > 0)
    rhs = g.hamilton_rhs(E.get("doublePendulum"))
    rows, trace, margin, fail = g.evolve(rhs, case["q0"] + case["p0"], case["ts"], 2)
    assert fail == 0 and len(trace) == case["attempts"]
    np.testing.assert_allclose(np.array(rows), np.array(case["rows"]), rtol=1e-12, atol=1e-14)


def test_the_two_bindings_differ_where_they_should():
    """Same steps up to the first output time (one interval), different ones afterwards; the states
    agree to the controller's tolerance."""
    by = {(c["system"], c["start"], c["api"]): c for c in FIX["cases"] if c["ts"][-1] > 0}
    differing = 0
    for (name, start, api), c1 in by.items():
        if api != 1:
            continue
        c2 = by[(name, start, 2)]
        t1 = c1["ts"][1]
        first1 = [a for a in c1["trace"] if a[0] <= t1]
        first2 = [a for a in c2["trace"] if a[0] <= t1]
        assert first1 == first2
        np.testing.assert_array_equal(np.array(c1["rows"][1]), np.array(c2["rows"][1]))
        differing += c1["trace"] != c2["trace"]
        assert np.max(np.abs(np.array(c1["rows"]) - np.array(c2["rows"]))) < 1e-5
    assert differing >= 4
