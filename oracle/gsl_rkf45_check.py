#!/usr/bin/env python3
"""A THIRD, independent statement of the adaptive stepper behind stepHam / evolveHam -- and the
per-attempt fixtures it produces (tests/golden/gsl_rkf45_trace.json).

TEST INFRASTRUCTURE.  The oracle (oracle/hamk_oracle.c) and the device code (hamk_device.hpp,
hamk_wave.hpp) restate GSL's rkf45 stepper, standard step-size control and evolve loop from the same
recollection by the same author; a test of one against the other cannot catch an error they share.
This file shares nothing with them:

  * the right-hand side is Hamilton's equations taken from the SYMBOLIC Hamiltonian
    H(q, p) = p.K(q)^-1.p / 2 + U(q), K = J^T M J (sympy; dq = dH/dp, dp = -dH/dq, lambdified to
    plain Python floats) -- no tape, no jets, no hamEqs algebra of Hamilton.hs:375-387;
  * the Runge-Kutta-Fehlberg 4(5) pair is taken as exact rationals from the literature (Fehlberg
    1969; Hairer-Norsett-Wanner I, table II.5.1 "Fehlberg 4(5)") and CHECKED here by its order
    conditions up to order 5 (4) before use;
  * step-size control and the evolve loop are written from the GSL reference manual's description
    ("Adaptive Step-size Control": D_i = eps_abs + eps_rel (a_y |y_i| + a_dydt h |y'_i|); the step
    is redone with h S (E/D)^(-1/q) when E/D > 1.1, grown by S (E/D)^(-1/(q+1)) when E/D < 0.5;
    S = 0.9, factors limited to [1/5, 5]; "Evolution": advance towards t1 with the suggested h, the
    last step clipped to hit t1 exactly) and from the two loops of hmatrix-gsl's gsl-ode.c:
        api 1  (-DGSLODE1)  for each ti: while (t < ti) gsl_odeiv_evolve_apply(...)
        api 2  (default)    for each ti: gsl_odeiv2_driver_apply(d, &t, ti, y)
    with the one documented difference of gsl_odeiv2_evolve_apply: "Change of step size is not
    suggested in the final step, because that step can be very small compared to previous step".
    hmatrix-gsl's odeSolveV passes a_y = a_dydt = 1 (`XX'` control); Hamilton.hs:447-448 passes
    h0 = (t1 - t0)/100 and eps_abs = eps_rel = 1.49012e-08.  The controller's order is q = 5.

Run:  python oracle/gsl_rkf45_check.py        (writes the fixture; a few seconds)
The fixture is compared with the oracle's own trace (orc_evolve_ham_trace) in the CPU suite
(tests/test_gsl_independent.py) and with the GPU's sub-step counts and states in the GPU suite.
"""
from __future__ import annotations

import json
import math
import os
import sys
from fractions import Fraction as F

import sympy as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hamilton_amd import examples as E   # noqa: E402  (system DEFINITIONS only: f, U, inertias)

OUT = os.path.join(ROOT, "tests", "golden", "gsl_rkf45_trace.json")
EPS = 1.49012e-08          # Hamilton.hs:448

# ---- Fehlberg 4(5), exact ---------------------------------------------------------------------
C = [F(0), F(1, 4), F(3, 8), F(12, 13), F(1), F(1, 2)]
A = [[],
     [F(1, 4)],
     [F(3, 32), F(9, 32)],
     [F(1932, 2197), F(-7200, 2197), F(7296, 2197)],
     [F(439, 216), F(-8), F(3680, 513), F(-845, 4104)],
     [F(-8, 27), F(2), F(-3544, 2565), F(1859, 4104), F(-11, 40)]]
B5 = [F(16, 135), F(0), F(6656, 12825), F(28561, 56430), F(-9, 50), F(2, 55)]      # 5th order: the solution GSL advances with
B4 = [F(25, 216), F(0), F(1408, 2565), F(2197, 4104), F(-1, 5), F(0)]              # 4th order: only its difference to B5 is used


def check_tableau():
    """Row sums and the order conditions (all rooted trees) through order 5 for B5, order 4 for B4."""
    n = 6
    for i in range(n):
        assert sum(A[i], F(0)) == C[i], i
    a = [[A[i][j] if j < len(A[i]) else F(0) for j in range(n)] for i in range(n)]
    Ac = [sum(a[i][j] * C[j] for j in range(n)) for i in range(n)]
    Ac2 = [sum(a[i][j] * C[j] ** 2 for j in range(n)) for i in range(n)]
    Ac3 = [sum(a[i][j] * C[j] ** 3 for j in range(n)) for i in range(n)]
    AAc = [sum(a[i][j] * Ac[j] for j in range(n)) for i in range(n)]
    AAc2 = [sum(a[i][j] * Ac2[j] for j in range(n)) for i in range(n)]
    AcAc = [sum(a[i][j] * C[j] * Ac[j] for j in range(n)) for i in range(n)]
    AAAc = [sum(a[i][j] * AAc[j] for j in range(n)) for i in range(n)]
    dot = lambda b, v: sum(b[i] * v[i] for i in range(n))
    one = [F(1)] * n
    conds = {1: [(one, F(1))],
             2: [(C, F(1, 2))],
             3: [([c * c for c in C], F(1, 3)), (Ac, F(1, 6))],
             4: [([c ** 3 for c in C], F(1, 4)), ([C[i] * Ac[i] for i in range(n)], F(1, 8)), (Ac2, F(1, 12)), (AAc, F(1, 24))],
             5: [([c ** 4 for c in C], F(1, 5)), ([C[i] ** 2 * Ac[i] for i in range(n)], F(1, 10)), ([C[i] * Ac2[i] for i in range(n)], F(1, 15)),
                 ([C[i] * AAc[i] for i in range(n)], F(1, 30)), ([Ac[i] ** 2 for i in range(n)], F(1, 20)), (Ac3, F(1, 20)),
                 (AcAc, F(1, 40)), (AAc2, F(1, 60)), (AAAc, F(1, 120))]}
    for order, cs in conds.items():
        for v, rhs in cs:
            assert dot(B5, v) == rhs, ("B5", order)
            if order <= 4:
                assert dot(B4, v) == rhs, ("B4", order)
    assert any(dot(B4, v) != rhs for v, rhs in conds[5])          # ... and B4 really is only 4th order


A_f = [[float(x) for x in row] for row in A]
B5_f = [float(x) for x in B5]
ERR_f = [float(b5 - b4) for b5, b4 in zip(B5, B4)]                 # y5 - y4 = h sum (b5 - b4) k


# ---- Hamilton's equations from the symbolic Hamiltonian -------------------------------------------
def hamilton_rhs(spec):
    n = spec.n
    q = sp.symbols(f"q0:{n}", real=True)
    p = sp.symbols(f"p0:{n}", real=True)
    ops = E._Ops(sp)
    x = [sp.sympify(e) for e in spec.coords(q, ops)]
    U = sp.sympify(spec.potential_of_q(q, ops))
    J = sp.Matrix([[sp.diff(x[k], q[i]) for i in range(n)] for k in range(spec.m)])
    K = (J.T * sp.diag(*spec.inertia) * J).applyfunc(sp.simplify)
    pv = sp.Matrix(p)
    H = (pv.T * K.inv() * pv)[0, 0] / 2 + U
    dq = [sp.diff(H, p[i]) for i in range(n)]
    dp = [-sp.diff(H, q[i]) for i in range(n)]
    f = sp.lambdify(list(q) + list(p), dq + dp, modules="math", cse=True)
    return lambda y: list(f(*y))


# ---- one attempt of the stepper -------------------------------------------------------------------
def rkf45_attempt(rhs, y, f0, h):
    k = [f0]
    for s in range(1, 6):
        ys = [y[i] + h * sum(A_f[s][j] * k[j][i] for j in range(s)) for i in range(len(y))]
        k.append(rhs(ys))
    ynew = [y[i] + h * sum(B5_f[j] * k[j][i] for j in range(6)) for i in range(len(y))]
    yerr = [h * sum(ERR_f[j] * k[j][i] for j in range(6)) for i in range(len(y))]
    return ynew, yerr


def control(y, yerr, dydt, h, eps_abs, eps_rel):
    """GSL standard control, a_y = a_dydt = 1, method order 5: returns (verdict, suggested h, ratio)."""
    ratio = sys.float_info.min
    for yi, ei, fi in zip(y, yerr, dydt):
        D = eps_abs + eps_rel * (abs(yi) + abs(h * fi))
        ratio = max(ratio, abs(ei) / abs(D))
    if ratio > 1.1:
        return "decrease", h * max(0.2, 0.9 * ratio ** (-1.0 / 5.0)), ratio
    if ratio < 0.5:
        return "increase", h * min(5.0, max(1.0, 0.9 * ratio ** (-1.0 / 6.0))), ratio
    return "keep", h, ratio


def evolve(rhs, y0, ts, api, h0=None, eps_abs=EPS, eps_rel=EPS, max_attempts=100000):
    """Rows at each ts[i] (row 0 = y0), the attempt trace [(t reached, h tried, +1 / 0 / -1)], the
    smallest distance of any control ratio from its two thresholds, and the failure code."""
    y = list(y0)
    t = ts[0]
    h = h0 if (h0 is not None and h0 > 0) else (ts[1] - ts[0]) / 100.0        # Hamilton.hs:447
    sign = 1.0 if (api == 1 or h > 0.0) else -1.0
    rows, trace, margin, fail = [list(y)], [], math.inf, 0
    dydt = rhs(y)
    for ti in ts[1:]:
        if api == 2 and not fail and sign * (ti - t) < 0.0:
            fail = 2                                                           # GSL_EINVAL: wrong side of t
        while not fail and sign * (ti - t) > 0.0:
            # ---- one evolve_apply: attempts until one is accepted --------------------------------
            t0, h_try = t, h
            while True:
                assert len(trace) < max_attempts
                final = (ti - t0 >= 0.0 and h_try > ti - t0) or (ti - t0 < 0.0 and h_try < ti - t0)
                if final:
                    h_try = ti - t0
                ynew, yerr = rkf45_attempt(rhs, y, dydt, h_try)
                fnew = rhs(ynew)
                t_reached = ti if final else t0 + h_try
                verdict, h_next, ratio = control(ynew, yerr, fnew, h_try, eps_abs, eps_rel)
                margin = min(margin, abs(ratio - 1.1), abs(ratio - 0.5))
                if verdict == "decrease":
                    if abs(h_next) < abs(h_try) and t_reached + h_next != t_reached:
                        trace.append((t_reached, h_try, 0))
                        h_try = h_next
                        continue                                               # redo the step from (t0, y)
                    if api == 2:                                               # cannot shrink: GSL_FAILURE, state stays advanced
                        trace.append((t_reached, h_try, -1))
                        y, dydt, t, h, fail = ynew, fnew, t_reached, h_next, 1
                        break
                    h_next = h_try                                             # api 1: keep the step size, accept
                trace.append((t_reached, h_try, 1))
                y, dydt, t = ynew, fnew, t_reached
                if api == 1 or not final:
                    h = h_next                                                 # api 2 keeps the old suggestion on a final step
                break
        rows.append(list(y))
    return rows, trace, margin, fail


# ---- fixtures -------------------------------------------------------------------------------------
CASES = {
    "doublePendulum": dict(starts=[("seInit", None), ("swinging", ([1.1, -0.7], [0.4, -0.9]))], ts=[0.0, 0.1, 0.25, 0.6, 0.61]),
    "twoBody": dict(starts=[("seInit", None), ("eccentric", ([1.6, 0.3], [-0.03, 0.45]))], ts=[0.0, 0.4, 1.0, 2.5]),
    "spring": dict(starts=[("seInit", None), ("stretched", ([0.4, 0.15, -0.3], [0.2, -0.4, 0.3]))], ts=[0.0, 0.1, 0.25, 0.6]),
}


def momenta_of(spec, q, qd):
    """p = J^T M J qd, symbolically (initial Config -> Phase; Hamilton.hs:279-284)."""
    n = spec.n
    qs = sp.symbols(f"q0:{n}", real=True)
    x = [sp.sympify(e) for e in spec.coords(qs, E._Ops(sp))]
    J = sp.Matrix([[sp.diff(x[k], qs[i]) for i in range(n)] for k in range(spec.m)])
    K = J.T * sp.diag(*spec.inertia) * J
    Kn = K.subs(dict(zip(qs, q)))
    return [float(v) for v in (Kn * sp.Matrix(qd)).evalf(30)]


def main():
    check_tableau()
    out = {"generator": "oracle/gsl_rkf45_check.py", "eps": EPS, "cases": []}
    for name, cfg in CASES.items():
        spec = E.get(name)
        rhs = hamilton_rhs(spec)
        for label, start in cfg["starts"]:
            q, qd = (list(spec.q0), list(spec.qd0)) if start is None else start
            p = momenta_of(spec, q, qd)
            for api in (1, 2):
                rows, trace, margin, fail = evolve(rhs, q + p, cfg["ts"], api)
                assert fail == 0 and margin > 1e-6, (name, label, api, margin)      # no decision hangs on the last bits
                out["cases"].append({"system": name, "start": label, "api": api, "q0": q, "p0": p, "ts": cfg["ts"],
                                     "rows": rows, "attempts": len(trace), "accepted": sum(1 for a in trace if a[2] == 1),
                                     "trace": [[t, h, a] for t, h, a in trace], "min_threshold_margin": margin})
                print(name, label, "api", api, "attempts", len(trace), "margin %.2e" % margin)
            # backward in time: only gsl_odeiv2 integrates a decreasing grid
            tsb = [-t for t in cfg["ts"]]
            rows, trace, margin, fail = evolve(rhs, q + p, tsb, 2)
            assert fail == 0 and margin > 1e-6
            out["cases"].append({"system": name, "start": label, "api": 2, "q0": q, "p0": p, "ts": tsb, "rows": rows,
                                 "attempts": len(trace), "accepted": sum(1 for a in trace if a[2] == 1),
                                 "trace": [[t, h, a] for t, h, a in trace], "min_threshold_margin": margin})
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
