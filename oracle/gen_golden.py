#!/usr/bin/env python3
"""Generate tests/golden/*.json: high-precision, independently derived vectors.

TEST INFRASTRUCTURE.  The reference has no golden vectors, known-answer tests or
fixtures for this path (/root/reference/test/Spec.hs:1-2 is a stub) and cannot
be executed here (no ghc/cabal/GSL), so these fixtures are DERIVED, not
recorded: parity stays "unpinned" in the sense of the task statement.  What they
provide is an arithmetic-independent check of both the C oracle and the HIP
path:

  * sympy differentiates the restated example systems (hamilton_amd/examples.py,
    following /root/reference/app/Examples.hs) symbolically: J, dJ/dq_i, grad U;
  * mpmath (50 digits) evaluates the reference's formulas literally
    (Hamilton.hs:262-269 momenta, :316-324 velocities, :341-361 energies,
    :370-387 hamEqs with explicit inverse);
  * the result is cross-checked IN THIS SCRIPT against (dH/dp, -dH/dq) obtained
    by high-precision numerical differentiation of H(q,p) = p.K^-1.p/2 + U --
    i.e. against Hamilton's equations themselves, not the reference's algebra;
  * mpmath.odefun (Taylor series, 40 digits) integrates the ODE for trajectory
    truth at a few times (truncation-level checks of RK4 / RKF45).

Run:  python oracle/gen_golden.py      (takes ~1-2 min; output is committed)
"""
from __future__ import annotations

import json
import os
import sys

import mpmath as mp
import numpy as np
import sympy as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamilton_amd import examples as E   # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
DIGITS = 50
NPOINTS = 12


def exactify(expr):
    """Replace sympy Floats (fp64 constants of the definition) by exact rationals."""
    reps = {f: sp.Rational(*float(f).as_integer_ratio()) for f in expr.atoms(sp.Float)}
    return expr.xreplace(reps)


def fmt(x) -> str:
    return mp.nstr(mp.mpf(x), 30, min_fixed=0, max_fixed=0)


# Every Python float that meets a sympy expression becomes the exact rational it denotes.  Left to its default, sympy turns the float
# into a 53-bit Float and does arithmetic on it on the spot -- `beta * (r + 1.5)` is distributed with a ROUNDED beta * 1.5,
# exp(-21.97 (y + 1)) becomes 2.868e-10 * exp(-21.97 y) -- and the fixture would carry fp64 roundings that exactify() cannot undo
# (found in round 6 against the hand-written fixtures of gen_golden_byhand.py: up to 1e-15 relative in the wall forces of room / spring / bezier).
from sympy.core.sympify import converter as _converter   # noqa: E402
import math as _math                                      # noqa: E402
_converter[float] = lambda f: sp.Rational(*f.as_integer_ratio()) if _math.isfinite(f) else sp.Float(f)


def symbolic(spec: E.SystemSpec):
    q = sp.symbols(f"q0:{spec.n}", real=True)
    ops = E._Ops(sp)
    x = [exactify(sp.sympify(e)) for e in spec.coords(q, ops)]
    U = exactify(sp.sympify(spec.potential_of_q(q, ops)))
    J = [[sp.diff(x[k], q[i]) for i in range(spec.n)] for k in range(spec.m)]
    dJ = [[[sp.diff(J[k][j], q[i]) for j in range(spec.n)] for k in range(spec.m)] for i in range(spec.n)]
    gU = [sp.diff(U, q[i]) for i in range(spec.n)]
    lam = lambda e: sp.lambdify(q, e, modules="mpmath")
    return dict(q=q, x=lam(x), U=lam(U), J=lam(J), dJ=lam(dJ), gU=lam(gU))


def mat(rows):
    return mp.matrix([[mp.mpf(v) for v in r] for r in rows])


def evaluate_point(spec, S, qv, qdv):
    """Everything the public API can return at one Config, via the reference's formulas."""
    n, m = spec.n, spec.m
    qv = [mp.mpf(v) for v in qv]
    qdv = mp.matrix([mp.mpf(v) for v in qdv])
    M = mp.diag([mp.mpf(v) for v in spec.inertia])
    x = S["x"](*qv)
    J = mat(S["J"](*qv))
    dJ = [mat(r) for r in S["dJ"](*qv)]
    gU = mp.matrix(S["gU"](*qv))
    U = mp.mpf(S["U"](*qv))
    # momenta: tr j #> diag m #> j #> qd                       Hamilton.hs:267
    p = J.T * (M * (J * qdv))
    # velocities: inv (tr j <> m <> j) #> p                     Hamilton.hs:321-324
    K = J.T * M * J
    Ki = K ** -1
    vel = Ki * p
    keC = (qdv.T * p)[0] / 2                                   # :288-296
    keP = (vel.T * p)[0] / 2                                   # :341-349
    # hamEqs                                                     :375-387
    dHdp = Ki * p
    dTdq = [-(p.T * (Ki * (J.T * (M * (dJ[i] * (Ki * p))))))[0] for i in range(n)]
    dHdq = [dTdq[i] + gU[i] for i in range(n)]
    dq = [dHdp[i] for i in range(n)]
    dp = [-dHdq[i] for i in range(n)]

    # independent cross-check: Hamilton's equations by numerical differentiation of H
    def Hfun(*args):
        qq, pp = args[:n], mp.matrix(args[n:])
        Jq = mat(S["J"](*qq))
        Kq = Jq.T * M * Jq
        return (pp.T * mp.lu_solve(Kq, pp))[0] / 2 + S["U"](*qq)

    at = tuple(qv) + tuple(p)
    for i in range(n):
        od = [0] * (2 * n)
        od[n + i] = 1
        ref_dq = mp.diff(Hfun, at, tuple(od))
        od = [0] * (2 * n)
        od[i] = 1
        ref_dp = -mp.diff(Hfun, at, tuple(od))
        scale = 1 + abs(ref_dq) + abs(ref_dp)
        assert abs(ref_dq - dq[i]) < mp.mpf(10) ** -18 * scale, (spec.name, "dq", i, ref_dq, dq[i])
        assert abs(ref_dp - dp[i]) < mp.mpf(10) ** -18 * scale, (spec.name, "dp", i, ref_dp, dp[i])

    return dict(
        q=[fmt(v) for v in qv], qd=[fmt(v) for v in qdv], p=[fmt(v) for v in p],
        x=[fmt(v) for v in x], vel=[fmt(v) for v in vel],
        jac=[[fmt(J[k, i]) for i in range(n)] for k in range(m)],
        keC=fmt(keC), keP=fmt(keP), pe=fmt(U), lagrangian=fmt(keC - U), hamiltonian=fmt(keP + U),
        dq=[fmt(v) for v in dq], dp=[fmt(v) for v in dp],
        cond_hint=fmt(mp.norm(K, 1) * mp.norm(Ki, 1)),
    )


def trajectory_truth(spec, S, times):
    """y(t) from the reference initial Config (Examples.hs seInit), Taylor-series ODE solve."""
    n = spec.n
    M = mp.diag([mp.mpf(v) for v in spec.inertia])

    def rhs(t, y):
        qq, pp = list(y[:n]), mp.matrix(y[n:])
        J = mat(S["J"](*qq))
        dJ = [mat(r) for r in S["dJ"](*qq)]
        gU = S["gU"](*qq)
        K = J.T * M * J
        v = mp.lu_solve(K, pp)
        u = M * (J * v)
        out = [v[i] for i in range(n)]
        for i in range(n):
            dT = -(u.T * (dJ[i] * v))[0]
            out.append(-(dT + gU[i]))
        return out

    q0 = [mp.mpf(v) for v in spec.q0]
    qd0 = mp.matrix([mp.mpf(v) for v in spec.qd0])
    J0 = mat(S["J"](*q0))
    p0 = J0.T * (M * (J0 * qd0))
    y0 = q0 + [p0[i] for i in range(n)]
    sol = mp.odefun(rhs, 0, y0, tol=mp.mpf(10) ** -30, degree=30)
    rows = []
    for t in times:
        y = sol(mp.mpf(t))
        rows.append(dict(t=repr(float(t)), q=[fmt(v) for v in y[:n]], p=[fmt(v) for v in y[n:]]))
    return dict(q0=[fmt(v) for v in q0], p0=[fmt(p0[i]) for i in range(n)], states=rows)


def evaluate_chain_point(spec, qv, qdv):
    """The N-link chains of BASELINE config 5 (examples.chain) from their CLOSED-FORM mechanics -- no tape, no sympy, no AD:
    x_k = l sum_{j<=k} sin th_j, y_k = -l sum_{j<=k} cos th_j, unit inertias, U = 5 sum_k y_k give
        K[a][b] = l^2 (N - max(a, b)) cos(th_a - th_b)            (mass matrix J^T M J, 0-based a, b)
        U       = -5 l sum_j (N - j) cos th_j
        dq      = v = K^-1 p                                      (Hamilton.hs:381, :386; mpmath LU at 50 digits)
        dp_i    = 1/2 v^T (dK/dth_i) v - dU/dth_i
                = -l^2 v_i sum_b (N - max(i, b)) v_b sin(th_i - th_b) - 5 l (N - i) sin th_i
    (dK[a][b]/dth_i = -l^2 (N - max(a,b)) sin(th_a - th_b) (delta_ai - delta_bi); the two halves of the quadratic form
    are equal by antisymmetry).  `main` asserts that this path and the generic symbolic one (evaluate_point) agree to 40
    digits on chain4 and chain8 before it is used for N = 16, 32, where 2N x N x N symbolic second derivatives are slow."""
    N = spec.n
    ell = mp.mpf(1) / N
    th = [mp.mpf(v) for v in qv]
    qd = mp.matrix([mp.mpf(v) for v in qdv])
    c = lambda a, b: N - max(a, b)
    K = mp.matrix(N, N)
    for a in range(N):
        for b in range(N):
            K[a, b] = ell * ell * c(a, b) * mp.cos(th[a] - th[b])
    p = K * qd
    Ki = K ** -1
    v = Ki * p
    U = -5 * ell * sum((N - j) * mp.cos(th[j]) for j in range(N))
    keC = (qd.T * p)[0] / 2
    keP = (v.T * p)[0] / 2
    dp = []
    for i in range(N):
        acc = sum(c(i, b) * v[b] * mp.sin(th[i] - th[b]) for b in range(N))
        dp.append(-ell * ell * v[i] * acc - 5 * ell * (N - i) * mp.sin(th[i]))
    x = []
    ax, ay = mp.mpf(0), mp.mpf(0)
    for j in range(N):
        ax += ell * mp.sin(th[j])
        ay -= ell * mp.cos(th[j])
        x += [ax, ay]
    return dict(
        q=[fmt(t) for t in th], qd=[fmt(t) for t in qd], p=[fmt(t) for t in p], x=[fmt(t) for t in x],
        vel=[fmt(t) for t in v], keC=fmt(keC), keP=fmt(keP), pe=fmt(U), lagrangian=fmt(keC - U), hamiltonian=fmt(keP + U),
        dq=[fmt(t) for t in v], dp=[fmt(t) for t in dp], cond_hint=fmt(mp.norm(K, 1) * mp.norm(Ki, 1)),
    )


def chain_points(spec):
    """Point 0: the spec's initial Config (at rest); 12 points of the C5 sampling box with the chain MOVING (the box has
    qd = 0, where p = 0 and the solve is trivial): qd[j][i] = 0.3 cos(0.7 (12 j + i)), computed in fp64."""
    q, _ = E.sample_config(spec, 0, NPOINTS)
    qd = 0.3 * np.cos(0.7 * np.arange(spec.n * NPOINTS, dtype=np.float64).reshape(spec.n, NPOINTS))
    return [(spec.q0, spec.qd0)] + [(q[:, i], qd[:, i]) for i in range(NPOINTS)]


CLOSED_FORM_CHAINS = ["chain8", "chain16", "chain32"]

SYSTEMS = [
    ("pendulum", (0.01, 0.1, 1.0)),
    ("doublePendulum", (0.01, 0.1, 1.0)),
    ("room", (0.01, 0.1, 1.0)),
    ("twoBody", (0.01, 0.1, 1.0)),
    ("spring", (0.01, 0.1, 0.5)),
    ("bezier", (0.01, 0.1, 1.0)),
    ("threeBodyPolar", (0.002, 0.02)),
    ("chain4", ()),
    ("opcodeZoo", ()),
]


def main():
    mp.mp.dps = DIGITS
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    if not only or only & set(CLOSED_FORM_CHAINS):
        # the closed form against the generic symbolic derivation where both are cheap
        for name in ("chain4", "chain8"):
            spec = E.get(name)
            S = symbolic(spec)
            for qv, qdv in chain_points(spec)[:3]:
                a, b = evaluate_point(spec, S, qv, qdv), evaluate_chain_point(spec, qv, qdv)
                for key in ("p", "x", "vel", "dq", "dp", "keC", "keP", "pe", "hamiltonian"):
                    av = a[key] if isinstance(a[key], list) else [a[key]]
                    bv = b[key] if isinstance(b[key], list) else [b[key]]
                    for u, w in zip(av, bv):
                        assert abs(mp.mpf(u) - mp.mpf(w)) <= mp.mpf(10) ** -27 * (1 + abs(mp.mpf(u))), (name, key, u, w)
            print("closed-form chain mechanics == symbolic derivation:", name, flush=True)
        for name in CLOSED_FORM_CHAINS:
            if only and name not in only:
                continue
            spec = E.get(name)
            pts = [evaluate_chain_point(spec, qv, qdv) for qv, qdv in chain_points(spec)]
            doc = dict(
                system=name, m=spec.m, n=spec.n, inertia=list(spec.inertia), cite=spec.cite,
                generator="oracle/gen_golden.py evaluate_chain_point (mpmath %s, %d digits)" % (mp.__version__, DIGITS),
                note="derived fixtures from the chain's closed-form mass matrix K[a][b] = l^2 (N - max(a,b)) cos(th_a - th_b) and "
                     "Hamilton's equations written out by hand (no tape, no AD, no sympy); LU at 50 digits; checked against the "
                     "symbolic derivation on chain4 / chain8 by the generator.  Point 0: the spec's initial Config; points 1-12: "
                     "examples.sample_config(spec, 0, 12) positions with qd[j][i] = 0.3 cos(0.7 (12 j + i)).  No `jac` entry.",
                points=pts,
            )
            path = os.path.join(OUT, f"{name}.json")
            with open(path, "w") as fh:
                json.dump(doc, fh, indent=1)
            print("wrote", path, len(pts), "points", flush=True)
    for name, times in SYSTEMS:
        if only and name not in only:
            continue
        spec = E.get(name)
        S = symbolic(spec)
        pts = [evaluate_point(spec, S, spec.q0, spec.qd0)]
        q, qd = E.sample_config(spec, 0, NPOINTS)
        for i in range(NPOINTS):
            pts.append(evaluate_point(spec, S, q[:, i], qd[:, i]))
        doc = dict(
            system=name, m=spec.m, n=spec.n, inertia=list(spec.inertia), cite=spec.cite,
            generator="oracle/gen_golden.py (sympy %s, mpmath %s, %d digits)" % (sp.__version__, mp.__version__, DIGITS),
            note="derived fixtures: symbolic differentiation + 50-digit evaluation of Hamilton.hs:262-387; "
                 "point 0 is the reference's initial Config; others from examples.sample_config(spec, 0, %d)" % NPOINTS,
            points=pts,
        )
        if times:
            mp.mp.dps = 40
            doc["trajectory"] = trajectory_truth(spec, S, times)
            mp.mp.dps = DIGITS
        path = os.path.join(OUT, f"{name}.json")
        with open(path, "w") as fh:
            json.dump(doc, fh, indent=1)
        print("wrote", path, len(pts), "points", flush=True)


if __name__ == "__main__":
    main()
