#!/usr/bin/env python3
"""Generate tests/golden/byhand.json: 50-digit values of the reference's OWN example systems from mechanics written out by hand.

TEST INFRASTRUCTURE.  tests/golden/<system>.json (oracle/gen_golden.py) differentiates the Python restatements in
hamilton_amd/examples.py -- the same definitions that produce the tapes the oracle and the HIP kernels run, so a
transcription error against /root/reference/app/Examples.hs would be invisible to fixture, oracle and GPU alike.  This
generator shares NOTHING with that file: no import of hamilton_amd, no tape, no AD, no sympy.  For each system below the
mass matrix K(q) = J^T M J, the potential and Hamilton's equations

    dq = K^-1 p,        dp_i = 1/2 v^T (dK/dq_i) v - dU/dq_i,   v = K^-1 p

are written out by hand from the Haskell source (file:line per function) and evaluated with mpmath at 50 digits; the points
(q, qd) are read as data from the existing fixture files.  Each block is cross-checked in this script against central
differences of H(q, p) = 1/2 p.K^-1.p + U at 50 digits -- Hamilton's equations themselves.

Constants follow the Haskell text: a literal is the fp64 it denotes, products / quotients of fp64 parameters that the source
computes in Double (`realToFrac (-(m2 / mT))`, `log (0.9 / (1 - 0.9)) / width`) are computed in fp64 here and then widened.

Run:  python oracle/gen_golden_byhand.py        (seconds; output is committed)
"""
from __future__ import annotations

import json
import math
import os

import mpmath as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
DIGITS = 50
mp.mp.dps = DIGITS          # before anything below builds a constant (two_body_of's mu is a product of widened fp64 values)


def fmt(x) -> str:
    return mp.nstr(mp.mpf(x), 30, min_fixed=0, max_fixed=0)


def F(x: float):
    """An fp64 value, exactly."""
    return mp.mpf(float(x))


# ------------------------------------------------------------------------------------------------
# each system: a function (q, qd) -> everything the public API returns there, by hand
# ------------------------------------------------------------------------------------------------
def finish(q, qd, x, K, dK, U, dU):
    """p = K qd (Hamilton.hs:262-269 in closed form), v = K^-1 p (:316-324), energies (:288-361), hamEqs (:370-387)."""
    n = len(q)
    K = mp.matrix(K)
    qdv = mp.matrix(qd)
    p = K * qdv
    v = mp.lu_solve(K, p)
    keC = (qdv.T * p)[0] / 2
    keP = (v.T * p)[0] / 2
    dp = [(v.T * (mp.matrix(dK[i]) * v))[0] / 2 - dU[i] for i in range(n)]
    return dict(q=[fmt(t) for t in q], qd=[fmt(t) for t in qd], p=[fmt(t) for t in p], x=[fmt(t) for t in x],
                vel=[fmt(t) for t in v], keC=fmt(keC), keP=fmt(keP), pe=fmt(U), lagrangian=fmt(keC - U), hamiltonian=fmt(keP + U),
                dq=[fmt(t) for t in v], dp=[fmt(t) for t in dp])


def pendulum(q, qd):
    """app/Examples.hs:61-73: masses (1,1); (x, y) = (sin th, 0.5 - cos th); U = y.   K = cos^2 + sin^2 = 1."""
    th, = q
    return finish(q, qd, [mp.sin(th), F(0.5) - mp.cos(th)], [[mp.mpf(1)]], [[[mp.mpf(0)]]], F(0.5) - mp.cos(th), [mp.sin(th)])


def double_pendulum_of(m1: float, m2: float, y_offset: int, g: float = 5.0):
    """app/Examples.hs:75-94 (y_offset 1: y1 = 1 - cos th1, y2 = 1 - cos th1 - cos th2 / 2) and README.md:88-103 (y_offset 0: no `1 -`;
    masses 1 1 2 2, U = (y1 + 2 y2) * 5).  Link 1 has length 1, link 2 length 1/2:
        T = 1/2 m1 w1^2 + 1/2 m2 (w1^2 + w2^2/4 + w1 w2 cos(th1 - th2))
        K = [[m1 + m2, m2/2 cos D], [m2/2 cos D, m2/4]],  D = th1 - th2
        U = g (m1 y1 + m2 y2)"""
    def f(q, qd):
        t1, t2 = q
        a, b = F(m1), F(m2)
        D = t1 - t2
        c, s = mp.cos(D), mp.sin(D)
        off = mp.mpf(y_offset)
        y1 = off - mp.cos(t1)
        y2 = off - mp.cos(t1) - mp.cos(t2) / 2
        x = [mp.sin(t1), y1, mp.sin(t1) + mp.sin(t2) / 2, y2]
        K = [[a + b, b / 2 * c], [b / 2 * c, b / 4]]
        dK = [[[0, -b / 2 * s], [-b / 2 * s, 0]], [[0, b / 2 * s], [b / 2 * s, 0]]]
        U = F(g) * (a * y1 + b * y2)
        dU = [F(g) * (a + b) * mp.sin(t1), F(g) * b * mp.sin(t2) / 2]
        return finish(q, qd, x, K, dK, U, dU)
    return f


def two_body_of(m1: float, m2: float):
    """app/Examples.hs:118-142: masses (m1,m1,m2,m2); r1 = r * (-(m2/mT)), r2 = r * (m1/mT) with the two ratios computed in Double;
    x = (r1 cos th, r1 sin th, r2 cos th, r2 sin th); U = -(m1 m2)/r.  Each body moves on a circle of radius |c_i| r:
        K = diag(mu, mu r^2),  mu = m1 c1^2 + m2 c2^2   (= m1 m2 / mT up to the rounding of c1, c2)
        th is cyclic: dp_th = 0;  dp_r = mu r v_th^2 - m1 m2 / r^2"""
    mT = m1 + m2
    c1, c2 = F(-(m2 / mT)), F(m1 / mT)
    mu = F(m1) * c1 * c1 + F(m2) * c2 * c2
    gm = F(m1 * m2)

    def f(q, qd):
        r, th = q
        x = [r * c1 * mp.cos(th), r * c1 * mp.sin(th), r * c2 * mp.cos(th), r * c2 * mp.sin(th)]
        K = [[mu, 0], [0, mu * r * r]]
        dK = [[[0, 0], [0, 2 * mu * r]], [[0, 0], [0, 0]]]
        return finish(q, qd, x, K, dK, -(gm / r), [gm / (r * r), mp.mpf(0)])
    return f


def logistic(pos: float, ht: float, width: float):
    """app/Examples.hs:601-605: ht / (1 + exp(-(beta (x - pos)))), beta = log(0.9 / (1 - 0.9)) / width in Double.  Returns (L, L')."""
    beta = F(math.log(0.9 / (1 - 0.9)) / width)

    def L(x):
        e = mp.exp(-(beta * (x - F(pos))))
        return F(ht) / (1 + e), F(ht) * beta * e / (1 + e) ** 2
    return L


def room(q, qd):
    """app/Examples.hs:96-116: masses (1,1), identity coordinates (K = I); U = 2y + (1 - L(-1,10,.1)(y)) + L(1,10,.1)(y)
    + (1 - L(-2,10,.1)(x)) + L(2,10,.1)(x)."""
    x, y = q
    (b, db), (t, dt_) = logistic(-1, 10, 0.1)(y), logistic(1, 10, 0.1)(y)
    (le, dle), (ri, dri) = logistic(-2, 10, 0.1)(x), logistic(2, 10, 0.1)(x)
    U = 2 * y + (1 - b) + t + (1 - le) + ri
    Z = [[0, 0], [0, 0]]
    return finish(q, qd, [x, y], [[1, 0], [0, 1]], [Z, Z], U, [-dle + dri, 2 - db + dt_])


def spring_of(mB: float, mW: float, k: float):
    """app/Examples.hs:144-162: masses (mB, mW, mW); coordinates (r, r + (1+x) sin th, (1+x)(-cos th)); U = k x^2/2 + (1 - L(-1.5,25,.1)(r))
    + L(1.5,25,.1)(r) + mB (1+x)(-cos th)  -- gravity on the weight multiplies by mB as the source does (:157).  With l = 1 + x:
        K = [[mB + mW, mW sin th, mW l cos th], [mW sin th, mW, 0], [mW l cos th, 0, mW l^2]]"""
    a, w, kk = F(mB), F(mW), F(k)

    def f(q, qd):
        r, x, th = q
        l, s, c = 1 + x, mp.sin(th), mp.cos(th)
        (le, dle), (ri, dri) = logistic(-1.5, 25, 0.1)(r), logistic(1.5, 25, 0.1)(r)
        K = [[a + w, w * s, w * l * c], [w * s, w, 0], [w * l * c, 0, w * l * l]]
        Z = [[0, 0, 0], [0, 0, 0], [0, 0, 0]]
        dKx = [[0, 0, w * c], [0, 0, 0], [w * c, 0, 2 * w * l]]
        dKt = [[0, w * c, -w * l * s], [w * c, 0, 0], [-w * l * s, 0, 0]]
        U = kk * x * x / 2 + (1 - le) + ri + a * (l * (-c))
        dU = [-dle + dri, kk * x - a * c, a * l * s]
        return finish(q, qd, [r, r + l * s, l * (-c)], K, [Z, dKx, dKt], U, dU)
    return f


BLOCKS = [
    # block name,            fixture file the (q, qd) points come from, extra points,                         by-hand mechanics
    ("pendulum",              "pendulum",       [],                                                            pendulum),
    ("doublePendulum",        "doublePendulum", [],                                                            double_pendulum_of(1.0, 1.0, 1)),
    ("doublePendulumReadme",  "doublePendulum", [(("1", "0"), ("0", "0.5"))],                                  double_pendulum_of(1.0, 2.0, 0)),
    ("twoBody",               "twoBody",        [],                                                            two_body_of(5.0, 0.5)),
    ("room",                  "room",           [],                                                            room),
    ("spring",                "spring",         [],                                                            spring_of(2.0, 1.0, 10.0)),
]
CITE = {"pendulum": "app/Examples.hs:61-73", "doublePendulum": "app/Examples.hs:75-94 (defaults m1 = m2 = 1, :250-267)",
        "doublePendulumReadme": "README.md:88-103 (masses 1 1 2 2, y = -cos, U = (y1 + 2 y2) * 5), config0 README.md:124-126",
        "twoBody": "app/Examples.hs:118-142 (defaults m1 = 5, m2 = 0.5, :279-305)", "room": "app/Examples.hs:96-116, logistic :601-605",
        "spring": "app/Examples.hs:144-162 (defaults mB = 2, mW = 1, k = 10, :306-341), logistic :601-605"}


def check_against_hamiltons_equations(name, fn, pt):
    """dq = dH/dp, dp = -dH/dq by 50-digit central differences of H assembled from the block's own K and U."""
    q = [mp.mpf(t) for t in pt["q"]]
    p = [mp.mpf(t) for t in pt["p"]]
    n = len(q)

    def H(*a):
        qq, pp = list(a[:n]), mp.matrix(a[n:])
        zero = [mp.mpf(0)] * n
        # K v = p at qq: read K off the block by applying it to unit velocities (p = K qd)
        cols = [[mp.mpf(t) for t in fn(qq, [mp.mpf(1) if j == i else mp.mpf(0) for j in range(n)])["p"]] for i in range(n)]
        K = mp.matrix([[cols[j][i] for j in range(n)] for i in range(n)])
        U = mp.mpf(fn(qq, zero)["pe"])
        return (pp.T * mp.lu_solve(K, pp))[0] / 2 + U

    at = tuple(q) + tuple(p)
    for i in range(n):
        od = [0] * (2 * n); od[n + i] = 1
        dq = mp.diff(H, at, tuple(od), h=mp.mpf(10) ** -12)
        od = [0] * (2 * n); od[i] = 1
        dp = -mp.diff(H, at, tuple(od), h=mp.mpf(10) ** -12)
        # the fixture strings carry 30 digits (fmt), so H above is assembled from 30-digit K and U: 1e-16 is what survives the differences
        scale = 1 + abs(dq) + abs(dp)
        assert abs(dq - mp.mpf(pt["dq"][i])) < mp.mpf(10) ** -15 * scale, (name, "dq", i, dq, pt["dq"][i])
        assert abs(dp - mp.mpf(pt["dp"][i])) < mp.mpf(10) ** -15 * scale, (name, "dp", i, dp, pt["dp"][i])


def main():
    mp.mp.dps = DIGITS
    blocks = {}
    for name, src, extra, fn in BLOCKS:
        with open(os.path.join(GOLDEN, f"{src}.json")) as fh:
            pts_in = [(pt["q"], pt["qd"]) for pt in json.load(fh)["points"]]
        pts = [fn([mp.mpf(t) for t in q], [mp.mpf(t) for t in qd]) for q, qd in list(extra) + pts_in]
        for pt in pts[:4]:
            check_against_hamiltons_equations(name, fn, pt)
        blocks[name] = dict(cite=CITE[name], points=pts)
        print(f"{name}: {len(pts)} points, first {min(4, len(pts))} checked against central differences of H", flush=True)
    doc = dict(generator="oracle/gen_golden_byhand.py (mpmath %s, %d digits)" % (mp.__version__, DIGITS),
               note="derived fixtures: mass matrix, potential and Hamilton's equations of the reference's example systems written out BY HAND "
                    "from app/Examples.hs / README.md -- no tape, no AD, no sympy, nothing imported from hamilton_amd; points (q, qd) are those of "
                    "tests/golden/<system>.json (doublePendulumReadme: README's config0 first, then the doublePendulum points)",
               blocks=blocks)
    path = os.path.join(GOLDEN, "byhand.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
