"""The reference's example systems, restated as polymorphic Python functions.

Each definition follows /root/reference/app/Examples.hs (line numbers per
function below) with the CLI defaults of Examples.hs:230-359.  They are the
only concrete inputs the reference provides and therefore the benchmark and
parity workloads.  Two further systems named by
BASELINE.json's configs 4 and 5 have no reference counterpart and are defined
here as SURVEY.md section 8d specifies (`three_body_polar`, `chain`).

A definition is a `SystemSpec`: inertias, coordinate map `f`, potential `u`,
whether `u` is over generalized (mkSystem) or cartesian (mkSystem') inputs, the
reference's initial `Config`, and the ensemble sampling box used by bench and
tests.  `f`/`u` only use the vocabulary of `hamilton_amd.tracer` so they run
on floats, on `Var` (tape recording), and on mpmath/sympy numbers alike.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Sequence, Tuple

import numpy as np

from . import tracer as T

U_GENERALIZED = 0   # mkSystem  (Hamilton.hs:201-225)
U_CARTESIAN = 1     # mkSystem' (Hamilton.hs:238-254)


class _Ops:
    """Floating-method dispatch so one definition serves floats, Var, mpmath, sympy."""

    def __init__(self, mod=None):
        self.mod = mod

    def __getattr__(self, name):
        if self.mod is not None:
            return getattr(self.mod, name)
        return getattr(T, name)


_T = _Ops()


@dataclass
class SystemSpec:
    name: str
    m: int
    n: int
    inertia: Tuple[float, ...]
    f: Callable          # f(q: list, ops) -> list of m
    u: Callable          # u(z: list, ops) -> scalar; z = q (U_GENERALIZED) or x (U_CARTESIAN)
    u_space: int
    q0: Tuple[float, ...]            # reference initial Config positions
    qd0: Tuple[float, ...]           # reference initial Config velocities
    q_box: Tuple[Tuple[float, float], ...] = ()    # ensemble sampling box for positions
    qd_box: Tuple[Tuple[float, float], ...] = ()   # ... and velocities
    dt: float = 0.01
    cite: str = ""

    def coords(self, q, ops=_T):
        return list(self.f(list(q), ops))

    def potential_of_q(self, q, ops=_T):
        """U as a function of generalized coordinates (mkSystem' composes u . f, :254)."""
        if self.u_space == U_CARTESIAN:
            return self.u(self.coords(q, ops), ops)
        return self.u(list(q), ops)

    def trace(self):
        tf = T.trace(lambda q: self.f(q, _T), self.n, self.m)
        n_u_in = self.m if self.u_space == U_CARTESIAN else self.n
        tu = T.trace(lambda z: self.u(z, _T), n_u_in, None)
        return tf, tu


# ---------------------------------------------------------------------------------
# helpers restated from Examples.hs
# ---------------------------------------------------------------------------------

def logistic(pos, ht, width, x, ops=_T):
    """Examples.hs:601-605.  beta = log(0.9/(1-0.9))/width, computed in fp64 as written."""
    beta = math.log(0.9 / (1 - 0.9)) / width
    return ht / (1 + ops.exp(-(beta * (x - pos))))


def _choose(n, k):
    return math.factorial(n) // (math.factorial(n - k) * math.factorial(k))


def bezier_curve(ps: Sequence[Tuple[float, float]], t, ops=_T):
    """Examples.hs:607-627: sum_i C(n,i) (1-t)^(n-i) t^i * P_i, folded from `pure 0`."""
    npts = len(ps) - 1
    acc = [0.0, 0.0]
    for i, pt in enumerate(ps):
        w = _choose(npts, i) * (1 - t) ** (npts - i) * t ** i
        acc = [acc[0] + pt[0] * w, acc[1] + pt[1] * w]
    return acc


# ---------------------------------------------------------------------------------
# the six reference systems
# ---------------------------------------------------------------------------------

def pendulum(theta0_deg: float = 0.0, omega0: float = 1.0) -> SystemSpec:
    """Examples.hs:61-73; defaults :230-249 (-a 0 -v 1); degrees -> radians at :391."""
    theta0 = theta0_deg / 180 * math.pi
    return SystemSpec(
        name="pendulum", m=2, n=1, inertia=(1.0, 1.0),
        f=lambda q, o: [o.sin(q[0]), 0.5 - o.cos(q[0])],
        u=lambda x, o: x[1],
        u_space=U_CARTESIAN,
        q0=(theta0,), qd0=(omega0,),
        q_box=((-math.pi, math.pi),), qd_box=((-1.0, 1.0),),
        cite="app/Examples.hs:61-73")


def double_pendulum(m1: float = 1.0, m2: float = 1.0) -> SystemSpec:
    """Examples.hs:75-94; defaults :250-267 (--m1 1 --m2 1)."""
    def f(q, o):
        t1, t2 = q
        return [o.sin(t1), 1 - o.cos(t1), o.sin(t1) + o.sin(t2) / 2, 1 - o.cos(t1) - o.cos(t2) / 2]

    def u(x, o):
        return 5 * (m1 * x[1] + m2 * x[3])

    return SystemSpec(
        name="doublePendulum", m=4, n=2, inertia=(m1, m1, m2, m2), f=f, u=u,
        u_space=U_CARTESIAN,
        q0=(math.pi / 2, 0.0), qd0=(0.0, 0.0),
        q_box=((-math.pi, math.pi), (-math.pi, math.pi)),
        qd_box=((-1.0, 1.0), (-1.0, 1.0)),
        cite="app/Examples.hs:75-94")


def double_pendulum_readme() -> SystemSpec:
    """README.md:88-103, the package's worked example: masses (1,1,2,2), y measured from the pivot (`-cos`, no `1 -`), potential
    `(y1 + 2 * y2) * 5`; config0 = Cfg (1, 0) (0, 0.5) (README.md:124-126).  Same dynamics as `double_pendulum(1, 2)`; U differs by a constant."""
    def f(q, o):
        t1, t2 = q
        return [o.sin(t1), -o.cos(t1), o.sin(t1) + o.sin(t2) / 2, -o.cos(t1) - o.cos(t2) / 2]

    def u(x, o):
        return (x[1] + 2 * x[3]) * 5

    return SystemSpec(
        name="doublePendulumReadme", m=4, n=2, inertia=(1.0, 1.0, 2.0, 2.0), f=f, u=u,
        u_space=U_CARTESIAN,
        q0=(1.0, 0.0), qd0=(0.0, 0.5),
        q_box=((-math.pi, math.pi), (-math.pi, math.pi)),
        qd_box=((-1.0, 1.0), (-1.0, 1.0)),
        cite="README.md:88-103, :124-126")


def room(theta_deg: float = 45.0) -> SystemSpec:
    """Examples.hs:96-116; default :268-278 (-a 45); the app converts degrees to radians."""
    th = theta_deg / 180 * math.pi      # Examples.hs:392

    def u(q, o):
        x, y = q
        return (2 * y
                + (1 - logistic(-1, 10, 0.1, y, o))
                + logistic(1, 10, 0.1, y, o)
                + (1 - logistic(-2, 10, 0.1, x, o))
                + logistic(2, 10, 0.1, x, o))

    return SystemSpec(
        name="room", m=2, n=2, inertia=(1.0, 1.0),
        f=lambda q, o: [q[0], q[1]], u=u, u_space=U_GENERALIZED,
        q0=(-1.0, 0.25), qd0=(math.cos(th), math.sin(th)),
        q_box=((-1.5, 1.5), (-0.7, 0.7)), qd_box=((-1.0, 1.0), (-1.0, 1.0)),
        cite="app/Examples.hs:96-116")


def two_body(m1: float = 5.0, m2: float = 0.5, omega0: float = 0.5) -> SystemSpec:
    """Examples.hs:118-142; defaults :279-305 (--m1 5 --m2 0.5 -v 0.5)."""
    mT = m1 + m2

    def f(q, o):
        r, th = q
        r1 = r * (-(m2 / mT))
        r2 = r * (m1 / mT)
        return [r1 * o.cos(th), r1 * o.sin(th), r2 * o.cos(th), r2 * o.sin(th)]

    def u(q, o):
        return -((m1 * m2) / q[0])

    return SystemSpec(
        name="twoBody", m=4, n=2, inertia=(m1, m1, m2, m2), f=f, u=u,
        u_space=U_GENERALIZED,
        q0=(2.0, 0.0), qd0=(0.0, omega0),
        q_box=((1.5, 2.5), (0.0, 2 * math.pi)),
        qd_box=((-0.05, 0.05), (0.3, 0.6)),
        cite="app/Examples.hs:118-142")


def spring(mB: float = 2.0, mW: float = 1.0, k: float = 10.0, x0: float = 0.1) -> SystemSpec:
    """Examples.hs:144-162; defaults :306-341 (-b 2 -w 1 -k 10 -x 0.1).

    The gravity term multiplies by mB (block mass), Examples.hs:157 -- restated
    literally, not "fixed" (SURVEY.md Appendix B)."""
    def f(q, o):
        r, x, th = q
        return [r, r + (1 + x) * o.sin(th), (1 + x) * (-o.cos(th))]

    def u(q, o):
        r, x, th = q
        return (k * x ** 2 / 2
                + (1 - logistic(-1.5, 25, 0.1, r, o))
                + logistic(1.5, 25, 0.1, r, o)
                + mB * ((1 + x) * (-o.cos(th))))

    return SystemSpec(
        name="spring", m=3, n=3, inertia=(mB, mW, mW), f=f, u=u,
        u_space=U_GENERALIZED,
        q0=(0.0, x0, 0.0), qd0=(1.0, 0.0, -0.5),
        q_box=((-1.0, 1.0), (-0.2, 0.2), (-0.5, 0.5)),
        qd_box=((-0.5, 0.5), (-0.5, 0.5), (-0.5, 0.5)),
        cite="app/Examples.hs:144-162")


BEZIER_DEFAULT = ((-1.0, -1.0), (-2.0, 1.0), (0.0, 1.0), (1.0, -1.0), (2.0, 1.0))  # Examples.hs:350


def bezier(ps: Sequence[Tuple[float, float]] = BEZIER_DEFAULT) -> SystemSpec:
    """Examples.hs:164-183, bezierCurve :607-627; default control points :350."""
    ps = tuple((float(a), float(b)) for a, b in ps)

    def u(q, o):
        t = q[0]
        return (1 - logistic(0, 5, 0.05, t, o)) + logistic(1, 5, 0.05, t, o)

    return SystemSpec(
        name="bezier", m=2, n=1, inertia=(1.0, 1.0),
        f=lambda q, o: bezier_curve(ps, q[0], o), u=u, u_space=U_GENERALIZED,
        q0=(0.5,), qd0=(0.25,),
        q_box=((0.2, 0.8),), qd_box=((-0.3, 0.3),),
        cite="app/Examples.hs:164-183")


# ---------------------------------------------------------------------------------
# BASELINE.json configs 4 and 5 -- NOT in the reference
# ---------------------------------------------------------------------------------

def three_body_polar() -> SystemSpec:
    """Config 4: three unit-mass planar bodies in polar coordinates, System 6 6.

    q = (r1, phi1, r2, phi2, r3, phi3); x = (r_i cos phi_i, r_i sin phi_i);
    U = -sum_{i<j} 1/|x_i - x_j| on cartesian coordinates.  Build-defined."""
    def f(q, o):
        out = []
        for i in range(3):
            r, ph = q[2 * i], q[2 * i + 1]
            out += [r * o.cos(ph), r * o.sin(ph)]
        return out

    def u(x, o):
        acc = 0.0
        for i in range(3):
            for j in range(i + 1, 3):
                dx = x[2 * i] - x[2 * j]
                dy = x[2 * i + 1] - x[2 * j + 1]
                acc = acc - 1 / o.sqrt(dx * dx + dy * dy)
        return acc

    q_box, qd_box, q0 = [], [], []
    for i in range(3):
        c = 2 * math.pi * i / 3
        q_box += [(1.0, 2.0), (c - 0.3, c + 0.3)]
        qd_box += [(-0.05, 0.05), (0.2, 0.4)]
        q0 += [1.5, c]
    return SystemSpec(
        name="threeBodyPolar", m=6, n=6, inertia=(1.0,) * 6, f=f, u=u,
        u_space=U_CARTESIAN,
        q0=tuple(q0), qd0=(0.0, 0.3) * 3,
        q_box=tuple(q_box), qd_box=tuple(qd_box), dt=0.002,
        cite="SURVEY.md section 8d C4 (no reference counterpart)")


def chain(N: int) -> SystemSpec:
    """Config 5: N-link planar pendulum chain, System (2N) N.  Build-defined.

    x_k = sum_{j<=k} l sin th_j, y_k = -sum_{j<=k} l cos th_j, l = 1/N, unit
    inertias, U = 5 sum_k y_k (cartesian)."""
    ell = 1.0 / N

    def f(q, o):
        xs, ys = [], []
        ax, ay = 0.0, 0.0
        for j in range(N):
            ax = ax + ell * o.sin(q[j])
            ay = ay - ell * o.cos(q[j])
            xs.append(ax)
            ys.append(ay)
        out = []
        for k in range(N):
            out += [xs[k], ys[k]]
        return out

    def u(x, o):
        acc = 0.0
        for k in range(N):
            acc = acc + x[2 * k + 1]
        return 5 * acc

    return SystemSpec(
        name=f"chain{N}", m=2 * N, n=N, inertia=(1.0,) * (2 * N), f=f, u=u,
        u_space=U_CARTESIAN,
        q0=tuple(0.5 for _ in range(N)), qd0=(0.0,) * N,
        q_box=tuple((-math.pi / 2, math.pi / 2) for _ in range(N)),
        qd_box=tuple((0.0, 0.0) for _ in range(N)), dt=0.005,
        cite="SURVEY.md section 8d C5 (no reference counterpart)")


def dense(N: int) -> SystemSpec:
    """`denseN`: a System N N whose coordinate map has a DENSE Jacobian (build-defined benchmark workload for the
    wave-cooperative / matrix-core kernels, BASELINE.json configs[4] "larger-n dense solve / MFMA crossover").

    x_k = 2 q_k + sum_j (a_kj sin q_j + b_kj cos q_j) with small fixed a, b: every dx_k/dq_j is non-zero and distinct (n^2
    Jacobian entries, so the four-lane mapping is never chosen), n sincos evaluations per right-hand side, K = J^T J stays
    well conditioned (J = 2 I + O(0.1 n^(1/2))); U = 1/2 |x|^2 keeps the motion bounded."""
    s = 1.0 / N

    def f(q, o):
        sn = [o.sin(q[j]) for j in range(N)]
        cs = [o.cos(q[j]) for j in range(N)]
        out = []
        for k in range(N):
            acc = 2.0 * q[k]
            for j in range(N):
                a = s * (0.2 + 0.1 * ((3 * k + 7 * j) % 11))
                b = s * (0.15 + 0.1 * ((5 * k + 2 * j) % 7))
                acc = acc + a * sn[j] + b * cs[j]
            out.append(acc)
        return out

    def u(x, o):
        acc = 0.0
        for k in range(N):
            acc = acc + x[k] * x[k]
        return 0.5 * acc

    return SystemSpec(
        name=f"dense{N}", m=N, n=N, inertia=(1.0,) * N, f=f, u=u, u_space=U_CARTESIAN,
        q0=tuple(0.1 for _ in range(N)), qd0=(0.0,) * N,
        q_box=tuple((-1.0, 1.0) for _ in range(N)), qd_box=tuple((-0.5, 0.5) for _ in range(N)), dt=0.01,
        cite="build-defined (dense-Jacobian benchmark system; no reference counterpart)")


def dense_distinct(N: int) -> SystemSpec:
    """`denseDN`: `denseN` with every coefficient DISTINCT (a_kj, b_kj from a fixed irrational sequence instead of the 11 x 7 repeating
    values of `dense`): no product a_kj sin q_j is shared between two outputs, which is what a dense map that was not built from a
    small table looks like to the code generator (build-defined benchmark / test workload)."""
    s = 1.0 / N
    frac = lambda x: x - math.floor(x)

    def f(q, o):
        sn = [o.sin(q[j]) for j in range(N)]
        cs = [o.cos(q[j]) for j in range(N)]
        out = []
        for k in range(N):
            acc = 2.0 * q[k]
            for j in range(N):
                a = s * (0.2 + frac(0.6180339887498949 * (1 + k * N + j)))
                b = s * (0.15 + 0.7 * frac(0.4142135623730951 * (3 + j * N + k)))
                acc = acc + a * sn[j] + b * cs[j]
            out.append(acc)
        return out

    def u(x, o):
        acc = 0.0
        for k in range(N):
            acc = acc + x[k] * x[k]
        return 0.5 * acc

    return SystemSpec(
        name=f"denseD{N}", m=N, n=N, inertia=(1.0,) * N, f=f, u=u, u_space=U_CARTESIAN,
        q0=tuple(0.1 for _ in range(N)), qd0=(0.0,) * N,
        q_box=tuple((-1.0, 1.0) for _ in range(N)), qd_box=tuple((-0.5, 0.5) for _ in range(N)), dt=0.01,
        cite="build-defined (dense-Jacobian benchmark system, distinct coefficients; no reference counterpart)")


def dense_mixed(N: int) -> SystemSpec:
    """`denseMixedN`: a System (N+1) N with a dense Jacobian whose sincos sites are NOT inputs (sin(q_j + 0.3 q_(j+1))) and a
    potential over the GENERALIZED coordinates (build-defined test workload: the branches of the four-lane kernels' dense path that
    `denseN` -- sites = inputs, cartesian potential -- does not reach)."""
    s = 1.0 / N

    def f(q, o):
        sn = [o.sin(q[j] + 0.3 * q[(j + 1) % N]) for j in range(N)]
        out = []
        for k in range(N):
            acc = 1.5 * q[k]
            for j in range(N):
                acc = acc + (s * (0.2 + 0.1 * ((3 * k + 7 * j) % 11))) * sn[j]
            out.append(acc)
        acc = 0.0
        for j in range(N):
            acc = acc + (0.1 * s) * o.cos(q[j])
        out.append(acc)
        return out

    def u(q, o):
        acc = 0.0
        for j in range(N):
            acc = acc + 0.5 * q[j] * q[j] + 0.1 * o.cos(q[j] - q[(j + 1) % N])
        return acc

    return SystemSpec(
        name=f"denseMixed{N}", m=N + 1, n=N, inertia=tuple(1.0 + 0.5 * (k % 2) for k in range(N + 1)), f=f, u=u, u_space=U_GENERALIZED,
        q0=tuple(0.1 for _ in range(N)), qd0=(0.0,) * N,
        q_box=tuple((-1.0, 1.0) for _ in range(N)), qd_box=tuple((-0.5, 0.5) for _ in range(N)), dt=0.01,
        cite="build-defined (dense-Jacobian test system, sites not inputs, generalized potential; no reference counterpart)")


def pendulums(N: int) -> SystemSpec:
    """`pendulumsN`: N UNCOUPLED unit pendulums as one System (2N) N (build-defined test workload): x_k = sin q_k, y_k = -cos q_k,
    U = 5 sum y_k.  Every output depends on exactly one input, K is diagonal: the wave kernels' accumulation of K issues one
    16-column block per four rows, at a block offset that grows with k -- the case of `SinkK::flush<LO, HI>` a chain (LO = 0)
    and a dense map (everything) never reach."""
    def f(q, o):
        out = []
        for k in range(N):
            out += [o.sin(q[k]), -o.cos(q[k])]
        return out

    def u(x, o):
        acc = 0.0
        for k in range(N):
            acc = acc + x[2 * k + 1]
        return 5 * acc

    return SystemSpec(
        name=f"pendulums{N}", m=2 * N, n=N, inertia=tuple(1.0 + 0.25 * (k % 3) for k in range(2 * N)), f=f, u=u, u_space=U_CARTESIAN,
        q0=tuple(0.3 for _ in range(N)), qd0=(0.0,) * N,
        q_box=tuple((-1.5, 1.5) for _ in range(N)), qd_box=tuple((-1.0, 1.0) for _ in range(N)), dt=0.01,
        cite="build-defined (block-sparse Jacobian test system; no reference counterpart)")


REGISTRY = {
    "pendulum": pendulum,
    "doublePendulum": double_pendulum,
    "doublePendulumReadme": double_pendulum_readme,
    "room": room,
    "twoBody": two_body,
    "spring": spring,
    "bezier": bezier,
    "threeBodyPolar": three_body_polar,
}


def mixed_inertia(base: str) -> SystemSpec:
    """`base` with inertias of MIXED SIGN (not a physical example; name `<base>~mixed`).

    The reference never asks whether K = J^T M J is definite: `velocities` and `hamEqs` invert it with
    hmatrix `inv` -- LU with partial pivoting (Hamilton.hs:321, :381) -- so a `System` whose inertia
    vector has non-positive entries is a legal input there as long as K is invertible.  These specs are
    the parity workload for that case: K symmetric, indefinite, generically invertible (per-point
    tolerances scale with cond K).  Pattern: inertia k keeps its magnitude and is negated (x 0.6) where
    (7 k + 3) mod 5 == 0 -- for the chains the x and y inertias of a body then differ in sign."""
    spec = get(base)
    w = tuple((-0.6 * v) if (7 * k + 3) % 5 == 0 else v for k, v in enumerate(spec.inertia))
    from dataclasses import replace
    return replace(spec, name=f"{base}~mixed", inertia=w,
                   cite=spec.cite + "; inertias of mixed sign (build-defined, Hamilton.hs:321/:381 `inv`)")


def get(name: str) -> SystemSpec:
    if name.endswith("~mixed"):
        return mixed_inertia(name[:-6])
    if name.startswith("chain"):
        return chain(int(name[5:]))
    if name.startswith("denseD"):
        return dense_distinct(int(name[6:]))
    if name.startswith("denseMixed"):
        return dense_mixed(int(name[10:]))
    if name.startswith("dense"):
        return dense(int(name[5:]))
    if name.startswith("pendulums"):
        return pendulums(int(name[9:]))
    return REGISTRY[name]()


# ---------------------------------------------------------------------------------
# ensemble sampling: counter-based, shard-invariant
# ---------------------------------------------------------------------------------

SEED = 20241008
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(index: np.ndarray, fld: int, seed: int = SEED) -> np.ndarray:
    """U[0,1) from (seed, global trajectory index, field): same value on any shard layout."""
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) ^ (index.astype(np.uint64) * np.uint64(0xD1342543DE82EF95))) & _M64
        key = _splitmix64(key + np.uint64(fld) * np.uint64(0x2545F4914F6CDD1D))
    return (key >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def sample_config(spec: SystemSpec, start: int, count: int, seed: int = SEED):
    """Positions q[n][count] and velocities qd[n][count] of trajectories start..start+count-1."""
    idx = np.arange(start, start + count, dtype=np.uint64)
    q = np.empty((spec.n, count), dtype=np.float64)
    qd = np.empty((spec.n, count), dtype=np.float64)
    for j in range(spec.n):
        lo, hi = spec.q_box[j]
        q[j] = lo + (hi - lo) * uniform01(idx, 2 * j, seed)
        lo, hi = spec.qd_box[j]
        qd[j] = lo + (hi - lo) * uniform01(idx, 2 * j + 1, seed)
    return q, qd


def opcode_zoo() -> SystemSpec:
    """Not a physical example: a System 4 2 whose coordinate map and potential together use
    EVERY tape opcode (include/hamk.h), so the parity tests exercise each derivative rule of
    the device jets (value, gradient, second order) against the oracle's independent rules."""
    def f(q, o):
        a, b = q
        e = o.exp(-(0.3 * a)) + o.log(2.0 + b * b)
        t = o.tan(0.4 * a) + o.tanh(b) - o.atan(a * b)
        h = o.sinh(0.5 * a) * o.cosh(0.25 * b) + o.sqrt(3.0 + a + b * b)
        g = o.asin(0.3 * o.sin(a)) + o.acos(0.4 * o.cos(b)) + 1 / (2.5 + o.sin(a + b))
        return [a + 0.1 * e, b + 0.1 * t, 0.2 * h + a * b / 3, 0.1 * g - b]

    def u(q, o):
        a, b = q
        return (o.atan2(1.5 + a, 2.0 + b) + (2.0 + a) ** (1.3 + 0.1 * b) + (3.0 + b) ** 1.7 + (a - 0.2) ** 3
                + (2.0 + a * a) ** (-2) + o.asinh(a - b) + o.acosh(2.0 + b * b) + o.atanh(0.3 * o.sin(a * b))
                + 2.0 ** (0.5 * a) - a / (1.5 + b * b))

    return SystemSpec(
        name="opcodeZoo", m=4, n=2, inertia=(1.0, 2.0, 0.5, 1.5), f=f, u=u, u_space=U_GENERALIZED,
        q0=(0.3, -0.2), qd0=(0.5, -0.4),
        q_box=((-0.8, 0.8), (-0.8, 0.8)), qd_box=((-1.0, 1.0), (-1.0, 1.0)),
        cite="build-defined (covers every hamk_opcode)")


REGISTRY["opcodeZoo"] = opcode_zoo


def abs_zoo() -> SystemSpec:
    """Not a physical example: a System 3 2 whose coordinate map and potential use the two Num methods
    the reference's own systems never call -- `abs` and `signum` (opcodes 27, 28; derivative of |x| is
    signum x, as `ad` differentiates it).  The sampling box stays away from the kinks at 0."""
    def f(q, o):
        a, b = q
        return [a + 0.2 * abs(b), b + 0.1 * abs(a) * a, abs(a - b) + 0.3 * o.sin(a)]

    def u(q, o):
        a, b = q
        return abs(a) * abs(b) + 0.3 * o.signum(b) * b ** 2 + abs(o.cos(a + b))

    return SystemSpec(
        name="absZoo", m=3, n=2, inertia=(1.0, 2.0, 1.5), f=f, u=u, u_space=U_GENERALIZED,
        q0=(0.5, -0.6), qd0=(0.2, -0.3),
        q_box=((0.3, 0.9), (-1.0, -0.35)), qd_box=((-0.5, 0.5), (-0.5, 0.5)),
        cite="build-defined (hamk_opcode 27, 28)")


REGISTRY["absZoo"] = abs_zoo
