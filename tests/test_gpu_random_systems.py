"""GPU: randomly generated systems (random expression DAGs over the tape vocabulary) through the
whole product path -- tracer -> tape -> codegen -> hiprtc -> kernels -- against the CPU oracle.
Guards the generic machinery (AD rules of every jet type, the reverse sweep, codegen corner cases
such as shared subexpressions, constants, outputs that are inputs, unused inputs) beyond the
hand-written example systems."""
import numpy as np
import pytest

from hamilton_amd import examples as E
from hamilton_amd import tracer as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(hamk_lib):
    from hamilton_amd import api as _api
    if hamk_lib.hamk_device_count() < 1:
        pytest.fail("no HIP device visible")
    return _api


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(1.0, np.abs(np.asarray(b)))))


def random_expr(rng, leaves, depth, o):
    """A smooth, domain-safe random expression over `leaves` (bounded inputs in [-1, 1])."""
    if depth == 0 or rng.random() < 0.15:
        v = leaves[rng.integers(len(leaves))]
        return v if rng.random() < 0.8 else v * float(np.round(rng.uniform(-1.5, 1.5), 2))
    a = random_expr(rng, leaves, depth - 1, o)
    kind = rng.integers(14)
    if kind <= 3:
        b = random_expr(rng, leaves, depth - 1, o)
        return [a + b, a - b, a * b, a / (2.0 + b * b)][kind]
    if kind == 4: return o.sin(a)
    if kind == 5: return o.cos(a)
    if kind == 6: return o.tanh(a)
    if kind == 7: return o.exp(0.3 * a)
    if kind == 8: return o.sqrt(1.5 + a * a)
    if kind == 9: return o.log(2.0 + a * a)
    if kind == 10: return o.atan(a)
    if kind == 11: return a ** int(rng.integers(2, 4))
    if kind == 12: return -a
    return (2.0 + a * a) ** float(np.round(rng.uniform(-1.5, 1.5), 2))


def random_spec(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 5))
    m = n + int(rng.integers(0, 3))
    prog_seed = int(rng.integers(1 << 30))
    u_cart = bool(rng.integers(2))
    inertia = tuple(float(np.round(rng.uniform(0.5, 2.0), 2)) for _ in range(m))

    def f(q, o):
        r = np.random.default_rng(prog_seed)
        out = []
        for k in range(m):
            g = random_expr(r, list(q), 3, o)
            out.append(q[k % n] * (1.0 if k < n else 0.5) + 0.25 * g)     # J = I-ish + perturbation: full rank
        return out

    def u(z, o):
        r = np.random.default_rng(prog_seed + 1)
        return random_expr(r, list(z), 3, o) + 0.5 * z[0] * z[0]

    box = tuple((-1.0, 1.0) for _ in range(n))
    return E.SystemSpec(name=f"random{seed}", m=m, n=n, inertia=inertia, f=f, u=u,
                        u_space=E.U_CARTESIAN if u_cart else E.U_GENERALIZED,
                        q0=(0.1,) * n, qd0=(0.2,) * n, q_box=box, qd_box=box, dt=0.005, cite="random")


def poly_trig_spec(seed):
    """Random TRIGONOMETRIC-POLYNOMIAL coordinate maps with the structure that makes a mass matrix simplify (round 6): planar bodies at
    (base + rho cos phi, base + rho sin phi), rho in {constant, q_i, 1 + q_i / 2}, phi an integer-ish combination of inputs, each body
    hanging from the origin or from the previous body -- random pendulums, polar particles and mixtures, optionally with a free
    cartesian coordinate.  The class of maps for which the generator derives K and dT/dq SYMBOLICALLY (hamk_codegen.cpp
    symbolic_mass_matrix: sin^2 + cos^2 = 1); the random expression DAGs above never are (exp, tanh, sqrt ...)."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 5))
    bodies = int(rng.integers(1, 4))
    extra = int(rng.integers(0, 2))
    m = 2 * bodies + extra
    prog_seed = int(rng.integers(1 << 30))
    u_cart = bool(rng.integers(2))
    inertia = tuple(float(np.round(rng.uniform(0.5, 2.0), 2)) for _ in range(m))

    def f(q, o):
        r = np.random.default_rng(prog_seed)
        out, bx, by = [], 0.0, 0.0
        used = set()
        for b in range(bodies):
            i = b % n                                              # every coordinate drives something (full-rank J)
            used.add(i)
            kind = int(r.integers(3))
            j = int(r.integers(n))
            phi = q[i] if kind == 0 else (q[i] + float(r.choice([1.0, -1.0, 2.0])) * q[j] if (kind == 1 and j != i) else float(r.choice([1.0, 2.0])) * q[i])
            rk = int(r.integers(3))
            l = int(r.integers(n))
            rho = float(np.round(r.uniform(0.5, 1.5), 2)) if (rk == 0 or l == i) else (1.5 + q[l] if rk == 1 else 1.0 + 0.5 * q[l])
            if not isinstance(rho, float):
                used.add(l)
            x, y = bx + rho * o.cos(phi), by + rho * o.sin(phi)
            out += [x, y]
            if r.random() < 0.5:
                bx, by = x, y                                      # the next body hangs from this one
        for _ in range(extra):
            out.append(sum(q[k] for k in range(n)) * 0.7)
        for k in range(n):                                         # coordinates nothing drives yet: a direct cartesian component
            if k not in used and extra == 0:
                out[-1] = out[-1] + 1.3 * q[k]
        return out

    def u(z, o):
        acc = 0.0
        for k, zk in enumerate(z):
            acc = acc + (0.5 + 0.1 * k) * zk * zk
        return acc + 0.2 * o.cos(z[0])

    box = tuple((-1.0, 1.0) for _ in range(n))
    return E.SystemSpec(name=f"polytrig{seed}", m=m, n=n, inertia=inertia, f=f, u=u,
                        u_space=E.U_CARTESIAN if u_cart else E.U_GENERALIZED,
                        q0=(0.1,) * n, qd0=(0.2,) * n, q_box=box, qd_box=box, dt=0.005, cite="random (trigonometric polynomial)")


POLY_TRIG_SEEDS = [0, 2, 3, 5, 6, 7, 8, 11, 13, 14, 15]      # (the seeds whose random map has full rank: median cond K < 100)


@pytest.mark.parametrize("seed", POLY_TRIG_SEEDS)
def test_random_trigonometric_polynomial_systems_vs_oracle(api, oracle_lib, seed):
    """The symbolic mass matrix on maps nobody hand-checked: hamEqs, velocities, energies, RK4 and stepHam of random
    trigonometric-polynomial systems against the oracle (which differentiates the tape numerically-exactly and knows no algebra)."""
    spec = poly_trig_spec(seed)
    s = api.system_from_spec(spec)
    o = oracle_lib.OracleSystem(spec)
    B = 257
    q, qd = E.sample_config(spec, 5, B)
    p = o.to_phase_batch(q, qd)
    dq, dp = api.hamEqs(s, api.Phase(q, p))
    odq, odp, ost = o.hameqs_batch(q, p)
    ok = (ost == 0) & (np.asarray(s.last_status) == 0)
    assert ok.mean() > 0.9
    cond = np.array([np.linalg.cond(o.jacobian(q[:, i]).T @ np.diag(spec.inertia) @ o.jacobian(q[:, i])) for i in range(B)])
    ok &= cond < 1e6
    assert relerr(np.asarray(dq)[:, ok], odq[:, ok]) < 1e-9 and relerr(np.asarray(dp)[:, ok], odp[:, ok]) < 1e-9, (seed, "HAS_SYM_K = true" in s.source)
    assert relerr(np.asarray(api.hamiltonian(s, api.Phase(q, p)))[ok], o.observe_batch(q, p)[2][ok]) < 1e-9
    ph = api.rk4Steps(spec.dt, 4, s, api.Phase(q, p))
    oq, op = o.rk4_steps_batch(q, p, spec.dt, 4)
    # a trajectory that passes a near-singular K on its way amplifies roundoff without bound -- in the oracle as much as in the kernel
    # (polytrig15, lane 58: cond K 1e4 -> 6e7 -> overflow within four steps): only lanes whose ORACLE path stays well-conditioned compare
    wq, wp = q, p
    for _ in range(4):
        wq, wp = o.rk4_steps_batch(wq, wp, spec.dt, 1)
        fin = np.isfinite(wq).all(0) & np.isfinite(wp).all(0)
        c = np.array([np.linalg.cond(o.jacobian(wq[:, i]).T @ np.diag(spec.inertia) @ o.jacobian(wq[:, i])) if fin[i] else np.inf for i in range(B)])
        ok &= fin & (c < 1e6)
    assert ok.mean() > 0.8
    assert relerr(np.asarray(ph.positions)[:, ok], oq[:, ok]) < 1e-9 and relerr(np.asarray(ph.momenta)[:, ok], op[:, ok]) < 1e-9


@pytest.mark.parametrize("seed", range(16))
def test_random_system_vs_oracle(api, oracle_lib, seed):
    spec = random_spec(seed)
    s = api.system_from_spec(spec)
    o = oracle_lib.OracleSystem(spec)
    B = 130
    q, qd = E.sample_config(spec, 99, B)
    p = o.to_phase_batch(q, qd)
    assert relerr(api.momenta(s, api.Config(q, qd)), p) < 1e-12
    odq, odp, ost = o.hameqs_batch(q, p)
    good = ost == 0
    dq, dp = api.hamEqs(s, api.Phase(q, p))
    cond = np.array([np.linalg.cond(o.jacobian(q[:, i]).T @ np.diag(spec.inertia) @ o.jacobian(q[:, i])) for i in range(B)])
    tol = 1e-11 * np.maximum(1.0, cond / 100.0)
    err = np.maximum(np.abs(dq - odq).max(0) / np.maximum(1.0, np.abs(odq).max(0)),
                     np.abs(dp - odp).max(0) / np.maximum(1.0, np.abs(odp).max(0)))
    assert np.all(err[good] <= tol[good]), (seed, float(np.max(err / tol)))
    ke, pe_, h = o.observe_batch(q, p)
    assert relerr(api.pe(s, q), pe_) < 1e-12
    herr = np.abs(api.hamiltonian(s, api.Phase(q, p)) - h) / np.maximum(1.0, np.abs(h))
    assert np.all(herr <= tol), (seed, float(np.max(herr / tol)))
    ph = api.rk4Steps(spec.dt, 2, s, api.Phase(q, p))
    oq, op = o.rk4_steps_batch(q, p, spec.dt, 2)
    e2 = np.maximum(np.abs(ph.positions - oq).max(0), np.abs(ph.momenta - op).max(0)) / np.maximum(1.0, np.abs(op).max(0))
    assert np.all(e2 <= 10 * tol), (seed, float(np.max(e2 / tol)))
    st = api.stepHam(0.02, s, api.Phase(q, p))
    sq, sp, sns = o.step_ham_batch(q, p, 0.02)
    same = np.asarray(s.last_nsub) == sns
    assert same.mean() > 0.9
    e3 = np.maximum(np.abs(st.positions - sq).max(0), np.abs(st.momenta - sp).max(0)) / np.maximum(1.0, np.abs(sp).max(0))
    assert np.all(e3[same] <= 100 * tol[same]), (seed, float(np.max(e3[same] / tol[same])))


@pytest.mark.parametrize("variant", ["R", "wave", "quad"])
@pytest.mark.parametrize("seed", [0, 1, 4, 7, 12, 15])
def test_random_system_other_code_paths(api, oracle_lib, monkeypatch, seed, variant):
    """The same random systems through the reverse-sweep variant (MODE_R), through the wave-cooperative kernels and
    through the four-lanes-per-trajectory kernels (both forced on small n, through the ABI's options)."""
    from hamilton_amd import _abi
    spec = random_spec(seed)
    opt = {"R": {"ad_mode": _abi.AD_R}, "wave": {"mapping": _abi.MAP_WAVE}, "quad": {"mapping": _abi.MAP_QUAD}}[variant]
    s = api.system_from_spec(spec, opt)
    marker = {"R": "MODE_R = true", "wave": "HAMK_INSTANTIATE_WAVE", "quad": "HAMK_INSTANTIATE_QUAD"}[variant]
    assert marker in s.source
    o = oracle_lib.OracleSystem(spec)
    B = 70
    q, qd = E.sample_config(spec, 5, B)
    p = o.to_phase_batch(q, qd)
    odq, odp, _ = o.hameqs_batch(q, p)
    dq, dp = api.hamEqs(s, api.Phase(q, p))
    cond = np.array([np.linalg.cond(o.jacobian(q[:, i]).T @ np.diag(spec.inertia) @ o.jacobian(q[:, i])) for i in range(B)])
    tol = 1e-11 * np.maximum(1.0, cond / 100.0)
    err = np.maximum(np.abs(dq - odq).max(0) / np.maximum(1.0, np.abs(odq).max(0)),
                     np.abs(dp - odp).max(0) / np.maximum(1.0, np.abs(odp).max(0)))
    assert np.all(err <= tol), (seed, variant, float(np.max(err / tol)))
    ph = api.rk4Steps(spec.dt, 2, s, api.Phase(q, p))
    oq, op = o.rk4_steps_batch(q, p, spec.dt, 2)
    e2 = np.maximum(np.abs(ph.positions - oq).max(0), np.abs(ph.momenta - op).max(0)) / np.maximum(1.0, np.abs(op).max(0))
    assert np.all(e2 <= 10 * tol), (seed, variant, float(np.max(e2 / tol)))
