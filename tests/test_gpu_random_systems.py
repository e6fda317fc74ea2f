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
