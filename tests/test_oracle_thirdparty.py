"""CPU: hold the oracle's "restated from memory" legs against the third-party library CLASSES the
reference really calls.

The reference cannot be built here (Haskell + GSL), and it holds no golden vectors (test/Spec.hs:1-2),
so parity stays unpinned.  Two of the three third-party legs of the path can still be checked against the
actual library rather than against a recollection of it:

  * hmatrix `inv` (Hamilton.hs:321, :381) is LAPACK dgesv against the identity; `numpy.linalg.inv` binds
    exactly that routine.  oracle/hamk_oracle.c `lu_inverse` (unblocked partial-pivoting LU) is compared
    with it on K = J^T M J of every golden point, on an ill-conditioned twoBody point (r -> 0), on the
    chains, and on matrices that force row exchanges -- to the backward-error bound of Gaussian
    elimination, cond(K) * eps * small constant;
  * `ad`'s jacobianT / hessianF / grad (Hamilton.hs:221-224) are an operator-overloading AD engine;
    `torch.autograd.functional.jacobian / hessian` in fp64 is an independent engine of that kind.  The
    oracle's second-order tape interpreter (J, the Hessian tensor in the reference's `dJ/dq_i` layout,
    grad U incl. the composition u . f of mkSystem') is compared with it on every example system.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import ALL_GOLDEN_SYSTEMS, fvec, load_golden
from hamilton_amd import examples as E

EPS = np.finfo(np.float64).eps


def orc_inverse(oracle_lib, A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    out = np.empty_like(A)
    dp = ctypes.POINTER(ctypes.c_double)
    info = oracle_lib.lib().orc_inverse(ctypes.c_int(n), A.ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out, info


def inv_bound(A, Ainv_ref):
    """|Ainv_lu - Ainv| <= c n eps cond(A) |Ainv| for two backward-stable inversions of the same matrix."""
    n = A.shape[0]
    cond = np.linalg.cond(A)
    return 8.0 * n * EPS * cond * np.max(np.abs(Ainv_ref)) + 1e-300


def mass_matrix(o, spec, q):
    J = o.jacobian(q)
    return J.T @ np.diag(np.asarray(spec.inertia, dtype=np.float64)) @ J


@pytest.mark.parametrize("name", ALL_GOLDEN_SYSTEMS)
def test_inv_is_lapack_dgesv_on_golden_mass_matrices(oracle_lib, name):
    spec = E.get(name)
    o = oracle_lib.OracleSystem(spec)
    for pt in load_golden(name)["points"]:
        q, p = fvec(pt["q"]), fvec(pt["p"])
        K = mass_matrix(o, spec, q)
        ref = np.linalg.inv(K)                                  # LAPACK dgesv(K, I): what hmatrix `inv` binds
        got, info = orc_inverse(oracle_lib, K)
        assert info == 0
        assert np.max(np.abs(got - ref)) <= inv_bound(K, ref)
        # ... and the call site: velocities = inv(K) #> p, Hamilton.hs:316-324
        v_ref = ref @ p
        assert np.max(np.abs(o.velocities(q, p) - v_ref)) <= inv_bound(K, ref) * max(1.0, np.max(np.abs(p))) * K.shape[0]


def test_inv_on_an_ill_conditioned_two_body_point(oracle_lib):
    """twoBody as r -> 0: K = diag(mu, mu r^2), cond = 1/r^2 (SURVEY hard parts: singular / ill-conditioned K)."""
    spec = E.get("twoBody")
    o = oracle_lib.OracleSystem(spec)
    for r in (1e-3, 1e-5, 3e-7):
        q = np.array([r, 0.7])
        K = mass_matrix(o, spec, q)
        assert np.linalg.cond(K) > 0.5 / r ** 2
        ref = np.linalg.inv(K)
        got, info = orc_inverse(oracle_lib, K)
        assert info == 0
        assert np.max(np.abs(got - ref)) <= inv_bound(K, ref)


@pytest.mark.parametrize("name", ["chain8", "chain16", "chain32"])
def test_inv_on_chain_mass_matrices(oracle_lib, name):
    spec = E.get(name)
    o = oracle_lib.OracleSystem(spec)
    q, _ = E.sample_config(spec, 0, 4)
    for i in range(4):
        K = mass_matrix(o, spec, q[:, i])
        ref = np.linalg.inv(K)
        got, info = orc_inverse(oracle_lib, K)
        assert info == 0
        assert np.max(np.abs(got - ref)) <= inv_bound(K, ref)


def test_inv_pivots_like_lapack(oracle_lib):
    """General (non-symmetric, pivoting) matrices, and the exactly singular case: dgesv's info > 0 is hmatrix's
    exception (Hamilton.hs:321,381); numpy raises LinAlgError on the same input."""
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 6, 17, 32):
        for _ in range(6):
            A = rng.standard_normal((n, n))
            A[0, 0] = 1e-14 * A[0, 0]                           # partial pivoting must move row 0 away
            ref = np.linalg.inv(A)
            got, info = orc_inverse(oracle_lib, A)
            assert info == 0
            assert np.max(np.abs(got - ref)) <= inv_bound(A, ref)
    S = np.array([[1.0, 2.0], [2.0, 4.0]])
    with pytest.raises(np.linalg.LinAlgError):
        np.linalg.inv(S)
    got, info = orc_inverse(oracle_lib, S)
    assert info == 1 and np.all(np.isnan(got))


# ---- AD: the oracle's tape interpreter vs torch.autograd (fp64) -------------------------------------------------

class _TorchOps:
    """The vocabulary hamilton_amd.examples uses (sin, cos, exp, ...), on fp64 torch scalars."""

    def __getattr__(self, name):
        if name == "signum":
            return lambda x: torch.sign(torch.as_tensor(x, dtype=torch.float64))
        fn = getattr(torch, name)
        if name == "atan2":
            return lambda y, x: fn(torch.as_tensor(y, dtype=torch.float64), torch.as_tensor(x, dtype=torch.float64))
        return lambda x: fn(torch.as_tensor(x, dtype=torch.float64))


_TO = _TorchOps()
AD_SYSTEMS = ALL_GOLDEN_SYSTEMS + ["absZoo", "chain8"]


def _as_t(v):
    return v if isinstance(v, torch.Tensor) else torch.tensor(float(v), dtype=torch.float64)


@pytest.mark.parametrize("name", AD_SYSTEMS)
def test_jacobian_hessian_gradient_match_torch_autograd(oracle_lib, name):
    spec = E.get(name)
    o = oracle_lib.OracleSystem(spec)

    def coords(qt):
        return torch.stack([_as_t(v) for v in spec.coords(list(qt), _TO)])

    def potential(qt):
        return _as_t(spec.potential_of_q(list(qt), _TO))

    qs, _ = E.sample_config(spec, 0, 5)
    for i in range(5):
        q = qs[:, i].copy()
        qt = torch.tensor(q, dtype=torch.float64)
        J = torch.autograd.functional.jacobian(coords, qt).numpy()                         # jacobianT, Hamilton.hs:221
        scale = max(1.0, float(np.max(np.abs(J))))
        assert np.max(np.abs(o.jacobian(q) - J)) <= 1e-13 * scale
        # hessianF (:222) in the layout `tr2 . fmap vec2l` gives it (:227-233): H[i] = dJ/dq_i, i.e.
        # H[i][k][j] = d^2 x_k / dq_i dq_j
        H = np.stack([torch.autograd.functional.hessian(lambda z, k=k: coords(z)[k], qt).numpy() for k in range(spec.m)])
        H_ref = np.transpose(H, (1, 0, 2))
        hs = max(1.0, float(np.max(np.abs(H_ref))))
        assert np.max(np.abs(o.hessian(q) - H_ref)) <= 1e-12 * hs
        g = torch.autograd.functional.jacobian(potential, qt).numpy()                      # grad, :224 (u . f for mkSystem', :254)
        assert np.max(np.abs(o.grad_pe(q) - g)) <= 1e-12 * max(1.0, float(np.max(np.abs(g))))
        assert abs(o.pe(q) - float(potential(qt))) <= 1e-13 * max(1.0, abs(o.pe(q)))


@pytest.mark.parametrize("name", ["doublePendulum", "spring", "threeBodyPolar", "chain8", "chain16", "chain32", "chain12~mixed"])
def test_hameqs_from_torch_autograd_of_the_hamiltonian(oracle_lib, name):
    """(dq, dp) = (dH/dp, -dH/dq) with H = 1/2 p K^-1 p + U assembled in torch and differentiated by autograd --
    no hamEqs algebra (Hamilton.hs:375-387) involved -- against the oracle's literal restatement of it."""
    spec = E.get(name)
    o = oracle_lib.OracleSystem(spec)
    m = torch.tensor(spec.inertia, dtype=torch.float64)

    def H(y):
        q, p = y[:spec.n], y[spec.n:]
        J = torch.autograd.functional.jacobian(
            lambda z: torch.stack([_as_t(v) for v in spec.coords(list(z), _TO)]), q, create_graph=True)
        K = J.T @ torch.diag(m) @ J
        return 0.5 * p @ torch.linalg.solve(K, p) + _as_t(spec.potential_of_q(list(q), _TO))

    qs, qds = E.sample_config(spec, 0, 3)
    if "chain" in name:                                       # the chains' box is at rest: p = 0 would make K^-1 p trivial
        qds = 0.3 * np.cos(0.7 * np.arange(spec.n * 3, dtype=np.float64).reshape(spec.n, 3))
    for i in range(3):
        q = qs[:, i].copy()
        p = o.momenta(q, qds[:, i].copy())
        y = torch.tensor(np.concatenate([q, p]), dtype=torch.float64)
        g = torch.autograd.functional.jacobian(H, y).numpy()
        dq, dp = o.hameqs(q, p)
        Jq = o.jacobian(q)
        cond = np.linalg.cond(Jq.T @ np.diag(spec.inertia) @ Jq)
        s = max(1.0, float(np.max(np.abs(g)))) * max(1.0, cond / 1e3)
        assert np.max(np.abs(dq - g[spec.n:])) <= 1e-11 * s
        assert np.max(np.abs(dp + g[:spec.n])) <= 1e-11 * s
