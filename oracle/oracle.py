"""ctypes front-end of oracle/libhamk_oracle.so (plain-C restatement of the reference).

TEST INFRASTRUCTURE -- never imported by the product path (hamilton_amd/).
PARITY UNPINNED (no runnable reference, no reference golden vectors): pinned
instead against tests/golden/*.json from oracle/gen_golden.py.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhamk_oracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_lp = ctypes.POINTER(ctypes.c_long)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "hamk_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libhamk_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_system_create.restype = ctypes.c_void_p
        _lib.orc_pe.restype = ctypes.c_double
        _lib.orc_keC.restype = ctypes.c_double
        _lib.orc_keP.restype = ctypes.c_double
        _lib.orc_lagrangian.restype = ctypes.c_double
        _lib.orc_hamiltonian.restype = ctypes.c_double
        _lib.orc_system_get_gsl_api.restype = ctypes.c_int
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _vec(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


class OracleSystem:
    """One `System m n` (Hamilton.hs:160-169) held by the C oracle."""

    def __init__(self, spec):
        self.spec = spec
        self.m, self.n = spec.m, spec.n
        tf, tu = spec.trace()
        f_ops, f_n, f_outs = tf.as_ctypes()
        u_ops, u_n, u_outs = tu.as_ctypes()
        inertia = _vec(spec.inertia)
        L = lib()
        self._h = ctypes.c_void_p(L.orc_system_create(
            ctypes.c_int(spec.m), ctypes.c_int(spec.n), _d(inertia),
            f_ops, ctypes.c_int(f_n), f_outs,
            u_ops, ctypes.c_int(u_n), ctypes.c_int32(tu.outs[0]), ctypes.c_int(spec.u_space)))

    def __del__(self):
        try:
            if self._h:
                lib().orc_system_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- single trajectory, reference-shaped -------------------------------------
    def coords(self, q):
        q = _vec(q); x = np.empty(self.m)
        lib().orc_coords(self._h, _d(q), _d(x)); return x

    def jacobian(self, q):
        q = _vec(q); J = np.empty((self.m, self.n))
        lib().orc_jacobian(self._h, _d(q), _d(J)); return J

    def hessian(self, q):
        q = _vec(q); H = np.empty((self.n, self.m, self.n))
        lib().orc_hessian(self._h, _d(q), _d(H)); return H

    def pe(self, q):
        q = _vec(q); return lib().orc_pe(self._h, _d(q))

    def grad_pe(self, q):
        q = _vec(q); g = np.empty(self.n)
        lib().orc_grad_pe(self._h, _d(q), _d(g)); return g

    def momenta(self, q, qd):
        q, qd = _vec(q), _vec(qd); p = np.empty(self.n)
        lib().orc_momenta(self._h, _d(q), _d(qd), _d(p)); return p

    def velocities(self, q, p):
        q, p = _vec(q), _vec(p); v = np.empty(self.n)
        lib().orc_velocities(self._h, _d(q), _d(p), _d(v)); return v

    def keC(self, q, qd):
        q, qd = _vec(q), _vec(qd); return lib().orc_keC(self._h, _d(q), _d(qd))

    def keP(self, q, p):
        q, p = _vec(q), _vec(p); return lib().orc_keP(self._h, _d(q), _d(p))

    def lagrangian(self, q, qd):
        q, qd = _vec(q), _vec(qd); return lib().orc_lagrangian(self._h, _d(q), _d(qd))

    def hamiltonian(self, q, p):
        q, p = _vec(q), _vec(p); return lib().orc_hamiltonian(self._h, _d(q), _d(p))

    def hameqs(self, q, p):
        q, p = _vec(q), _vec(p); dq = np.empty(self.n); dp = np.empty(self.n)
        lib().orc_hameqs(self._h, _d(q), _d(p), _d(dq), _d(dp)); return dq, dp

    def rk4_steps(self, q, p, dt, nsteps):
        q, p = _vec(q).copy(), _vec(p).copy()
        lib().orc_rk4_steps(self._h, _d(q), _d(p), ctypes.c_double(dt), ctypes.c_int(nsteps)); return q, p

    def step_ham(self, dt, q, p, counts: Optional[list] = None):
        q, p = _vec(q).copy(), _vec(p).copy()
        c = (ctypes.c_long * 4)()
        lib().orc_step_ham(self._h, ctypes.c_double(dt), _d(q), _d(p), c)
        if counts is not None:
            counts[:] = list(c)
        return q, p

    def evolve_ham(self, q0, p0, ts, h0=0.0, eps_abs=0.0, eps_rel=0.0, counts: Optional[list] = None):
        q0, p0, ts = _vec(q0), _vec(p0), _vec(ts)
        out = np.empty((len(ts), 2 * self.n))
        c = (ctypes.c_long * 4)()
        lib().orc_evolve_ham(self._h, _d(q0), _d(p0), ctypes.c_int(len(ts)), _d(ts), _d(out),
                             ctypes.c_double(h0), ctypes.c_double(eps_abs), ctypes.c_double(eps_rel), c)
        if counts is not None:
            counts[:] = list(c)
        return out[:, :self.n].copy(), out[:, self.n:].copy()

    # ---- SoA ensembles [n][B] ---------------------------------------------------------
    def hameqs_batch(self, q, p, threads=0):
        q, p = _vec(q), _vec(p); B = q.shape[1]
        dq, dp = np.empty_like(q), np.empty_like(p); st = np.zeros(B, dtype=np.int32)
        lib().orc_hameqs_batch(self._h, ctypes.c_long(B), _d(q), _d(p), _d(dq), _d(dp),
                               st.ctypes.data_as(_ip), ctypes.c_int(threads))
        return dq, dp, st

    def to_phase_batch(self, q, qd, threads=0):
        q, qd = _vec(q), _vec(qd); B = q.shape[1]; p = np.empty_like(q)
        lib().orc_to_phase_batch(self._h, ctypes.c_long(B), _d(q), _d(qd), _d(p), ctypes.c_int(threads))
        return p

    def from_phase_batch(self, q, p, threads=0):
        q, p = _vec(q), _vec(p); B = q.shape[1]; qd = np.empty_like(q); st = np.zeros(B, dtype=np.int32)
        lib().orc_from_phase_batch(self._h, ctypes.c_long(B), _d(q), _d(p), _d(qd),
                                   st.ctypes.data_as(_ip), ctypes.c_int(threads))
        return qd, st

    def observe_batch(self, q, p, threads=0):
        q, p = _vec(q), _vec(p); B = q.shape[1]
        ke, pe, h = np.empty(B), np.empty(B), np.empty(B)
        lib().orc_observe_batch(self._h, ctypes.c_long(B), _d(q), _d(p), _d(ke), _d(pe), _d(h), ctypes.c_int(threads))
        return ke, pe, h

    def observe_config_batch(self, q, qd, threads=0):
        q, qd = _vec(q), _vec(qd); B = q.shape[1]
        ke, lag = np.empty(B), np.empty(B)
        lib().orc_observe_config_batch(self._h, ctypes.c_long(B), _d(q), _d(qd), _d(ke), _d(lag), ctypes.c_int(threads))
        return ke, lag

    def coords_batch(self, q, threads=0):
        q = _vec(q); B = q.shape[1]; x = np.empty((self.m, B))
        lib().orc_coords_batch(self._h, ctypes.c_long(B), _d(q), _d(x), ctypes.c_int(threads))
        return x

    def rk4_steps_batch(self, q, p, dt, nsteps, threads=0):
        q, p = _vec(q).copy(), _vec(p).copy(); B = q.shape[1]
        lib().orc_rk4_steps_batch(self._h, ctypes.c_long(B), _d(q), _d(p), ctypes.c_double(dt),
                                  ctypes.c_int(nsteps), ctypes.c_int(threads))
        return q, p

    def step_ham_batch(self, q, p, dt, threads=0):
        q, p = _vec(q).copy(), _vec(p).copy(); B = q.shape[1]; ns = np.zeros(B, dtype=np.int32)
        self.last_fail = np.zeros(B, dtype=np.int32)
        lib().orc_step_ham_batch(self._h, ctypes.c_long(B), _d(q), _d(p), ctypes.c_double(dt),
                                 ns.ctypes.data_as(_ip), self.last_fail.ctypes.data_as(_ip), ctypes.c_int(threads))
        return q, p, ns

    def evolve_ham_batch(self, q0, p0, ts, h0=0.0, eps_abs=0.0, eps_rel=0.0, threads=0):
        q0, p0, ts = _vec(q0), _vec(p0), _vec(ts); B = q0.shape[1]; nt = len(ts)
        qo, po = np.empty((nt, self.n, B)), np.empty((nt, self.n, B)); ns = np.zeros(B, dtype=np.int32)
        self.last_fail = np.zeros(B, dtype=np.int32)
        lib().orc_evolve_ham_batch(self._h, ctypes.c_long(B), _d(q0), _d(p0), ctypes.c_int(nt), _d(ts),
                                   _d(qo), _d(po), ctypes.c_double(h0), ctypes.c_double(eps_abs),
                                   ctypes.c_double(eps_rel), ns.ctypes.data_as(_ip),
                                   self.last_fail.ctypes.data_as(_ip), ctypes.c_int(threads))
        return qo, po, ns

    # ---- which GSL binding of hmatrix-gsl's gsl-ode.c the adaptive stepper follows --------
    @property
    def gsl_api(self) -> int:
        return int(lib().orc_system_get_gsl_api(self._h))

    @gsl_api.setter
    def gsl_api(self, api: int):
        lib().orc_system_set_gsl_api(self._h, ctypes.c_int(int(api)))

    def evolve_ham_trace(self, q0, p0, ts, h0=0.0, eps_abs=0.0, eps_rel=0.0, cap=100000):
        """One trajectory with the per-attempt trace [(t reached, h tried, 1 accepted / 0 rejected /
        -1 api-2 failure), ...]; returns (q rows, p rows, counts, trace)."""
        q0, p0, ts = _vec(q0), _vec(p0), _vec(ts)
        out = np.empty((len(ts), 2 * self.n))
        c = (ctypes.c_long * 4)()
        tr = np.zeros((cap, 3))
        lib().orc_evolve_ham_trace(self._h, _d(q0), _d(p0), ctypes.c_int(len(ts)), _d(ts), _d(out),
                                   ctypes.c_double(h0), ctypes.c_double(eps_abs), ctypes.c_double(eps_rel), c,
                                   _d(tr), ctypes.c_long(cap))
        k = int(c[1] + c[2] + (1 if c[3] == 1 else 0))
        return out[:, :self.n].copy(), out[:, self.n:].copy(), list(c), tr[:min(k, cap)].copy()


def max_threads() -> int:
    return lib().orc_max_threads()
