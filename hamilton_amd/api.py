"""Host-side mirror of `Numeric.Hamilton` over the C ABI of libhamk.so.

Same names, argument order and meaning as the reference's export list
(/root/reference/src/Numeric/Hamilton.hs:28-70); the one semantic extension is
that every state may be an ENSEMBLE: positions/velocities/momenta are arrays of
shape [n] (one trajectory, the reference's case) or [n, B] (B independent
trajectories, structure-of-arrays).  numpy arrays take the host-staged path
(HAMK_MEM_HOST); torch CUDA tensors are used in place on the GPU
(HAMK_MEM_DEVICE, launched on torch's current stream).

Python cannot spell a prime: `mkSystem'` is `mkSystem_`, `evolveHam'` is
`evolveHam_`, `evolveHamC'` is `evolveHamC_`.

Error behaviour: API misuse / toolchain / HIP failures raise `HamkError` (the
reference raises Haskell exceptions: Hamilton.hs:425,444,462).  A singular mass
matrix does NOT raise for ensembles: the per-trajectory status word is kept in
`System.last_status` (bit HAMK_ST_SINGULAR) and the affected lanes hold NaN --
for a single trajectory ([n]-shaped input) `SingularSystem` is raised, matching
hmatrix's exception out of `inv` (Hamilton.hs:321,381).
"""
from __future__ import annotations

import ctypes
from typing import Callable, List, Optional, Sequence
import contextlib

import numpy as np

from . import _abi
from . import tracer as T
from ._abi import HamkError, MEM_DEVICE, MEM_HOST, ST_DRIFT, ST_SINGULAR

try:  # torch is plumbing (device memory, streams); the package works without it on host arrays
    import torch
except Exception:  # pragma: no cover
    torch = None

U_GENERALIZED, U_CARTESIAN = 0, 1


class SingularSystem(ArithmeticError):
    """K = J^T M J not invertible for a single-trajectory call."""


# ---------------------------------------------------------------------------------------
# array plumbing
# ---------------------------------------------------------------------------------------
def _is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


class _Arr:
    """A [rows, B] fp64 array on host (numpy) or device (torch.cuda)."""

    def __init__(self, data, rows: int, name: str):
        self.single = False
        if _is_torch(data) and data.is_cuda:
            t = data
            if t.dtype != torch.float64:
                t = t.to(torch.float64)
            if t.dim() == 1:
                t = t.reshape(rows, 1)
                self.single = True
            if t.dim() != 2 or t.shape[0] != rows:
                raise ValueError(f"{name}: expected shape [{rows}] or [{rows}, B], got {tuple(data.shape)}")
            self.a = t.contiguous()
            self.device = True
        else:
            a = data.detach().cpu().numpy() if _is_torch(data) else np.asarray(data, dtype=np.float64)
            if a.ndim == 1:
                a = a.reshape(rows, 1)
                self.single = True
            if a.ndim != 2 or a.shape[0] != rows:
                raise ValueError(f"{name}: expected shape [{rows}] or [{rows}, B], got {np.shape(data)}")
            self.a = np.ascontiguousarray(a, dtype=np.float64)
            self.device = False
        self.B = int(self.a.shape[1])

    @property
    def ptr(self):
        return self.a.data_ptr() if self.device else self.a.ctypes.data

    @property
    def mem(self):
        return MEM_DEVICE if self.device else MEM_HOST

    def like(self, rows: Optional[int] = None, dtype="f8", lead: Sequence[int] = ()):
        shape = tuple(lead) + ((rows, self.B) if rows is not None else (self.B,))
        if self.device:
            dt = torch.float64 if dtype == "f8" else torch.int32
            return torch.empty(shape, dtype=dt, device=self.a.device)
        return np.empty(shape, dtype=np.float64 if dtype == "f8" else np.int32)

    def clone(self):
        return self.a.clone() if self.device else self.a.copy()


def _pair(a, name_a: str, b, name_b: str, rows: int):
    """Two arrays of one state (positions + velocities / momenta): same ensemble size, same place."""
    xa, xb = _Arr(a, rows, name_a), _Arr(b, rows, name_b)
    if xa.B != xb.B or xa.single != xb.single:
        raise ValueError(f"{name_a} and {name_b} describe different ensembles ({xa.B} vs {xb.B} trajectories)")
    if xa.device != xb.device or (xa.device and xa.a.device != xb.a.device):
        raise ValueError(f"{name_a} and {name_b} must live in the same memory (both host arrays or both tensors of one GPU)")
    return xa, xb


def _in_place(x: "_Arr", original, name: str):
    """inplace=True only makes sense if the array handed to the library IS the caller's array."""
    same = (x.a.data_ptr() == original.data_ptr()) if (x.device and _is_torch(original)) else \
        (isinstance(original, np.ndarray) and x.a.ctypes.data == original.ctypes.data)
    if not same:
        raise ValueError(f"inplace=True needs {name} as a contiguous float64 [n, B] array (a converted copy would be advanced instead)")


def _ptr(x):
    if x is None:
        return None
    return x.data_ptr() if _is_torch(x) else x.ctypes.data


def _shape_out(x, single: bool):
    """[rows, 1] -> [rows] (and [1] -> scalar) when the caller passed one trajectory."""
    if not single:
        return x
    if x.ndim == 1:
        return float(x[0])
    return x[..., 0]


# ---------------------------------------------------------------------------------------
# Systems and states                                              Hamilton.hs:103-169
# ---------------------------------------------------------------------------------------
class Config:
    """`Cfg { cfgPositions, cfgVelocities }` (Hamilton.hs:103-115)."""

    def __init__(self, positions, velocities):
        self.positions = positions
        self.velocities = velocities

    cfgPositions = property(lambda self: self.positions)
    cfgVelocities = property(lambda self: self.velocities)

    def __repr__(self):
        return f"Cfg {{cfgPositions = {self.positions!r}, cfgVelocities = {self.velocities!r}}}"


class Phase:
    """`Phs { phsPositions, phsMomenta }` (Hamilton.hs:133-145)."""

    def __init__(self, positions, momenta):
        self.positions = positions
        self.momenta = momenta

    phsPositions = property(lambda self: self.positions)
    phsMomenta = property(lambda self: self.momenta)

    def __repr__(self):
        return f"Phs {{phsPositions = {self.positions!r}, phsMomenta = {self.momenta!r}}}"


Cfg, Phs = Config, Phase


class System:
    """Opaque `System m n` (Hamilton.hs:160-169): owns a libhamk handle."""

    def __init__(self, m: int, n: int, inertia, tape_f: T.Tape, tape_u: T.Tape, u_space: int, options=None):
        """options: a `hamilton_amd._abi.HamkOptions` or a dict of its fields (hamk.h `hamk_options`; unset = the
        library's own choice) -- which lanes serve a trajectory, AD strategy, stepping bodies, sincos policy, GSL
        binding, self-check ...  Environment variables remain as test overrides of whatever is left to the library."""
        self.m, self.n = int(m), int(n)
        self.u_space = u_space
        self.tape_f, self.tape_u = tape_f, tape_u
        self.last_status = None
        self.last_nsub = None
        L = _abi.lib()
        f_ops, f_n, f_outs = tape_f.as_ctypes()
        u_ops, u_n, _ = tape_u.as_ctypes()
        w = (ctypes.c_double * self.m)(*[float(v) for v in inertia])
        h = ctypes.c_void_p()
        if isinstance(options, dict):
            options = _abi.HamkOptions(**options)
        _abi.check(L.hamk_system_create_ex(self.m, self.n, w, f_ops, f_n, f_outs, u_ops, u_n, tape_u.outs[0],
                                           u_space, ctypes.byref(options) if options is not None else None, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _abi.lib().hamk_system_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def options(self, B: int = -1) -> dict:
        """What a launch over B trajectories uses (B < 0: a large ensemble), every choice resolved
        (hamk_system_get_options): mapping, ad_mode, bodies, trig, ..., lanes_per_trajectory."""
        o = _abi.HamkOptions()
        _abi.check(_abi.lib().hamk_system_get_options(self._h, int(B), ctypes.byref(o)))
        return o.as_dict()

    def set_ensemble_size(self, total: int):
        """The size of the WHOLE ensemble this handle's launches are pieces of (hamk_system_set_ensemble_size): the library
        then picks the mapping for that size instead of each launch's own -- shards, chunks and resumed runs reproduce the
        one-launch bits whatever the split.  0: back to per-launch choice."""
        _abi.check(_abi.lib().hamk_system_set_ensemble_size(self._h, int(total)))
        return self

    def describe_batch(self, B: int = -1):
        """`source`, `build_info`, `code_size`, `code_object`, `kernel_bytes` below describe the specialisation a launch
        over B trajectories uses (default: the one built at creation = the large-ensemble one)."""
        _abi.check(_abi.lib().hamk_system_describe_batch(self._h, int(B)))
        return self

    @property
    def lanes_per_trajectory(self) -> int:
        """1 (lane kernels), 4 (quad), 16 / 32 / 64 (wave-cooperative) -- of the specialisation being described."""
        src = self.source
        if "HAMK_INSTANTIATE_WAVE" in src:
            return 16 if self.n <= 16 else (32 if self.n <= 32 else 64)
        return 4 if "HAMK_INSTANTIATE_QUAD" in src else 1

    @property
    def source(self) -> str:
        return _abi.lib().hamk_system_source(self._h).decode()

    @property
    def build_info(self) -> str:
        return _abi.lib().hamk_system_build_info(self._h).decode()

    @property
    def code_size(self) -> int:
        return int(_abi.lib().hamk_system_code_size(self._h))

    def kernel_bytes(self, kernel: str) -> int:
        return int(_abi.lib().hamk_system_kernel_bytes(self._h, kernel.encode()))

    @property
    def num_device_functions(self) -> int:
        return int(_abi.lib().hamk_system_kernel_bytes(self._h, None))

    def synchronize(self):
        _abi.check(_abi.lib().hamk_synchronize(self._h))

    @contextlib.contextmanager
    def _on(self, arr: _Arr):
        """Run the enclosed libhamk call where `arr` lives: a handle launches on the calling thread's
        CURRENT device (hamk.h), so tensors of cuda:1 need cuda:1 current for the duration of the
        call -- on torch's current stream of that device.  The handle keeps one module / staging
        state per device, so alternating devices costs nothing after the first use of each."""
        if arr.device:
            with torch.cuda.device(arr.a.device):
                stream = torch.cuda.current_stream(arr.a.device).cuda_stream
                _abi.check(_abi.lib().hamk_set_stream(self._h, ctypes.c_void_p(stream)))
                yield
        else:
            _abi.check(_abi.lib().hamk_set_stream(self._h, None))
            yield

    @property
    def gsl_api(self) -> int:
        """Which binding of hmatrix-gsl's gsl-ode.c stepHam / evolveHam follow: 2 = gsl_odeiv2
        driver (its default build, this library's default), 1 = old gsl_odeiv (-DGSLODE1).  hamk.h."""
        return int(_abi.lib().hamk_system_get_gsl_api(self._h))

    @gsl_api.setter
    def gsl_api(self, api: int):
        _abi.check(_abi.lib().hamk_system_set_gsl_api(self._h, int(api)))

    def code_object(self, which: int = 0) -> bytes:
        """The gfx950 ELF the kernels are loaded from (which = 1: the build without MachineLICM)."""
        L = _abi.lib()
        n = int(L.hamk_system_code_object(self._h, which, None, 0))
        if n == 0:
            return b""
        buf = ctypes.create_string_buffer(n)
        L.hamk_system_code_object(self._h, which, buf, n)
        return buf.raw

    def _after(self, status, single: bool, what: str):
        self.last_status = status
        if single and status is not None:
            st = int(status[0])
            if st & ST_SINGULAR:
                raise SingularSystem(f"{what}: mass matrix J^T M J is singular")


def mkSystem(inertia, f: Callable, u: Callable, n: int, options=None) -> System:
    """mkSystem (Hamilton.hs:201-225): potential over GENERALIZED coordinates.

    `f(q)` maps a list of n values to m values, `u(q)` to a scalar; both written
    against hamilton_amd's traced arithmetic (the `RealFloat a` of the reference).
    m = len(inertia); n cannot be read off a Python function and is explicit."""
    m = len(inertia)
    return System(m, n, inertia, T.trace(f, n, m), T.trace(u, n, None), U_GENERALIZED, options)


def mkSystem_(inertia, f: Callable, u: Callable, n: int, options=None) -> System:
    """mkSystem' (Hamilton.hs:238-254): potential over the underlying CARTESIAN coordinates."""
    m = len(inertia)
    return System(m, n, inertia, T.trace(f, n, m), T.trace(u, m, None), U_CARTESIAN, options)


def system_from_spec(spec, options=None) -> System:
    """Build a System from a hamilton_amd.examples.SystemSpec."""
    tf, tu = spec.trace()
    return System(spec.m, spec.n, spec.inertia, tf, tu, spec.u_space, options)


# ---------------------------------------------------------------------------------------
# state functions
# ---------------------------------------------------------------------------------------
def underlyingPos(s: System, q):
    """underlyingPos (Hamilton.hs:174-178)."""
    qa = _Arr(q, s.n, "positions")
    x = qa.like(s.m)
    with s._on(qa):
        _abi.check(_abi.lib().hamk_coords_batch(s._h, qa.B, qa.ptr, _ptr(x), qa.mem))
    return _shape_out(x, qa.single)


def momenta(s: System, c: Config):
    """momenta (Hamilton.hs:262-269)."""
    qa, va = _pair(c.positions, "positions", c.velocities, "velocities", s.n)
    p = qa.like(s.n)
    with s._on(qa):
        _abi.check(_abi.lib().hamk_to_phase_batch(s._h, qa.B, qa.ptr, va.ptr, _ptr(p), qa.mem))
    return _shape_out(p, qa.single)


def toPhase(s: System, c: Config) -> Phase:
    """toPhase (Hamilton.hs:279-284)."""
    return Phase(c.positions, momenta(s, c))


def velocities(s: System, ph: Phase):
    """velocities (Hamilton.hs:316-324)."""
    qa, pa = _pair(ph.positions, "positions", ph.momenta, "momenta", s.n)
    v, st = qa.like(s.n), qa.like(None, "i4")
    with s._on(qa):
        _abi.check(_abi.lib().hamk_from_phase_batch(s._h, qa.B, qa.ptr, pa.ptr, _ptr(v), _ptr(st), qa.mem))
    s._after(st, qa.single, "velocities")
    return _shape_out(v, qa.single)


def fromPhase(s: System, ph: Phase) -> Config:
    """fromPhase (Hamilton.hs:332-337)."""
    return Config(ph.positions, velocities(s, ph))


def _observe(s: System, q, p, which: str):
    if p is not None:
        qa, pa = _pair(q, "positions", p, "momenta", s.n)
    else:
        qa, pa = _Arr(q, s.n, "positions"), None
    out, st = qa.like(None), qa.like(None, "i4")
    ptrs = {"ke": None, "pe": None, "h": None}
    ptrs[which] = _ptr(out)
    with s._on(qa):
        _abi.check(_abi.lib().hamk_observe_batch(s._h, qa.B, qa.ptr, pa.ptr if pa is not None else None,
                                                 ptrs["ke"], ptrs["pe"], ptrs["h"], _ptr(st), qa.mem))
    if which != "pe":
        s._after(st, qa.single, which)
    return _shape_out(out, qa.single)


def keP(s: System, ph: Phase):
    """keP (Hamilton.hs:341-349)."""
    return _observe(s, ph.positions, ph.momenta, "ke")


def hamiltonian(s: System, ph: Phase):
    """hamiltonian (Hamilton.hs:353-361)."""
    return _observe(s, ph.positions, ph.momenta, "h")


def pe(s: System, q):
    """pe (Hamilton.hs:182-186)."""
    return _observe(s, q, None, "pe")


def _observe_config(s: System, c: Config, which: str):
    qa, va = _pair(c.positions, "positions", c.velocities, "velocities", s.n)
    out = qa.like(None)
    with s._on(qa):
        _abi.check(_abi.lib().hamk_observe_config_batch(
            s._h, qa.B, qa.ptr, va.ptr, _ptr(out) if which == "ke" else None, _ptr(out) if which == "lag" else None, qa.mem))
    return _shape_out(out, qa.single)


def keC(s: System, c: Config):
    """keC (Hamilton.hs:288-296)."""
    return _observe_config(s, c, "ke")


def lagrangian(s: System, c: Config):
    """lagrangian (Hamilton.hs:301-309)."""
    return _observe_config(s, c, "lag")


def hamEqs(s: System, ph: Phase):
    """hamEqs (Hamilton.hs:370-387): returns (dH/dp, -dH/dq)."""
    qa, pa = _pair(ph.positions, "positions", ph.momenta, "momenta", s.n)
    dq, dp, st = qa.like(s.n), qa.like(s.n), qa.like(None, "i4")
    with s._on(qa):
        _abi.check(_abi.lib().hamk_hameqs_batch(s._h, qa.B, qa.ptr, pa.ptr, _ptr(dq), _ptr(dp), _ptr(st), qa.mem))
    s._after(st, qa.single, "hamEqs")
    return _shape_out(dq, qa.single), _shape_out(dp, qa.single)


def sampleConfig(s: System, q_box, qd_box, start: int, count: int, seed: int, device=None) -> Config:
    """Initial Configs of trajectories start .. start + count - 1 of an ensemble, drawn ON THE DEVICE from the global index
    (hamk_sample_batch): uniform boxes q_box / qd_box = n pairs (lo, hi).  Same bits as
    `examples.sample_config` for any (start, count): shards need no host array and no scatter.  device: a torch CUDA
    device (tensors stay in HBM) or None (numpy arrays through the host-staged path)."""
    n = s.n
    if len(q_box) != n or len(qd_box) != n:
        raise ValueError(f"boxes must have {n} (lo, hi) pairs")
    arr = lambda xs: (ctypes.c_double * n)(*[float(x) for x in xs])
    qlo, qhi = arr([b[0] for b in q_box]), arr([b[1] for b in q_box])
    dlo, dhi = arr([b[0] for b in qd_box]), arr([b[1] for b in qd_box])
    if device is not None:
        dev = torch.device(device)
        q = torch.empty((n, int(count)), dtype=torch.float64, device=dev)
        qd = torch.empty_like(q)
    else:
        q, qd = np.empty((n, int(count))), np.empty((n, int(count)))
    qa = _Arr(q, n, "positions")
    with s._on(qa):
        _abi.check(_abi.lib().hamk_sample_batch(s._h, int(count), int(start), int(seed), qlo, qhi, dlo, dhi, _ptr(q), _ptr(qd), qa.mem))
    return Config(q, qd)


# ---------------------------------------------------------------------------------------
# time stepping
# ---------------------------------------------------------------------------------------
def stepHam(r: float, s: System, ph: Phase, inplace: bool = False) -> Phase:
    """stepHam (Hamilton.hs:390-402): adaptive RKF45 (GSL semantics) from 0 to r.  inplace=True
    advances the given arrays without copying (ensemble loops; the reference's signature is pure)."""
    qa, pa = _pair(ph.positions, "positions", ph.momenta, "momenta", s.n)
    if inplace:
        _in_place(qa, ph.positions, "positions"); _in_place(pa, ph.momenta, "momenta")
    q, p = (qa.a, pa.a) if inplace else (qa.clone(), pa.clone())
    st, ns = qa.like(None, "i4"), qa.like(None, "i4")
    with s._on(qa):
        _abi.check(_abi.lib().hamk_step_ham_batch(s._h, qa.B, _ptr(q), _ptr(p), float(r), _ptr(st), _ptr(ns), qa.mem))
    s.last_nsub = ns
    s._after(st, qa.single, "stepHam")
    return Phase(_shape_out(q, qa.single), _shape_out(p, qa.single))


def iterateStepHam(r: float, ncalls: int, s: System, ph: Phase, every: int = 0, inplace: bool = False):
    """`iterate (stepHam r s)` (README.md:150; the demo's frame loop, app/Examples.hs:429) -- `ncalls` consecutive
    stepHam r in ONE launch (hamk_step_ham_iterate): bit-identical to calling stepHam `ncalls` times, without
    the launch and synchronisation per call.  Returns the final Phase, or (final Phase, frames) when
    every > 0: frames = the Phase after every `every`-th call, arrays shaped [ncalls // every, n(, B)]."""
    qa, pa = _pair(ph.positions, "positions", ph.momenta, "momenta", s.n)
    if inplace:
        _in_place(qa, ph.positions, "positions"); _in_place(pa, ph.momenta, "momenta")
    q, p = (qa.a, pa.a) if inplace else (qa.clone(), pa.clone())
    st, ns = qa.like(None, "i4"), qa.like(None, "i4")
    rows = (int(ncalls) // int(every)) if every > 0 else 0
    fq = qa.like(s.n, lead=(rows,)) if rows else None
    fp = qa.like(s.n, lead=(rows,)) if rows else None
    with s._on(qa):
        _abi.check(_abi.lib().hamk_step_ham_iterate(s._h, qa.B, _ptr(q), _ptr(p), float(r), int(ncalls), int(every) if rows else 0,
                                                    _ptr(fq), _ptr(fp), _ptr(st), _ptr(ns), qa.mem))
    s.last_nsub = ns
    s._after(st, qa.single, "iterateStepHam")
    out = Phase(_shape_out(q, qa.single), _shape_out(p, qa.single))
    if every > 0:
        if rows == 0:
            return out, Phase(np.empty((0, s.n)), np.empty((0, s.n)))
        return out, Phase(_shape_out(fq, qa.single), _shape_out(fp, qa.single))
    return out


def evolveHam(s: System, p0: Phase, ts, h0: float = 0.0, eps_abs: float = 0.0, eps_rel: float = 0.0) -> List[Phase]:
    """evolveHam (Hamilton.hs:433-462): the state at each of the >= 2 times; element 0 is p0."""
    ts = np.ascontiguousarray(np.asarray(ts, dtype=np.float64))
    if ts.ndim != 1 or len(ts) < 2:
        raise ValueError("evolveHam needs at least two solution times (2 <= s)")
    qa, pa = _pair(p0.positions, "positions", p0.momenta, "momenta", s.n)
    nt = len(ts)
    qo, po = qa.like(s.n, lead=(nt,)), qa.like(s.n, lead=(nt,))
    st, ns = qa.like(None, "i4"), qa.like(None, "i4")
    with s._on(qa):
        _abi.check(_abi.lib().hamk_evolve_ham_batch(
            s._h, qa.B, qa.ptr, pa.ptr, nt, ts.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _ptr(qo), _ptr(po),
            float(h0), float(eps_abs), float(eps_rel), _ptr(st), _ptr(ns), qa.mem))
    s.last_nsub = ns
    s._after(st, qa.single, "evolveHam")
    return [Phase(_shape_out(qo[r], qa.single), _shape_out(po[r], qa.single)) for r in range(nt)]


def evolveHam_(s: System, p0: Phase, ts: Sequence[float]) -> List[Phase]:
    """evolveHam' (Hamilton.hs:409-429): [] -> []; [x] -> evolve over [0, x] and drop the first."""
    ts = list(ts)
    if not ts:
        return []
    if len(ts) == 1:
        return evolveHam(s, p0, [0.0, ts[0]])[1:]
    return evolveHam(s, p0, ts)


def stepHamC(r: float, s: System, c: Config) -> Config:
    """stepHamC (Hamilton.hs:502-515)."""
    return fromPhase(s, stepHam(r, s, toPhase(s, c)))


def evolveHamC(s: System, c0: Config, ts) -> List[Config]:
    """evolveHamC (Hamilton.hs:486-500)."""
    return [fromPhase(s, ph) for ph in evolveHam(s, toPhase(s, c0), ts)]


def evolveHamC_(s: System, c0: Config, ts: Sequence[float]) -> List[Config]:
    """evolveHamC' (Hamilton.hs:470-484)."""
    return [fromPhase(s, ph) for ph in evolveHam_(s, toPhase(s, c0), ts)]


def rk4Steps(dt: float, nsteps: int, s: System, ph: Phase, inplace: bool = False, drift_tol: float = 0.0) -> Phase:
    """Classic fixed-step RK4 over hamEqs -- named by BASELINE.json's north_star; the
    reference has no fixed-step integrator.  inplace=True advances the
    given device/host arrays without copying (bench path).  drift_tol > 0: the launch
    checks its own energy invariant and sets ST_DRIFT in `System.last_status` where
    |H_exit - H_entry| > drift_tol * max(1, |H_entry|) (hamk_rk4_steps_checked)."""
    qa, pa = _pair(ph.positions, "positions", ph.momenta, "momenta", s.n)
    if inplace:
        _in_place(qa, ph.positions, "positions"); _in_place(pa, ph.momenta, "momenta")
    q, p = (qa.a, pa.a) if inplace else (qa.clone(), pa.clone())
    st = qa.like(None, "i4")
    with s._on(qa):
        _abi.check(_abi.lib().hamk_rk4_steps_checked(s._h, qa.B, _ptr(q), _ptr(p), float(dt), int(nsteps), float(drift_tol),
                                                     _ptr(st), qa.mem))
    s.last_status = st
    return Phase(_shape_out(q, qa.single), _shape_out(p, qa.single))


# ---------------------------------------------------------------------------------------
# ensemble checkpoint: device- or host-resident state <-> one flat file
# ---------------------------------------------------------------------------------------
def saveCheckpoint(path: str, ph: Phase, n: int, steps_done: int = 0, seed: int = 0, t: float = 0.0) -> None:
    """hamk_checkpoint_write: q[n][B], p[n][B] (numpy or torch CUDA tensors) + bookkeeping."""
    qa, pa = _pair(ph.positions, "positions", ph.momenta, "momenta", n)
    if qa.device:
        torch.cuda.current_stream(qa.a.device).synchronize()     # the copy out runs on the null stream
    ctx = torch.cuda.device(qa.a.device) if qa.device else contextlib.nullcontext()
    with ctx:
        _abi.check(_abi.lib().hamk_checkpoint_write(str(path).encode(), n, qa.B, qa.ptr, pa.ptr, qa.mem,
                                                    int(steps_done), int(seed), float(t)))


def checkpointInfo(path: str) -> dict:
    n, B, st, seed, t = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_uint64(), ctypes.c_double()
    _abi.check(_abi.lib().hamk_checkpoint_info(str(path).encode(), ctypes.byref(n), ctypes.byref(B), ctypes.byref(st),
                                               ctypes.byref(seed), ctypes.byref(t)))
    return {"n": n.value, "B": B.value, "steps_done": st.value, "seed": seed.value, "t": t.value}


def loadCheckpoint(path: str, device=None):
    """Returns (Phase, info).  device=None: numpy arrays; a torch device: CUDA tensors on it."""
    info = checkpointInfo(path)
    n, B = info["n"], info["B"]
    if device is None:
        q, p = np.empty((n, B)), np.empty((n, B))
        _abi.check(_abi.lib().hamk_checkpoint_read(str(path).encode(), n, B, q.ctypes.data, p.ctypes.data, MEM_HOST))
    else:
        dev = torch.device(device)
        q = torch.empty((n, B), dtype=torch.float64, device=dev)
        p = torch.empty((n, B), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            _abi.check(_abi.lib().hamk_checkpoint_read(str(path).encode(), n, B, q.data_ptr(), p.data_ptr(), MEM_DEVICE))
    return Phase(q, p), info
