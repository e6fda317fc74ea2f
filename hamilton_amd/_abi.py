"""ctypes binding of libhamk.so -- exactly the entry points of include/hamk.h.

This is the stub a maintainer of the reference would write in Haskell as
`foreign import ccall` declarations; it contains no numerics.
The library is built in-tree (hamilton_amd/libhamk.so) and the import fails
loudly when it is missing: there is no CPU fallback for the product path.
"""
from __future__ import annotations

import ctypes
import os

try:
    # torch wheels bundle their own libamdhip64.so.7; it must be the first HIP runtime
    # mapped into the process, otherwise libhamk.so's rpath copy (/opt/rocm) and torch's
    # copy both initialise and the second one finds "No HIP GPUs".  With torch loaded
    # first, libhamk.so binds to the already-mapped soname: one runtime, shared streams.
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional plumbing
    torch = None

from .tracer import HamkOp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhamk.so")

HAMK_OK = 0
HAMK_ERR_INVALID, HAMK_ERR_TAPE, HAMK_ERR_COMPILE, HAMK_ERR_HIP, HAMK_ERR_NODEVICE, HAMK_ERR_UNSUPPORTED = \
    -1, -2, -3, -4, -5, -6
ST_SINGULAR, ST_NONFINITE, ST_UNDERFLOW, ST_MAXSTEPS, ST_DRIFT = 1, 2, 4, 8, 16
MEM_HOST, MEM_DEVICE = 0, 1

_dp = ctypes.c_void_p          # double*  (host or device address)
_ip = ctypes.c_void_p          # int32_t*
_h = ctypes.c_void_p           # hamk_system*
_i32, _i64, _f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_double

AUTO, ON, OFF = 0, 1, 2
MAP_LANE, MAP_WAVE, MAP_QUAD = 1, 2, 3
HAMK_COMM_ID_BYTES = 128
AD_H, AD_D, AD_R = 1, 2, 3
BODY_UNROLLED, BODY_STAGE_LOOP = 1, 2
TRIG_DIRECT, TRIG_TABLE, TRIG_TABLE_ROTATE = 1, 2, 3
BUILD_DEFAULT, BUILD_NOLICM = 1, 2
OPTIONS_VERSION = 0x484B0005                                # HAMK_OPTIONS_VERSION of include/hamk.h


class HamkOptions(ctypes.Structure):
    """`hamk_options` of include/hamk.h: what the library otherwise decides for itself (0 = HAMK_AUTO)."""
    _fields_ = [("size", ctypes.c_uint32), ("version", ctypes.c_uint32), ("mapping", ctypes.c_int32), ("ad_mode", ctypes.c_int32),
                ("rk4_body", ctypes.c_int32), ("rkf_body", ctypes.c_int32), ("trig", ctypes.c_int32),
                ("gsl_api", ctypes.c_int32), ("self_check", ctypes.c_int32), ("build", ctypes.c_int32),
                ("rk4_min_waves", ctypes.c_int32), ("k_reassoc", ctypes.c_int32),
                ("rk4_park", ctypes.c_int32), ("max_substeps", ctypes.c_int32), ("cache", ctypes.c_int32),
                ("lanes_per_trajectory", ctypes.c_int32), ("rkf_park", ctypes.c_int32), ("_align", ctypes.c_int32),
                ("ensemble_size", ctypes.c_int64), ("reserved", ctypes.c_int32 * 12)]

    def __init__(self, **kw):
        super().__init__()
        self.size = ctypes.sizeof(HamkOptions)
        self.version = OPTIONS_VERSION
        for k, v in kw.items():
            if k not in dict(self._fields_) or k in ("size", "version", "reserved", "_align"):
                raise TypeError(f"hamk_options has no field {k!r}")
            setattr(self, k, int(v))

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k not in ("size", "version", "reserved", "_align")}


# name -> (restype, argtypes); mirrors include/hamk.h declaration by declaration
SIGNATURES = {
    "hamk_system_create": (ctypes.c_int, [_i32, _i32, ctypes.POINTER(ctypes.c_double),
                                          ctypes.POINTER(HamkOp), _i32, ctypes.POINTER(ctypes.c_int32),
                                          ctypes.POINTER(HamkOp), _i32, _i32, _i32, ctypes.POINTER(_h)]),
    "hamk_system_create_ex": (ctypes.c_int, [_i32, _i32, ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(HamkOp), _i32, ctypes.POINTER(ctypes.c_int32),
                                             ctypes.POINTER(HamkOp), _i32, _i32, _i32, ctypes.POINTER(HamkOptions), ctypes.POINTER(_h)]),
    "hamk_options_init": (None, [ctypes.POINTER(HamkOptions)]),
    "hamk_system_get_options": (ctypes.c_int, [_h, _i64, ctypes.POINTER(HamkOptions)]),
    "hamk_system_describe_batch": (ctypes.c_int, [_h, _i64]),
    "hamk_system_set_ensemble_size": (ctypes.c_int, [_h, _i64]),
    "hamk_system_destroy": (None, [_h]),
    "hamk_system_dims": (ctypes.c_int, [_h, ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "hamk_set_stream": (ctypes.c_int, [_h, ctypes.c_void_p]),
    "hamk_synchronize": (ctypes.c_int, [_h]),
    "hamk_system_source": (ctypes.c_char_p, [_h]),
    "hamk_system_code_size": (_i64, [_h]),
    "hamk_system_build_info": (ctypes.c_char_p, [_h]),
    "hamk_system_kernel_bytes": (_i64, [_h, ctypes.c_char_p]),
    "hamk_coords_batch": (ctypes.c_int, [_h, _i64, _dp, _dp, _i32]),
    "hamk_to_phase_batch": (ctypes.c_int, [_h, _i64, _dp, _dp, _dp, _i32]),
    "hamk_from_phase_batch": (ctypes.c_int, [_h, _i64, _dp, _dp, _dp, _ip, _i32]),
    "hamk_observe_batch": (ctypes.c_int, [_h, _i64, _dp, _dp, _dp, _dp, _dp, _ip, _i32]),
    "hamk_observe_config_batch": (ctypes.c_int, [_h, _i64, _dp, _dp, _dp, _dp, _i32]),
    "hamk_hameqs_batch": (ctypes.c_int, [_h, _i64, _dp, _dp, _dp, _dp, _ip, _i32]),
    "hamk_sample_batch": (ctypes.c_int, [_h, _i64, _i64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _dp, _dp, _i32]),
    "hamk_rk4_steps": (ctypes.c_int, [_h, _i64, _dp, _dp, _f64, _i32, _ip, _i32]),
    "hamk_rk4_steps_checked": (ctypes.c_int, [_h, _i64, _dp, _dp, _f64, _i32, _f64, _ip, _i32]),
    "hamk_system_set_gsl_api": (ctypes.c_int, [_h, _i32]),
    "hamk_system_get_gsl_api": (_i32, [_h]),
    "hamk_system_code_object": (_i64, [_h, _i32, ctypes.c_void_p, _i64]),
    "hamk_checkpoint_write": (ctypes.c_int, [ctypes.c_char_p, _i32, _i64, _dp, _dp, _i32, _i64, ctypes.c_uint64, _f64]),
    "hamk_checkpoint_info": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(_i32), ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                            ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(_f64)]),
    "hamk_checkpoint_read": (ctypes.c_int, [ctypes.c_char_p, _i32, _i64, _dp, _dp, _i32]),
    "hamk_step_ham_batch": (ctypes.c_int, [_h, _i64, _dp, _dp, _f64, _ip, _ip, _i32]),
    "hamk_step_ham_iterate": (ctypes.c_int, [_h, _i64, _dp, _dp, _f64, _i32, _i32, _dp, _dp, _ip, _ip, _i32]),
    "hamk_evolve_ham_batch": (ctypes.c_int, [_h, _i64, _dp, _dp, _i32, ctypes.POINTER(ctypes.c_double), _dp, _dp,
                                             _f64, _f64, _f64, _ip, _ip, _i32]),
    "hamk_last_error": (ctypes.c_char_p, []),
    "hamk_version": (ctypes.c_char_p, []),
    "hamk_device_count": (ctypes.c_int, []),
    "hamk_set_device": (ctypes.c_int, [_i32]),
    "hamk_get_device": (ctypes.c_int, [ctypes.POINTER(_i32)]),
    "hamk_device_malloc": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _i64]),
    "hamk_device_free": (ctypes.c_int, [ctypes.c_void_p]),
    "hamk_memcpy": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _i64, _i32]),
    "hamk_gather_batch": (ctypes.c_int, [_i32, _i32, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_void_p),
                                         ctypes.c_void_p, _i32]),
    "hamk_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "hamk_comm_create": (ctypes.c_int, [ctypes.c_void_p, _i32, _i32, ctypes.POINTER(ctypes.c_void_p)]),
    "hamk_comm_allgather_batch": (ctypes.c_int, [ctypes.c_void_p, _i32, ctypes.POINTER(_i64), ctypes.c_void_p, ctypes.c_void_p]),
    "hamk_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
}

_lib = None


class HamkError(RuntimeError):
    """API / toolchain / HIP failure (the reference raises Haskell exceptions here)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"libhamk error {code}: {msg}")
        self.code = code


def _default_cache_dir():
    """A pre-compiled code-object cache that travels with the source tree (`.hamk_cache/`, filled by
    scripts/warm_cache.py or __graft_entry__.build(); hiprtc cross-compiles without a GPU) is used when
    the caller has not chosen a location: a fresh GPU box then measures instead of compiling.  Entries
    are verified (SHA-256 of key material and payload) and the directory must be owner-only, or
    libhamk ignores it (hamk_build.cpp)."""
    if "HAMK_CACHE_DIR" in os.environ:
        return
    d = os.path.join(os.path.dirname(_HERE), ".hamk_cache")
    try:
        st = os.lstat(d)
    except OSError:
        return
    import stat
    # the same test libhamk applies (hamk_build.cpp private_dir): a real directory of ours nobody else can touch.  A tree
    # unpacked by another user or with group/other bits fails it -- and libhamk has no fallback once HAMK_CACHE_DIR names
    # a rejected directory: then leave it unset, so the per-user cache ($XDG_CACHE_HOME/hamk, ~/.cache/hamk) applies.
    if stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0:
        os.environ["HAMK_CACHE_DIR"] = d
    else:
        import warnings
        warnings.warn(f"{d} is not an owner-only directory of this user: the in-tree code-object cache is ignored "
                      "(chmod 700 it, or set HAMK_CACHE_DIR)", RuntimeWarning, stacklevel=2)


def lib():
    global _lib
    if _lib is None:
        _default_cache_dir()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C hamilton_amd/csrc`.  hamilton_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int):
    if rc != HAMK_OK:
        raise HamkError(rc, lib().hamk_last_error().decode("utf-8", "replace"))
