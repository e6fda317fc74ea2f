"""CPU: the C-ABI library loads, exports every symbol include/hamk.h declares, specialises
systems with hiprtc (no GPU needed for that), and rejects malformed input with the
documented codes.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ALL_GOLDEN_SYSTEMS, ROOT
from hamilton_amd import examples as E
from hamilton_amd import tracer as T


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hamk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hamk_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(hamk_lib):
    names = declared_symbols()
    assert len(names) >= 19
    for name in names:
        assert hasattr(hamk_lib, name), f"{name} declared in include/hamk.h but not exported by libhamk.so"


def test_nothing_but_the_c_abi_is_exported(hamk_lib):
    """`nm -D --defined-only libhamk.so` = the prototypes of include/hamk.h, nothing else: -fvisibility=hidden and
    -fvisibility-inlines-hidden for the library's own code, the linker's version script (csrc/hamk.map) for what libstdc++'s headers
    instantiate into it (round 5's library also exported a handful of weak std::vector members)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "hamilton_amd", "libhamk.so")], capture_output=True, text=True, check=True).stdout
    rows = [ln.split() for ln in out.splitlines() if ln.strip()]
    assert sorted(r[-1] for r in rows) == declared_symbols()
    assert all(r[-2] == "T" for r in rows)


def test_communicator_argument_checks(hamk_lib):
    """hamk_comm_* (the RCCL all-gather for one-process-per-GPU hosts): what can be refused without a GPU is refused before RCCL is
    touched; destroying nothing is fine."""
    from hamilton_amd import _abi
    L = hamk_lib
    out = ctypes.c_void_p()
    ident = (ctypes.c_char * _abi.HAMK_COMM_ID_BYTES)()
    assert L.hamk_comm_unique_id(None) == _abi.HAMK_ERR_INVALID
    assert L.hamk_comm_create(None, 1, 0, ctypes.byref(out)) == _abi.HAMK_ERR_INVALID and not out.value
    assert L.hamk_comm_create(ident, 0, 0, ctypes.byref(out)) == _abi.HAMK_ERR_INVALID
    assert L.hamk_comm_create(ident, 2, 2, ctypes.byref(out)) == _abi.HAMK_ERR_INVALID
    assert L.hamk_comm_create(ident, 2, -1, ctypes.byref(out)) == _abi.HAMK_ERR_INVALID
    assert L.hamk_comm_create(ident, 1, 0, None) == _abi.HAMK_ERR_INVALID
    assert L.hamk_comm_allgather_batch(None, 2, None, None, None) == _abi.HAMK_ERR_INVALID
    assert b"communicator" in L.hamk_last_error()
    assert L.hamk_comm_destroy(None) == _abi.HAMK_OK
    text = open(os.path.join(ROOT, "include", "hamk.h")).read()
    assert int(re.search(r"#define HAMK_COMM_ID_BYTES (\d+)", text).group(1)) == _abi.HAMK_COMM_ID_BYTES == 128       # = NCCL_UNIQUE_ID_BYTES (rccl.h)


def test_binding_table_matches_header(hamk_lib):
    from hamilton_amd import _abi
    assert sorted(_abi.SIGNATURES) == declared_symbols()


def test_op_struct_layout_and_numbering():
    assert ctypes.sizeof(T.HamkOp) == 24
    text = open(os.path.join(ROOT, "include", "hamk.h")).read()
    enum = dict((k, int(v)) for k, v in re.findall(r"HAMK_OP_([A-Z0-9]+)\s*=\s*(\d+)", text))
    for name, val in enum.items():
        assert getattr(T, "OP_" + name) == val, name


@pytest.mark.parametrize("name", ALL_GOLDEN_SYSTEMS + ["chain8"])
def test_specialisation_compiles_for_gfx950(hamk_lib, name):
    """hamk_system_create = tape -> generated source -> hiprtc (gfx950); works without a GPU."""
    from hamilton_amd import api
    s = api.system_from_spec(E.get(name))
    assert s.code_size > 1000
    src = s.source
    assert "HAMK_INSTANTIATE(HamkSys)" in src and f"N = {s.n};" in src and f"M = {s.m};" in src


def test_malformed_tapes_are_rejected(hamk_lib):
    from hamilton_amd import _abi
    ops = (T.HamkOp * 2)()
    ops[0].op, ops[0].a = T.OP_INPUT, 0
    ops[1].op, ops[1].a = T.OP_SIN, 1          # forward reference to itself
    outs = (ctypes.c_int32 * 1)(1)
    w = (ctypes.c_double * 1)(1.0)
    h = ctypes.c_void_p()
    rc = hamk_lib.hamk_system_create(1, 1, w, ops, 2, outs, ops, 1, 0, 0, ctypes.byref(h))
    assert rc == _abi.HAMK_ERR_TAPE and b"earlier value" in hamk_lib.hamk_last_error()
    ops[1].op = 99
    rc = hamk_lib.hamk_system_create(1, 1, w, ops, 2, outs, ops, 1, 0, 0, ctypes.byref(h))
    assert rc == _abi.HAMK_ERR_TAPE
    rc = hamk_lib.hamk_system_create(0, 1, w, ops, 1, outs, ops, 1, 0, 0, ctypes.byref(h))
    assert rc == _abi.HAMK_ERR_INVALID
    rc = hamk_lib.hamk_system_create(1, 1, w, ops, 1, outs, ops, 1, 0, 7, ctypes.byref(h))
    assert rc == _abi.HAMK_ERR_INVALID


def test_sizes_beyond_the_kernels_are_refused(hamk_lib):
    """n <= 64 (m <= 128 on the wave path): larger systems get HAMK_ERR_UNSUPPORTED and a message, not
    a compile that never ends."""
    from hamilton_amd import _abi, api
    spec = E.chain(65)
    with pytest.raises(api.HamkError) as e:
        api.system_from_spec(spec)
    assert e.value.code == _abi.HAMK_ERR_UNSUPPORTED and "supported sizes" in str(e.value)


def test_defaults_of_the_wave_kernels(hamk_lib, monkeypatch):
    """What the library chooses for n > 16 on the wave mapping (each choice measured on MI355X): LDL^T in panels of 16
    with MFMA trailing updates -- the only factorisation since round 4, a small forced system being one panel --; the RK4
    kernel capped for two wavefronts per SIMD (stated in the generated source and reported since round 5), unless the host asks
    for one; the structure of the Jacobian handed to the matrix-core accumulation of K."""
    from hamilton_amd import _abi, api
    mid = api.system_from_spec(E.get("chain20"), {"mapping": _abi.MAP_WAVE})
    assert "hamk_wave.hpp" in mid.source and "#define HAMK_RK4_MIN_WAVES 2" in mid.source and mid.options()["rk4_min_waves"] == 2
    one = api.system_from_spec(E.get("chain20"), {"mapping": _abi.MAP_WAVE, "rk4_min_waves": 1})
    assert "#define HAMK_RK4_MIN_WAVES 1" in one.source and one.options()["rk4_min_waves"] == 1
    big = api.system_from_spec(E.get("chain33")).source
    assert "#define HAMK_RK4_MIN_WAVES_BIG 2" in big
    # chain: x_k, y_k depend on q_0..q_k -- the tables SinkK::flush reads its block range from
    import re
    hi = [int(x) for x in re.search(r"dep_hi\(int k\) \{\s*constexpr int w\[\d+\] = \{([^}]*)\}", mid.source).group(1).split(",")]
    lo = [int(x) for x in re.search(r"dep_lo\(int k\) \{\s*constexpr int w\[\d+\] = \{([^}]*)\}", mid.source).group(1).split(",")]
    seq = [int(x) for x in re.search(r"seq_out\(int k\) \{\s*constexpr int w\[\d+\] = \{([^}]*)\}", mid.source).group(1).split(",")]
    assert hi == [k // 2 for k in range(40)] and lo == [0] * 40 and sorted(seq) == list(range(40))
    monkeypatch.setenv("HAMK_WAVE", "1")
    small = api.system_from_spec(E.get("chain8"))
    assert "hamk_wave.hpp" in small.source and small.lanes_per_trajectory == 16
    assert "HAMK_WAVE_BLOCKED" not in small.source + big


def test_calls_fail_loudly_without_a_gpu(hamk_lib):
    """There is no CPU fallback: on a box without a GPU a compute call returns an error code."""
    from hamilton_amd import api
    if hamk_lib.hamk_device_count() > 0:
        pytest.skip("a GPU is visible")
    s = api.system_from_spec(E.get("pendulum"))
    with pytest.raises(api.HamkError):
        api.hamEqs(s, api.Phase(np.array([0.1]), np.array([0.2])))


@pytest.mark.parametrize("name", ["doublePendulum", "room", "opcodeZoo", "chain8"])
def test_kernels_come_from_the_build_that_spills_fewer_scalar_registers(hamk_lib, monkeypatch, name):
    """Every kernel is taken from whichever of the two builds (default options / without
    MachineLICM) spills fewer SGPRs, the default build on a tie (hamk_build.cpp build_code).  The
    one kernel ever seen to give run-to-run different results spilled 101;
    the headline RK4 kernel spills none either way and stays on the default build."""
    from hamilton_amd import api

    def spills(force):
        if force is None:
            monkeypatch.delenv("HAMK_NOLICM", raising=False)
        else:
            monkeypatch.setenv("HAMK_NOLICM", force)
        info = api.system_from_spec(E.get(name)).build_info
        rows = [re.match(r"(\S+) build=(\S+) bytes=(\d+) sgpr_spills=(-?\d+)", l).groups() for l in info.splitlines() if l]
        assert len(rows) == 9
        return {k: (b, int(n)) for k, b, _, n in rows}

    dflt, nolicm, chosen = spills("0"), spills("1"), spills(None)
    for k in chosen:
        want = "no-machine-licm" if nolicm[k][1] < dflt[k][1] else "default"
        assert chosen[k] == (want, min(dflt[k][1], nolicm[k][1])), (name, k, dflt[k], nolicm[k], chosen[k])
    assert chosen["hamk_rk4_steps_k"][1] == 0
    if name == "doublePendulum":
        assert chosen["hamk_rk4_steps_k"][0] == "default" and chosen["hamk_rkf45_k"] == ("no-machine-licm", 0)


def test_device_memory_entry_points_without_a_gpu(hamk_lib):
    """Device selection / memory / gather: argument checks work anywhere; without a GPU the calls that
    need one return an error code and a message instead of crashing."""
    from hamilton_amd import _abi
    L = hamk_lib
    assert L.hamk_memcpy(None, None, -1, 0) == _abi.HAMK_ERR_INVALID
    assert L.hamk_memcpy(None, None, 0, 0) == _abi.HAMK_OK
    buf = (ctypes.c_double * 4)()
    assert L.hamk_memcpy(buf, buf, 32, 9) == _abi.HAMK_ERR_INVALID          # unknown kind
    assert L.hamk_gather_batch(-1, 2, None, None, None, 0) == _abi.HAMK_ERR_INVALID
    assert L.hamk_gather_batch(0, 2, None, None, None, 0) == _abi.HAMK_OK     # nothing to gather
    assert L.hamk_gather_batch(0, 0, None, None, None, 0) == _abi.HAMK_ERR_INVALID
    p = ctypes.c_void_p()
    assert L.hamk_device_malloc(ctypes.byref(p), 0) == _abi.HAMK_OK and not p.value
    assert L.hamk_device_free(None) == _abi.HAMK_OK
    if L.hamk_device_count() == 0:
        d = ctypes.c_int32(-7)
        assert L.hamk_set_device(0) < 0 and L.hamk_last_error()
        assert L.hamk_get_device(ctypes.byref(d)) < 0
        assert L.hamk_device_malloc(ctypes.byref(p), 64) < 0 and not p.value


def test_argument_checks(hamk_lib):
    from hamilton_amd import api
    s = api.system_from_spec(E.get("pendulum"))
    with pytest.raises(ValueError):
        api.hamEqs(s, api.Phase(np.zeros((2, 3)), np.zeros((2, 3))))      # n = 1, not 2
    with pytest.raises(ValueError):
        api.evolveHam(s, api.Phase(np.zeros(1), np.zeros(1)), [0.0])       # needs 2 <= s
    assert api.evolveHam_(s, api.Phase(np.zeros(1), np.zeros(1)), []) == []
    with pytest.raises(ValueError, match="different ensembles"):
        api.hamEqs(s, api.Phase(np.zeros((1, 3)), np.zeros((1, 4))))
    with pytest.raises(ValueError, match="inplace=True"):               # a float32 array would be converted: not in place
        api.rk4Steps(0.01, 1, s, api.Phase(np.zeros((1, 3), dtype=np.float32), np.zeros((1, 3))), inplace=True)
    with pytest.raises(ValueError, match="inplace=True"):
        api.stepHam(0.01, s, api.Phase([0.1], [0.0]), inplace=True)


@pytest.mark.parametrize("name", ALL_GOLDEN_SYSTEMS + ["chain8"])
def test_kernels_stay_within_branch_reach(hamk_lib, name):
    """Every kernel's machine code stays well inside the +-128 KiB reach of a SOPP branch
    (libhamk falls back to the stage-loop bodies above 64 KiB; see hamk_dispatch.cpp variant_for)."""
    from hamilton_amd import api
    s = api.system_from_spec(E.get(name))
    for k in ("hamk_rk4_steps_k", "hamk_rkf45_k", "hamk_hameqs_k"):
        nbytes = s.kernel_bytes(k)
        assert 0 < nbytes < 100 * 1024, (name, k, nbytes)
    # every device function inlined: no call frames, no scratch for calls (a recursive helper
    # once slipped through as a real call and cost 120 VGPRs + spills in the adaptive stepper)
    assert s.num_device_functions == 9, (name, s.num_device_functions)      # 8 kernels of the path + the self-check's scribble kernel


@pytest.mark.parametrize("name", ["room", "spring", "twoBody", "opcodeZoo", "chain20"])
def test_wave_specialisation_compiles_for_gfx950(hamk_lib, monkeypatch, name):
    """The wave-cooperative module (hamk_wave.hpp) builds for gfx950 too -- forced here on small
    systems, incl. `room` whose coordinate map is the identity (outputs that are inputs)."""
    from hamilton_amd import api
    monkeypatch.setenv("HAMK_WAVE", "1")
    s = api.system_from_spec(E.get(name))
    assert "HAMK_INSTANTIATE_WAVE(HamkSys)" in s.source and s.num_device_functions == 9


def test_options_cross_the_abi_not_the_environment(hamk_lib, monkeypatch):
    """hamk_system_create_ex / hamk_system_get_options: a host language selects the specialisation through a struct;
    environment variables only fill what the struct leaves to the library (HAMK_AUTO)."""
    from hamilton_amd import _abi, api
    for k in ("HAMK_WAVE", "HAMK_AD_MODE", "HAMK_TRIG_LUT", "HAMK_RK4_LOOP", "HAMK_RKF_LOOP", "HAMK_GSL_API"):
        monkeypatch.delenv(k, raising=False)
    spec = E.get("chain8")
    d = api.system_from_spec(spec).options()
    assert d["mapping"] == _abi.MAP_LANE and d["ad_mode"] == _abi.AD_R and d["lanes_per_trajectory"] == 1
    assert d["gsl_api"] == 2 and d["self_check"] == _abi.ON and d["k_reassoc"] == _abi.ON
    s = api.system_from_spec(spec, {"mapping": _abi.MAP_WAVE, "ad_mode": _abi.AD_D, "trig": _abi.TRIG_DIRECT, "gsl_api": 1,
                                    "self_check": _abi.OFF, "max_substeps": 777})
    r = s.options()
    assert (r["mapping"], r["trig"], r["gsl_api"], r["self_check"], r["max_substeps"]) == (_abi.MAP_WAVE, _abi.TRIG_DIRECT, 1, _abi.OFF, 777)
    assert r["lanes_per_trajectory"] == 16 and "HAMK_INSTANTIATE_WAVE" in s.source and s.gsl_api == 1
    # the environment is an override of AUTO fields only
    monkeypatch.setenv("HAMK_AD_MODE", "D")
    monkeypatch.setenv("HAMK_GSL_API", "1")
    assert api.system_from_spec(spec).options()["ad_mode"] == _abi.AD_D
    assert api.system_from_spec(spec, {"ad_mode": _abi.AD_R, "gsl_api": 2}).options()["ad_mode"] == _abi.AD_R
    assert api.system_from_spec(spec, {"ad_mode": _abi.AD_R, "gsl_api": 2}).gsl_api == 2
    # misuse
    with pytest.raises(api.HamkError) as e:
        api.system_from_spec(spec, {"mapping": 9})
    assert e.value.code == _abi.HAMK_ERR_INVALID
    with pytest.raises(api.HamkError) as e:
        api.system_from_spec(E.get("chain20"), {"mapping": _abi.MAP_LANE})
    assert e.value.code == _abi.HAMK_ERR_UNSUPPORTED
    # a caller built against an older, shorter header: the tail of the struct stays AUTO
    o = _abi.HamkOptions(mapping=_abi.MAP_WAVE)
    o.size = 12
    o.k_reassoc = _abi.OFF                                    # beyond `size`: must be ignored
    assert api.system_from_spec(spec, o).options()["k_reassoc"] == _abi.ON
    o.size = 0
    with pytest.raises(api.HamkError):
        api.system_from_spec(spec, o)


def test_mapping_defaults_by_size_and_structure(hamk_lib):
    """n <= 16: one trajectory per lane; 17 <= n <= 32 with a sparse Jacobian (the chains): four lanes per trajectory for
    every kernel of the path; n > 32: wave-cooperative."""
    from hamilton_amd import _abi, api
    assert api.system_from_spec(E.get("chain16")).options()["mapping"] == _abi.MAP_LANE
    # ensembles under 32 768 leave the lane kernels from n = 11 (a lane kernel puts 64 trajectories in a wavefront);
    # the LDS-parked RK4 state from n = 14 (profiles/r03_rules_probe.jsonl)
    for name, small, park in (("chain10", _abi.MAP_LANE, _abi.OFF), ("chain12", _abi.MAP_QUAD, _abi.OFF), ("chain14", _abi.MAP_QUAD, _abi.ON)):
        t = api.system_from_spec(E.get(name))
        assert t.options(16384)["mapping"] == small and t.options(32768)["mapping"] == _abi.MAP_LANE, name
        assert t.options(65536)["rk4_park"] == park, name
    # the adaptive stepper's vectors parked in LDS / a run-time-indexed private array: lane kernels from n = 6 (with the
    # stage loop), quad kernels from n = 17 (profiles/r03_lane_rkf_park.jsonl, r03_quad_rkf_park.jsonl); never the wave kernels
    # (lane kernels: the stage-loop body -- from n = 4 -- IS the parked one since round 4, the option does not apply there)
    for name, want in (("spring", _abi.OFF), ("chain4", _abi.ON), ("threeBodyPolar", _abi.ON), ("chain8", _abi.ON), ("chain16", _abi.ON), ("chain20", _abi.ON), ("chain33", _abi.OFF)):
        assert api.system_from_spec(E.get(name)).options(65536)["rkf_park"] == want, name
    assert api.system_from_spec(E.get("chain14")).options(8192)["rkf_park"] == _abi.OFF          # (the quad module of a small ensemble: n < 17)
    # lane mapping: rkf_park follows rkf_body; an explicit value that contradicts it is REFUSED (round 5; it used to be ignored)
    with pytest.raises(api.HamkError, match="rkf_park = OFF cannot be honoured"):
        api.system_from_spec(E.get("chain8"), {"rkf_park": _abi.OFF})
    with pytest.raises(api.HamkError, match="rkf_park = ON cannot be honoured"):
        api.system_from_spec(E.get("doublePendulum"), {"rkf_park": _abi.ON})
    assert api.system_from_spec(E.get("chain8"), {"rkf_park": _abi.ON}).options()["rkf_park"] == _abi.ON
    assert api.system_from_spec(E.get("chain20"), {"rkf_park": _abi.OFF}).options()["rkf_park"] == _abi.OFF
    s = api.system_from_spec(E.get("chain20"))
    assert s.options()["mapping"] == _abi.MAP_QUAD and s.lanes_per_trajectory == 4 and "hamk_quad.hpp" in s.source
    assert s.num_device_functions == 9                        # the eight kernels of the path + the self-check's scribble kernel
    assert api.system_from_spec(E.get("chain33")).options()["mapping"] == _abi.MAP_WAVE


def test_systems_with_a_non_positive_inertia_run_on_kernels_that_pivot(hamk_lib, monkeypatch):
    """The reference's `inv` is LU with partial pivoting for EVERY K (Hamilton.hs:321, :381).  The four-lane kernels
    factorise without pivoting, so a system whose K need not be definite never reaches them: n <= 16 stays on the lane
    kernels whatever the ensemble size (solve_spd -> solve_lu per trajectory), n > 16 goes to the wave-cooperative ones
    (solve_pivoted), asking for HAMK_MAP_QUAD by name is HAMK_ERR_UNSUPPORTED with the reason -- never a launch that
    flags every trajectory singular."""
    from hamilton_amd import _abi, api
    monkeypatch.delenv("HAMK_QUAD", raising=False)
    t = api.system_from_spec(E.get("chain12~mixed"))
    assert t.options(8192)["mapping"] == _abi.MAP_LANE and t.options(65536)["mapping"] == _abi.MAP_LANE
    assert api.system_from_spec(E.get("chain12")).options(8192)["mapping"] == _abi.MAP_QUAD        # (positive inertias: as before)
    w = api.system_from_spec(E.get("chain20~mixed"))
    assert w.options()["mapping"] == _abi.MAP_WAVE and w.options(4096)["mapping"] == _abi.MAP_WAVE
    assert "INERTIA_POS = false" in w.source and "solve_pivoted" in open(os.path.join(ROOT, "hamilton_amd", "csrc", "hamk_wave.hpp")).read()
    assert w.num_device_functions == 9
    with pytest.raises(api.HamkError) as e:
        api.system_from_spec(E.get("chain20~mixed"), {"mapping": _abi.MAP_QUAD})
    assert e.value.code == _abi.HAMK_ERR_UNSUPPORTED and "pivot" in str(e.value)
    monkeypatch.setenv("HAMK_QUAD", "1")                       # the test override does not reach such a system either
    assert api.system_from_spec(E.get("chain12~mixed")).options(8192)["mapping"] == _abi.MAP_LANE


def test_environment_overrides_need_the_test_switch(hamk_lib):
    """HAMK_WAVE / HAMK_AD_MODE / ... change what libhamk.so runs ONLY in a process that sets HAMK_TEST_OVERRIDES=1 (this
    suite's conftest does); a host process that merely carries such a variable gets what hamk_options says."""
    import subprocess
    import sys
    prog = ("import sys; sys.path.insert(0, %r)\n"
            "from hamilton_amd import api, examples\n"
            "o = api.system_from_spec(examples.get('spring')).options()\n"
            "print(o['mapping'], o['ad_mode'], o['rk4_body'])\n") % ROOT
    base = {k: v for k, v in os.environ.items() if not k.startswith("HAMK_") or k in ("HAMK_CACHE_DIR",)}
    forced = dict(base, HAMK_WAVE="1", HAMK_AD_MODE="D", HAMK_RK4_LOOP="1")
    plain = subprocess.check_output([sys.executable, "-c", prog], env=base, text=True).split()
    ungated = subprocess.check_output([sys.executable, "-c", prog], env=forced, text=True).split()
    gated = subprocess.check_output([sys.executable, "-c", prog], env=dict(forced, HAMK_TEST_OVERRIDES="1"), text=True).split()
    from hamilton_amd import _abi
    assert ungated == plain == [str(_abi.MAP_LANE), str(_abi.AD_H), str(_abi.BODY_UNROLLED)]
    assert gated == [str(_abi.MAP_WAVE), str(_abi.AD_D), str(_abi.BODY_STAGE_LOOP)]


def test_options_of_another_layout_revision_are_refused(hamk_lib):
    """hamk_options carries its layout revision (HAMK_OPTIONS_VERSION; round 5 removed the dead field wave_blocked): a struct
    filled in against another header -- or never initialised -- is refused instead of being read field-shifted."""
    from hamilton_amd import _abi, api
    o = _abi.HamkOptions()
    hamk_lib.hamk_options_init(ctypes.byref(o))
    assert o.size == 128 and o.version == _abi.OPTIONS_VERSION
    assert "wave_blocked" not in dict(_abi.HamkOptions._fields_)
    spec = E.get("pendulum")
    good = api.system_from_spec(spec, {"mapping": _abi.MAP_LANE})
    assert good.options()["mapping"] == _abi.MAP_LANE
    bad = _abi.HamkOptions(mapping=_abi.MAP_LANE)
    bad.version = 3                                          # e.g. a round-4 struct whose second word was `mapping = QUAD`
    with pytest.raises(api.HamkError, match="layout revision"):
        api.system_from_spec(spec, bad)
