"""The C++ host mirror (include/hamilton.hpp) over the same C ABI: builds and specialises a
System on CPU; on a GPU reproduces the reference's initial-state facts."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def dp_binary(hamk_lib, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "double_pendulum")
    libdir = os.path.join(ROOT, "hamilton_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "double_pendulum.cpp"), "-o", out,
                           "-L" + libdir, "-lhamk", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_cpp_tracer_generates_the_same_coordinate_map(dp_binary):
    from hamilton_amd import api, examples
    src = subprocess.check_output([dp_binary], text=True)
    py = api.system_from_spec(examples.get("doublePendulum")).source
    body = lambda s: s[s.index("static void coords"):s.index("static A potential")]
    assert body(src) == body(py)                      # both recorders emit the identical tape for f
    assert "HAMK_INSTANTIATE(HamkSys)" in src and "N = 2;" in src and "M = 4;" in src


@pytest.mark.gpu
def test_cpp_host_runs_on_gpu(dp_binary):
    out = subprocess.check_output([dp_binary, "run"], text=True)
    m = re.search(r"hamEqs dq = (\S+) (\S+) dp = (\S+) (\S+)", out)
    dq0, dq1, dp0, dp1 = map(float, m.groups())
    assert abs(dq0) < 1e-15 and abs(dq1) < 1e-15 and abs(dp0 + 10) < 1e-13 and abs(dp1) < 1e-14
    m = re.search(r"stepHam q = (\S+) (\S+) p = (\S+) (\S+)", out)
    q0, q1, p0, p1 = map(float, m.groups())
    assert abs(q0 - 1.5705463267948976) < 1e-9 and abs(p0 + 0.0999999981) < 1e-9   # oracle: test_oracle_golden
    assert "evolveHam rows = 3" in out
    assert "iterateStepHam same = 1 frames = 2" in out
    m = re.search(r"options trig = (\d+) rkf_body = (\d+) gsl_api = (\d+) lanes = (\d+) dq = (\S+)", out)
    assert [int(x) for x in m.groups()[:4]] == [1, 2, 1, 1] and float(m.group(5)) < 1e-12      # same step under other choices
    m = re.search(r"hamiltonian = (\S+)", out)
    assert abs(float(m.group(1)) - 7.5) < 1e-7                                      # H conserved from seInit


@pytest.fixture(scope="module")
def bench_binary(hamk_lib, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "hamk_bench")
    libdir = os.path.join(ROOT, "hamilton_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "hamk_bench.cpp"), "-o", out,
                           "-L" + libdir, "-lhamk", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_cpp_bench_builds_and_refuses_without_gpu(bench_binary, hamk_lib):
    if hamk_lib.hamk_device_count() > 0:
        pytest.skip("a GPU is visible")
    r = subprocess.run([bench_binary, "--batch", "16"], capture_output=True, text=True)
    assert r.returncode == 3 and "HIP device" in r.stderr      # no CPU fallback


@pytest.mark.gpu
def test_cpp_bench_matches_python_host(bench_binary):
    """The compiled host (device-resident ensemble through hamk_device_malloc / hamk_memcpy /
    hamk_gather_batch) and the Python host agree bit for bit on the same seeded trajectories,
    including a two-shard run on one device gathered back in part order."""
    import json
    import numpy as np
    from hamilton_amd import api, examples as E
    spec = E.get("doublePendulum")
    s = api.system_from_spec(spec)
    B, nsteps, launches, warm = 1000, 7, 3, 1
    out = subprocess.check_output([bench_binary, "--batch", str(B), "--nsteps", str(nsteps), "--launches", str(launches),
                                   "--warmup", str(warm), "--dump-first", "5"], text=True)
    line = json.loads(out.splitlines()[0])
    assert line["n_gpus"] == 1 and line["status_flagged"] == 0 and line["value"] > 0
    q, qd = E.sample_config(spec, 0, B)
    ph = api.toPhase(s, api.Config(q, qd))
    h0 = api.hamiltonian(s, ph)
    for _ in range(launches + warm):
        ph = api.rk4Steps(0.01, nsteps, s, ph)
    rows = [l for l in out.splitlines() if l.startswith("traj ")]
    assert len(rows) == 5
    for i, l in enumerate(rows):
        m = re.match(r"traj \d+ q = (\S+) (\S+) p = (\S+) (\S+) H0 = (\S+)", l)
        got = [float(x) for x in m.groups()]
        want = [ph.positions[0, i], ph.positions[1, i], ph.momenta[0, i], ph.momenta[1, i], h0[i]]
        assert got == [float(w) for w in want], (i, got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("force_bcast", [False, True])
def test_rccl_allgather_through_the_c_abi(hamk_lib, monkeypatch, force_bcast):
    """hamk_comm_* on the one GPU of the box: a one-rank RCCL communicator (the real library, loaded on first use) gathers a shard
    [n][B] into [n][B] -- the equal-shard path (one ncclAllGather per row in one group) and the ragged path (one ncclBroadcast per
    (rank, row)).  More than one rank needs more than one device: the driver's 8-GPU box."""
    import numpy as np
    import torch
    from hamilton_amd import _abi
    L = hamk_lib
    if force_bcast:
        monkeypatch.setenv("HAMK_COMM_FORCE_BCAST", "1")
    monkeypatch.setenv("NCCL_SOCKET_IFNAME", "lo")            # (no network on the test box: bootstrap over loopback, no interface probing)
    monkeypatch.setenv("NCCL_IB_DISABLE", "1")
    ident = (ctypes.c_char * _abi.HAMK_COMM_ID_BYTES)()
    assert L.hamk_comm_unique_id(ident) == _abi.HAMK_OK, L.hamk_last_error()
    comm = ctypes.c_void_p()
    assert L.hamk_comm_create(ident, 1, 0, ctypes.byref(comm)) == _abi.HAMK_OK, L.hamk_last_error()
    try:
        n, B = 5, 1237
        src = torch.arange(n * B, dtype=torch.float64, device="cuda").reshape(n, B) * 0.5 - 3.0
        dst = torch.full((n, B), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        Bs = (ctypes.c_int64 * 1)(B)
        rc = L.hamk_comm_allgather_batch(comm, n, Bs, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()))
        assert rc == _abi.HAMK_OK, L.hamk_last_error()
        assert torch.equal(src, dst)
        assert L.hamk_comm_allgather_batch(comm, n, Bs, None, ctypes.c_void_p(dst.data_ptr())) == _abi.HAMK_ERR_INVALID
        Bs[0] = -1
        assert L.hamk_comm_allgather_batch(comm, n, Bs, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr())) == _abi.HAMK_ERR_INVALID
    finally:
        assert L.hamk_comm_destroy(comm) == _abi.HAMK_OK


@pytest.mark.gpu
def test_cpp_bench_one_process_per_gpu_matches_the_single_process_form(bench_binary, tmp_path):
    """tools/hamk_bench --world 1 --rank 0 (the process-per-GPU form of a native host: communicator id through a file, barrier and
    max-over-ranks timing and the final all-gather over RCCL through hamk_comm_*) ends with the same bits as the single-process form."""
    import json
    args = ["--batch", "1000", "--nsteps", "7", "--launches", "3", "--warmup", "1", "--dump-first", "5"]
    one = subprocess.check_output([bench_binary] + args, text=True).splitlines()
    idf = str(tmp_path / "hamk.id")
    # (a box without a network: RCCL's bootstrap is pointed at the loopback interface instead of probing every interface -- that probing
    # was seen to take 90 s on one box and 6 s on another)
    env = dict(os.environ, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1")
    per = subprocess.check_output([bench_binary, "--world", "1", "--rank", "0", "--id-file", idf] + args, text=True, timeout=900, env=env).splitlines()
    line = json.loads(next(l for l in per if l.startswith("{")))         # (RCCL prints its version banner on stdout first)
    assert line["n_gpus"] == 1 and line["status_flagged"] == 0 and line["value"] > 0 and "allgather_ms_rccl" in line
    assert not os.path.exists(idf)
    strip = lambda rows: [re.sub(r" H0 = \S+", "", l) for l in rows if l.startswith("traj ")]
    assert len(strip(per)) == 5 and strip(per) == strip(one)
    r = subprocess.run([bench_binary, "--world", "2", "--rank", "2", "--id-file", idf], capture_output=True, text=True)
    assert r.returncode == 2


@pytest.fixture(scope="module")
def c_binary(hamk_lib, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("c") / "abi_smoke")
    libdir = os.path.join(ROOT, "hamilton_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", out,
                           "-L" + libdir, "-lhamk", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_header_is_plain_c99(c_binary):
    """include/hamk.h compiles as strict C99 and a hand-written tape specialises from a C client."""
    out = subprocess.check_output([c_binary], text=True)
    assert "System 2 1" in out


@pytest.mark.gpu
def test_c_client_runs_on_gpu(c_binary):
    import math
    out = subprocess.check_output([c_binary, "run"], text=True)
    m = re.search(r"hamEqs dq = (\S+) dp = (\S+) status = (\S+)", out)
    dq, dp, st = float(m.group(1)), float(m.group(2)), int(m.group(3))
    assert st == 0 and abs(dq) < 1e-16 and abs(dp + math.sin(0.3)) < 1e-14         # K = 1: dq = p, dp = -dU/dtheta
