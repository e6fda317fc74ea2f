"""bench.py's host-side helpers that must never cost the bench line: the reference-toolchain probe (and the Haskell bench build it
triggers where GHC + cabal + GSL exist: mechanical, every failure reported, none raised) and the shader-clock sampler."""
import importlib.util
import os
import sys

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("hamk_bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hamk_bench_module"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_toolchain_probe_reports_and_never_raises(hamk_lib, monkeypatch, tmp_path):
    b = _bench()
    r = b.probe_reference_toolchain()
    assert set(t["tool"] for t in r["probed"]) == {"ghc", "cabal", "stack", "gsl-config"}
    if not r["available"]:
        assert "unavailable" in r["note"]
    # the build hook itself, on a box without cabal / without a checkout: a note, not an exception
    monkeypatch.setenv("HAMILTON_SRC", str(tmp_path / "no-such-checkout"))
    out = b.time_reference_haskell(timeout_s=5.0)
    assert "note" in out and "measured" not in out
    src = tmp_path / "hamilton"
    src.mkdir()
    (src / "hamilton.cabal").write_text("name: hamilton\n")
    monkeypatch.setenv("HAMILTON_SRC", str(src))
    monkeypatch.setenv("PATH", str(tmp_path))                     # no cabal on this PATH
    out = b.time_reference_haskell(timeout_s=5.0)
    assert "note" in out and "measured" not in out


def test_the_haskell_bench_sources_are_there():
    d = os.path.join(ROOT, "bindings", "haskell", "bench")
    for f in ("C1.hs", "hamilton-bench.cabal", "cabal.project"):
        assert os.path.exists(os.path.join(d, f)), f
    hs = open(os.path.join(d, "C1.hs")).read()
    # only the reference's PUBLIC API (it builds against an unpatched checkout), and the two figures bench.py reports
    assert "import           Numeric.Hamilton" in hs and "Numeric.Hamilton.HIP" not in hs
    assert "stepham_us_per_call" in hs and "rk4_steps_per_s_one_thread" in hs
    cabal = open(os.path.join(d, "hamilton-bench.cabal")).read()
    assert "main-is:          C1.hs" in cabal and "hamilton" in cabal


def test_clock_sampler_without_a_gpu_is_silent(hamk_lib):
    b = _bench()
    c = b.ClockSampler(0)                                         # no device here: no path, no thread, no figure
    c.start()
    assert c.stop() is None or isinstance(c.stop(), float)


def test_gpus_flag_starts_one_rank_per_gpu(hamk_lib, monkeypatch):
    """`python bench.py --gpus N` with no launcher around it starts N ranks through torch.distributed.run on 127.0.0.1 and passes its own
    arguments on; under a launcher whose WORLD_SIZE disagrees with --gpus it refuses to print a line."""
    import subprocess

    import pytest
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], os.path.join(ROOT, "bench.py"))
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher: --gpus must be the world it was started in
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        b.main()
    assert "WORLD_SIZE=2" in str(e.value.code)
