"""The reference-side patch of the Haskell shim (bindings/haskell/hamilton-hip.patch) applies cleanly to
the reference tree -- where that tree exists (this container; the GPU box does not have it: skipped)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")
def test_patch_applies_to_the_reference_tree(tmp_path):
    work = tmp_path / "ref"
    os.makedirs(work)
    shutil.copytree(os.path.join(REF, "src"), work / "src")
    shutil.copy(os.path.join(REF, "hamilton.cabal"), work / "hamilton.cabal")
    patch = os.path.join(ROOT, "bindings", "haskell", "hamilton-hip.patch")
    r = subprocess.run(["patch", "-p1", "--fuzz=0", "-i", patch], cwd=work, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    src = open(work / "src" / "Numeric" / "Hamilton.hs").read()
    assert "_sysHip" in src and "rk4StepsBatch" in src and "HIP.traceSystem" in src
    cabal = open(work / "hamilton.cabal").read()
    assert "Numeric.Hamilton.HIP" in cabal and "extra-libraries:  hamk" in cabal
    # purely additive: every line of the reference module survives (two are re-indented in the cabal file only)
    ref_lines = open(os.path.join(REF, "src", "Numeric", "Hamilton.hs")).read().splitlines()
    new_lines = set(src.splitlines())
    assert all(l in new_lines for l in ref_lines)


def test_shim_declares_every_entry_point_it_needs():
    """Each `foreign import` of the shim names a function include/hamk.h declares."""
    import re
    hs = open(os.path.join(ROOT, "bindings", "haskell", "Numeric", "Hamilton", "HIP.hs")).read()
    header = open(os.path.join(ROOT, "include", "hamk.h")).read()
    names = re.findall(r'foreign import ccall (?:safe |unsafe )?"&?(hamk_\w+)"', hs)
    assert len(names) >= 20
    for n in names:
        assert re.search(r"\b" + n + r"\s*\(", header), n
