"""SURVEY 8(f) row 2 through the drop-in boundary: the eigen-systems of a division on the engine are computed on the device
(mbamdSetRateMatricesFrom: all parts of a codon / covarion model in one asynchronous launch, warm-started from the chain's
current state) instead of MrBayes' GetEigens on the host (reference src/likelihood.c:10476-10804, src/utils.c:11201).  The
binary is the reference with EVERY binding (oracle/Makefile: ref-amd-full: patch_pars.py, patch_reports.py, patch_eigen.py +
the glue under integration/mrbayes/).

The chain must be the unpatched engine-backed binary's chain: same proposals, same accept / reject decisions; the sampled
log-likelihoods agree to 1e-7 relative (eigenvectors are only defined up to sign and order, transition matrices to ~1e-12).

  * CPU (`not gpu`): oracle/_ref/mb_emu_full against oracle/_ref/mb_emu        (TEST-ONLY host emulation)
  * GPU (`gpu`):     oracle/_ref/mb_amd_full against oracle/_ref/mb_amd
"""
import os
import re
import subprocess

import pytest

from mrbayes_amd import data as mbdata
from mrbayes_amd import tree as mbtree
from tools import refrun

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
REL = 1e-7


def _trace(binary, nex, env=None):
    out, _, files = refrun.run_mb(binary, nex, keep=("mk.p", "mc.p"), env=dict({"MBAMD_STATS": "1"}, **(env or {})))
    text = files.get("mk.p") or files.get("mc.p")
    assert text, out[-2500:]
    lines = [l for l in text.splitlines() if l and not l.startswith("[")]
    i = lines[0].split("\t").index("lnLike")
    return out, [float(l.split("\t")[i]) for l in lines[1:]]


def _cases(big):
    st = mbdata.synthetic_states(8 if not big else 20, 50 if not big else 400, 61, 6, 0.15, 0.0)
    tr = mbtree.random_tree(st.shape[0], 12, brlen=0.05)
    yield "codon M3, fixed topology", refrun.model_nexus("m3", st, tr, ngen=400, beagle="dynamic", fixed_topology=True).replace("samplefreq=400", "samplefreq=20")
    st = mbdata.synthetic_states(10 if not big else 40, 300 if not big else 3000, 4, 11, 0.15, 0.02)
    tr = mbtree.random_tree(st.shape[0], 12, brlen=0.05)
    yield "DNA GTR+G4, default moves, 2 chains", refrun.mcmc_nexus(st, tr, 600, beagle="dynamic", nchains=2).replace("samplefreq=600", "samplefreq=30")
    st = mbdata.synthetic_states(9, 120 if not big else 900, 20, 5, 0.15, 0.03)
    tr = mbtree.random_tree(9, 12, brlen=0.05)
    yield "protein, mixed amino-acid model (the rate matrix changes)", refrun.model_nexus("wag", st, tr, ngen=400, beagle="always").replace(
        "prset aamodelpr=fixed(wag);", "prset aamodelpr=mixed;").replace("samplefreq=400", "samplefreq=20")


def _check(plain, full, marker, big):
    for name, nex in _cases(big):
        _, t0 = _trace(plain, nex)
        out, t1 = _trace(full, nex)
        assert marker in out
        m = re.search(r"mbamd eigen: (\d+) rate-matrix sets decomposed on the device", out)
        assert m and int(m.group(1)) > 3, (name, out[-800:])
        assert len(t0) == len(t1) and len(t0) > 10
        worst = max(abs(a - b) / abs(a) for a, b in zip(t0, t1))
        assert worst <= REL, (name, worst, t0[-3:], t1[-3:])
        out, t2 = _trace(full, nex, env={"MBAMD_DEVICE_EIGEN": "0"})       # switched off: the host's GetEigens again
        assert "mbamd eigen:" not in out
        assert max(abs(a - b) / abs(a) for a, b in zip(t0, t2)) <= 1e-12


def test_device_eigen_binding_on_emulated_engine():
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference sources not present (build container only)")
    from tests.hostemu import build_emu
    build_emu.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/mb_emu", "_ref/mb_emu_full"], stdout=subprocess.DEVNULL)
    # (the emulation runs a workgroup's threads as fibers: the glue is what this test is about, so the solver keeps its 256-thread
    #  launch here -- the 1 024-thread one is emulated by tests/test_engine_hostemu.py::test_device_eigen*)
    os.environ["MBAMD_EIGEN_256"] = "1"
    try:
        _check(os.path.join(REF, "mb_emu"), os.path.join(REF, "mb_emu_full"), "mbamd", big=False)
    finally:
        os.environ.pop("MBAMD_EIGEN_256", None)


@pytest.mark.gpu
def test_device_eigen_binding_on_mi355x():
    if not (os.path.exists(os.path.join(REF, "mb_amd_full")) and os.path.exists(os.path.join(REF, "mb_amd"))):
        pytest.skip("oracle/_ref/mb_amd_full was not built (needs the reference sources at build time)")
    _check(os.path.join(REF, "mb_amd"), os.path.join(REF, "mb_amd_full"), "mbamd HIP gfx950", big=True)
