"""The single-node MPI shim (integration/mpi_shim, SURVEY 8(e) option (i)): its own self-test, and the reference's MPI
build (-DMPI_ENABLED, chains spread over ranks, src/mcmc.c:18331-18384) running on it -- with the native kernels, the
host-emulated engine and (-m gpu) the engine on an MI355X."""
import os
import subprocess
import tempfile

import pytest

from tools import refrun
from tests.test_mrbayes_dropin import _case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "integration", "mpi_shim")
REF = os.path.join(ROOT, "oracle", "_ref")


def _samples(cmd, nexus_text, timeout=900):
    """run a MrBayes command line on a NEXUS text; -> (stdout, the sampled parameter rows of the .p file)"""
    with tempfile.TemporaryDirectory() as wd:
        with open(os.path.join(wd, "run.nex"), "w") as fh:
            fh.write(nexus_text)
        res = subprocess.run(cmd + ["run.nex"], cwd=wd, capture_output=True, text=True, timeout=timeout)
        out = res.stdout + res.stderr
        assert res.returncode == 0 and "Analysis completed" in out, out[-2000:]
        with open(os.path.join(wd, "mc.p")) as fh:
            rows = [l for l in fh.read().split("\n") if l and not l.startswith("[ID")]
        with open(os.path.join(wd, "mc.t")) as fh:
            trees = [l for l in fh.read().split("\n") if l.strip().startswith("tree gen")]
        return out, rows, trees


def _nexus(beagle, ngen=1500, nchains=4):
    st, _ = _case(16, 300, 0.03)
    nex = refrun.mcmc_nexus(st, None, ngen, beagle=beagle, nchains=nchains)
    return nex.replace("samplefreq=%d" % ngen, "samplefreq=100")


def _build(*targets):
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference sources not present (build container only)")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")] + list(targets), stdout=subprocess.DEVNULL)


def test_shim_self_check(tmp_path):
    exe, run = str(tmp_path / "shim_check"), str(tmp_path / "mbamd_mpirun")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I", SHIM, os.path.join(ROOT, "tests", "mpi_shim", "shim_check.c"),
                           os.path.join(SHIM, "mbamd_mpi.c"), "-o", exe])
    subprocess.check_call(["gcc", "-O2", "-Wall", os.path.join(SHIM, "mbamd_mpirun.c"), "-o", run])
    for n in (1, 2, 3, 8):
        res = subprocess.run([run, "-n", str(n), exe], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0 and "mpi shim ok: %d ranks" % n in res.stdout, res.stdout + res.stderr
    # a rank that dies takes the job down with a non-zero status instead of hanging the others
    res = subprocess.run([run, "-n", "2", "/bin/sh", "-c", 'test "$MBAMD_MPI_RANK" = 1 && exit 3; exec sleep 30'], capture_output=True, timeout=60)
    assert res.returncode == 3


def _lnl_column(rows):
    return [float(r.split("\t")[1]) for r in rows[1:]]


def _check_mpi_binary(launcher, binary, marker, ranks=(2, 4), ngen=300):
    """The reference's MPI build seeds every rank differently (seed = globalSeed + rank + 1, src/mcmc.c:2330-2333), so an
    MPI run is not the serial chain.  What must hold: it completes on N ranks with the swap statistics gathered on rank 0;
    the same job twice gives the same samples (deterministic message order and rank-ordered reductions); and with the
    engine switched on or off inside the same binary and rank count the chains coincide while rounding has not yet made
    them part (the first sampled generations, to the engine's stated tolerance)."""
    for n in ranks:
        cmd = [launcher, "-n", str(n), binary]
        nex = _nexus("dynamic" if marker else None, ngen=ngen)
        out, rows, trees = _samples(cmd, nex)
        assert "Chain swap information" in out and len(rows) >= 4
        if marker:
            assert marker in out
        if n != ranks[0]:
            continue
        out2, rows2, trees2 = _samples(cmd, nex)
        assert rows2 == rows and trees2 == trees, n
        if marker:
            short = _nexus("dynamic", ngen=20).replace("samplefreq=100", "samplefreq=2")
            _, on, _ = _samples(cmd, short)
            _, off, _ = _samples(cmd, short.replace("set usebeagle=yes", "set usebeagle=no"))
            a, b = _lnl_column(on)[:4], _lnl_column(off)[:4]
            assert all(abs(x - y) <= 1e-5 * abs(y) for x, y in zip(a, b)), (a, b)


def test_reference_mpi_build_on_the_shim():
    _build("_ref/mb_mpi", "_ref/mbamd_mpirun")
    _check_mpi_binary(os.path.join(REF, "mbamd_mpirun"), os.path.join(REF, "mb_mpi"), None)
    # one rank of the MPI build is the serial binary with the rank's seed offset: both start from the same state
    _build("_ref/mb")
    nex = _nexus(None, ngen=100)
    _, serial, _ = _samples([os.path.join(REF, "mb")], nex)
    _, one, _ = _samples([os.path.join(REF, "mbamd_mpirun"), "-n", "1", os.path.join(REF, "mb_mpi")], nex)
    assert _lnl_column(serial)[0] == _lnl_column(one)[0]


def test_mpi_build_on_emulated_engine():
    from tests.hostemu import build_emu
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference sources not present (build container only)")
    build_emu.build()
    _build("_ref/mb_emu_mpi", "_ref/mbamd_mpirun")
    _check_mpi_binary(os.path.join(REF, "mbamd_mpirun"), os.path.join(REF, "mb_emu_mpi"), "mbamd", ranks=(2,), ngen=300)


@pytest.mark.gpu
def test_mpi_build_on_mi355x():
    """two and four ranks sharing the box's one GPU (with more GPUs the reference's rule spreads the ranks over them:
    resource (instance + rank) mod count, src/mbbeagle.c:201-207)"""
    need = [os.path.join(REF, n) for n in ("mb_amd_mpi", "mbamd_mpirun")]
    if not all(os.path.exists(p) for p in need):
        pytest.skip("oracle/_ref/mb_amd_mpi was not built (needs the reference sources at build time)")
    _check_mpi_binary(need[1], need[0], "mbamd HIP gfx950", ranks=(2, 4), ngen=1000)
