#!/usr/bin/env python3
"""Whole-MCMC rate of a Metropolis-coupled analysis (nruns x nchains chains, MrBayes' default move mix, DNA 500 x 20 000) with
the chains spread over MPI ranks -- the reference's own parallel design (src/mcmc.c:18331-18384) on integration/mpi_shim.
On a one-GPU box the ranks share the GPU (their host phases overlap each other's kernels); on an 8-GPU node the reference's
rule puts rank r on GPU r (src/mbbeagle.c:201-207).   python tools/mcmc_mpi.py [nruns nchains]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                               # noqa: E402
from mrbayes_amd import data as mbdata, tree as mbtree     # noqa: E402
from tools import refrun                                   # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def wall(cmd, text):
    with tempfile.TemporaryDirectory() as wd:
        with open(os.path.join(wd, "run.nex"), "w") as fh:
            fh.write(text)
        t0 = time.time()
        res = subprocess.run(cmd + ["run.nex"], cwd=wd, capture_output=True, text=True, timeout=3000)
        dt = time.time() - t0
        if res.returncode != 0 or "Analysis completed" not in res.stdout:
            raise SystemExit((res.stdout + res.stderr)[-2000:])
        return dt


def rate(cmd, nex_of, lo, hi):
    a, b = wall(cmd, nex_of(lo)), wall(cmd, nex_of(hi))
    return (hi - lo) / max(b - a, 1e-9)


def main():
    nruns, nchains = (int(x) for x in (sys.argv[1:3] if len(sys.argv) >= 3 else (2, 4)))
    with open(os.path.join(bench.GOLD, bench.CONFIGS["c2"][0] + ".json")) as fh:
        gold = json.load(fh)
    sy = gold["synthetic"]
    st = mbdata.synthetic_states(sy["ntaxa"], sy["nsites"], 4, sy["seed"], sy["p_mut"], sy["p_gap"])
    tr = mbtree.parse_newick(gold["newick"])

    def nex_of(beagle):
        def f(ngen):
            return refrun.mcmc_nexus(st, tr, ngen, beagle=beagle, nchains=nchains).replace("nruns=1", "nruns=%d" % nruns)
        return f
    out = {"workload": "bench_c2, default move mix, nruns=%d nchains=%d (%d chains)" % (nruns, nchains, nruns * nchains), "unit": "generations/s"}
    out["engine_serial"] = rate([os.path.join(REF, "mb_amd_pars")], nex_of("dynamic"), 300, 1800)
    for r in (2, 4, 8):
        if r > nruns * nchains:
            break
        out["engine_mpi_%d_ranks" % r] = rate([os.path.join(REF, "mbamd_mpirun"), "-n", str(r), os.path.join(REF, "mb_amd_mpi_pars")],
                                                nex_of("dynamic"), 300, 1800 * min(r, 4))
    if os.environ.get("MCMC_MPI_CPU", "1") != "0":
        out["reference_cpu_serial"] = rate([os.path.join(REF, "mb")], nex_of(None), 4, 12)
    import torch
    out["gpus"] = torch.cuda.device_count()
    out["host_cores"] = os.cpu_count()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
