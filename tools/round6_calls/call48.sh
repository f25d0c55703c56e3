#!/bin/bash
# call 48: the double-precision engine's small transfers through the ring copy too -- its GPU tests; the unmodified binary with
# beagleprecision=double on protein 200 x 10 000 and DNA 500 x 20 000, topology fixed, with / without (MBAMD_NO_RING_COPY=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c48; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -k "double or f64 or precision or fp64" 2>&1 | tail -3 | tee gpurun_out/c48/tests.txt
timeout 1500 python - <<'PY' 2>&1 | tee gpurun_out/c48/chains.txt
import os, sys, time
sys.path.insert(0, os.getcwd())
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
def rate(kind, st, tr, lo, hi, env):
    t = {}
    for n in (lo, hi):
        best = 1e9
        for rep in range(2):
            t0 = time.time()
            if kind == "gtr":
                nex = refrun.mcmc_nexus(st, tr, n, beagle="dynamic", fixed_topology=True).replace("beagleprecision=single", "beagleprecision=double")
            else:
                nex = refrun.model_nexus(kind, st, tr, ngen=n, beagle="dynamic", fixed_topology=True, precision="double")
            out, _ = refrun.run_mb(refrun.REF_MB_AMD, nex, env=env)
            assert "Analysis completed" in out, out[-800:]
            best = min(best, time.time() - t0)
        t[n] = best
    return (hi - lo) / (t[hi] - t[lo])
for kind, ntaxa, nsites, ns, lo, hi in (("wag", 200, 10000, 20, 500, 4500), ("gtr", 500, 20000, 4, 1000, 11000)):
    st = mbdata.synthetic_states(ntaxa, nsites, ns, 7, 0.15, 0.0)
    tr = mbtree.random_tree(ntaxa, 3, brlen=0.05)
    for env in ({}, {"MBAMD_NO_RING_COPY": "1"}):
        print(kind, ntaxa, nsites, "double precision, fixed topology", env or "(default)", "%.0f generations/s" % rate(kind, st, tr, lo, hi, env), flush=True)
PY
