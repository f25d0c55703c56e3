#!/bin/bash
# a longer soak of the fully bound binary: 500 000 generations x 4 chains (DNA), 100 000 x 2 (codon M3), and a double-precision run
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== DNA 200 x 5000, 500 000 generations x 4 chains, all bindings"
MB_BINARY=$PWD/oracle/_ref/mb_amd_full timeout 1500 python tools/soak_mcmc.py gtr 200 5000 500000 4
echo "== codon M3 60 x 2000, 100 000 generations x 2 chains, all bindings"
MB_BINARY=$PWD/oracle/_ref/mb_amd_full timeout 1500 python tools/soak_mcmc.py m3 60 2000 100000 2
echo "== DNA 200 x 5000 in double precision, 100 000 generations x 2 chains (fp64 tree walk for the path lists, device parsimony)"
MB_BINARY=$PWD/oracle/_ref/mb_amd_pars timeout 1500 python - <<'PY'
import os, sys, re
sys.path.insert(0, os.getcwd())
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
st = mbdata.synthetic_states(200, 5000, 4, 21, 0.15, 0.02)
tr = mbtree.random_tree(200, 22, brlen=0.05)
nex = refrun.mcmc_nexus(st, tr, 100000, beagle="dynamic", nchains=2).replace("beagleprecision=single", "beagleprecision=double")
out, wall = refrun.run_mb(os.environ["MB_BINARY"], nex, timeout=1400, env={"MBAMD_STATS": "1"})
print("completed" if "Analysis completed" in out else "FAILED", "wall %.1f s" % wall)
print("\n".join(l for l in out.splitlines() if re.match(r"\s+100000 -- ", l) or "Impl Name" in l or "rror" in l)[:600])
PY
} 2>&1 | tee gpurun_out/soak4.txt
