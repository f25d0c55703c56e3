#!/bin/bash
# round 3 soak: the reference with ALL bindings (device parsimony + reports + eigen: oracle/_ref/mb_amd_full) in long default-mix runs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== DNA 200 x 5000, 100 000 generations x 4 chains, all bindings"
MB_BINARY=$PWD/oracle/_ref/mb_amd_full timeout 900 python tools/soak_mcmc.py gtr 200 5000 100000 4
echo "== codon M3 60 x 2000, 20 000 generations x 2 chains, all bindings (device eigen-solver warm-started at every Q move)"
MB_BINARY=$PWD/oracle/_ref/mb_amd_full timeout 900 python tools/soak_mcmc.py m3 60 2000 20000 2
echo "== protein 100 x 3000, 30 000 generations x 2 chains"
MB_BINARY=$PWD/oracle/_ref/mb_amd_full timeout 900 python tools/soak_mcmc.py wag 100 3000 30000 2
echo "== DNA 200 x 5000, 5 000 generations with the host functions next to every device parsimony call (MBAMD_PARS_CHECK=1)"
MBAMD_PARS_CHECK=1 MB_BINARY=$PWD/oracle/_ref/mb_amd_full timeout 900 python tools/soak_mcmc.py gtr 200 5000 5000 2
echo "== DNA 500 x 20000 in double precision (fp64 tree walk), 3 000 generations"
MB_BINARY=$PWD/oracle/_ref/mb_amd_pars timeout 900 python - <<'PY'
import os, sys, re
sys.path.insert(0, os.getcwd())
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
st = mbdata.synthetic_states(500, 20000, 4, 21, 0.15, 0.02)
tr = mbtree.random_tree(500, 22, brlen=0.05)
nex = refrun.mcmc_nexus(st, tr, 3000, beagle="dynamic", nchains=1).replace("beagleprecision=single", "beagleprecision=double")
out, wall = refrun.run_mb(os.environ["MB_BINARY"], nex, timeout=850, env={"MBAMD_STATS": "1"})
print("completed" if "Analysis completed" in out else "FAILED", "wall %.1f s" % wall)
print("\n".join(l for l in out.splitlines() if re.match(r"\s+3000 -- ", l) or "Impl Name" in l or "rror" in l)[:600])
PY
} 2>&1 | tee gpurun_out/soak3.txt
