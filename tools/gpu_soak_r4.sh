#!/bin/bash
# Round-4 soak of the fully bound binary (every binding, incl. standard data): long default-mix runs must complete; double-precision runs
# exercise the queued operation lists and the rebuilt four-state walk
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== DNA 200 x 5000, 200 000 generations x 4 chains, all bindings"
MB_BINARY=$PWD/oracle/_ref/mb_amd_full timeout 900 python tools/soak_mcmc.py gtr 200 5000 200000 4
echo "== codon M3 60 x 2000, 50 000 generations x 2 chains, all bindings"
MB_BINARY=$PWD/oracle/_ref/mb_amd_full timeout 900 python tools/soak_mcmc.py m3 60 2000 50000 2
echo "== standard data 100 taxa x 2000 characters (2 ... 6 states, gamma), 50 000 generations x 2 chains, default moves, all bindings"
timeout 900 python - <<'PY'
import os, sys, re
sys.path.insert(0, os.getcwd())
from tests import std_cases
from tools import refrun
kw = dict(std_cases.BIG, ngen=50000)
nex = std_cases.synthetic_nexus(beagle="dynamic", **kw).replace("nchains=1", "nchains=2").replace(" startvals tau=t V=t;\n", "")
out, wall = refrun.run_mb(os.path.join(os.getcwd(), "oracle", "_ref", "mb_amd_full"), nex, timeout=850, env={"MBAMD_STATS": "1"})
print("completed" if "Analysis completed" in out else "FAILED", "wall %.1f s" % wall)
print("\n".join(l for l in out.splitlines() if re.match(r"\s+50000 -- ", l) or "standard data" in l or "rror" in l)[:700])
PY
for prec in double; do
echo "== DNA 200 x 5000 in double precision, 100 000 generations x 2 chains (fp64 tree walk, device parsimony)"
MB_BINARY=$PWD/oracle/_ref/mb_amd_pars timeout 900 python - <<'PY'
import os, sys, re
sys.path.insert(0, os.getcwd())
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
st = mbdata.synthetic_states(200, 5000, 4, 21, 0.15, 0.02)
tr = mbtree.random_tree(200, 22, brlen=0.05)
nex = refrun.mcmc_nexus(st, tr, 100000, beagle="dynamic", nchains=2).replace("beagleprecision=single", "beagleprecision=double")
out, wall = refrun.run_mb(os.environ["MB_BINARY"], nex, timeout=850, env={"MBAMD_STATS": "1"})
print("completed" if "Analysis completed" in out else "FAILED", "wall %.1f s" % wall)
print("\n".join(l for l in out.splitlines() if re.match(r"\s+100000 -- ", l) or "Impl Name" in l or "rror" in l)[:600])
PY
echo "== codon M3 60 x 2000 in double precision, 20 000 generations (queued per-part lists on the fp64 level kernels)"
timeout 900 python - <<'PY'
import os, sys, re
sys.path.insert(0, os.getcwd())
from tools import refrun
import subprocess
r = subprocess.run([sys.executable, "tools/soak_mcmc.py", "m3", "60", "2000", "20000", "1"], capture_output=True, text=True,
                   env=dict(os.environ, MB_BINARY=os.path.join(os.getcwd(), "oracle", "_ref", "mb_amd"), MB_PRECISION="double"))
print((r.stdout + r.stderr)[-900:])
PY
done
} 2>&1 | tee gpurun_out/soak_r4.txt
