#!/bin/bash
# Round 4, GPU call 53: the chain kernel under the stamp build (non-waiting stamps: where a wave's time passes per operation), protein double precision chain
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call53.log; : > $OUT
python - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
for case, kind, nst in (("bench_c3", "wag", 20), ("bench_c5", "m3", 61)):
    g = json.load(open(os.path.join(bench.GOLD, case + ".json")))
    s = g["synthetic"]
    st = mbdata.synthetic_states(s["ntaxa"], s["nsites"], nst, s["seed"], s["p_mut"], s["p_gap"])
    tr = mbtree.parse_newick(g["newick"])
    open("/tmp/%s.nex" % case, "w").write(refrun.model_nexus(kind, st, tr, ngen=600, beagle="dynamic", fixed_topology=True, precision="double", fname="/tmp/" + case))
PY
mkdir -p /tmp/st; ln -sf $PWD/build_x/libhmsbeagle_stamps.so /tmp/st/libhmsbeagle.so
for c in bench_c3 bench_c5; do
  echo "== $c" | tee -a $OUT
  LD_LIBRARY_PATH=/tmp/st:${LD_LIBRARY_PATH:-} timeout 200 oracle/_ref/mb_amd /tmp/$c.nex 2>&1 | grep "stamps\|Analysis completed" | tee -a $OUT
done
