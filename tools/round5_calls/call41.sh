#!/bin/bash
# round 5, call 41: k_pathg (a move's root-ward path at 20 / 61 states on two waves: sibling factors ahead of the chain) -- GPU parity,
# the kernels of the codon and protein chains with it and without (MBAMD_NO_PATHG=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_mrbayes_dropin.py tests/test_eigen_binding.py tests/test_reports_dropin.py -x -q -m gpu -k "partial or reject or dynamic or codon or protein or wag or m3 or eigen or general or covarion or reports" 2>&1 | tail -3
python - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
for case, kind, ns, ngen in (("bench_c5", "m3", 61, 1500), ("bench_c3", "wag", 20, 3000)):
    with open(os.path.join(bench.GOLD, case + ".json")) as fh:
        g = json.load(fh)
    s = g["synthetic"]
    st = mbdata.synthetic_states(s["ntaxa"], s["nsites"], ns, s["seed"], s["p_mut"], s["p_gap"])
    tr = mbtree.parse_newick(g["newick"])
    open("/tmp/%s.nex" % kind, "w").write(refrun.model_nexus(kind, st, tr, ngen=ngen, beagle="dynamic", fixed_topology=True))
PY
for kind in m3 wag; do
  for v in "" 1; do
    rm -rf /tmp/prof_$kind; (cd /tmp && env ${v:+MBAMD_NO_PATHG=1} timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$kind -o s -- $GRAFT_REPO_ROOT/oracle/_ref/mb_amd_full $kind.nex > /tmp/$kind.log 2>&1)
    db=$(find /tmp/prof_$kind -name "*.db" | head -1)
    echo "== $kind MBAMD_NO_PATHG=$v"; grep 'Analysis used\|     [0-9]* -- ' /tmp/$kind.log | tail -2
    python tools/rocpd_summary.py $db | cut -c1-170 | grep 'k_pathg\|k_walkg\|Name' | head -6
  done
done 2>&1 | tee gpurun_out/r5c41.log
