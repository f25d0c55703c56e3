#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { timeout 300 python bench.py --config ${CFG:-c2} --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
    elif 'amdgpu.ids' not in l: print(l.strip()[:300])
"; }
{
for AB in 12 28 44 60; do echo "== c2 ablate=$AB (4=no op 8=no barrier 16=no input request 32=no row request)"; MBAMD_WALK_ABLATE=$AB run; done
echo "== trace"
MBAMD_WALK_ABLATE=64 python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import synthetic_division
div = synthetic_division("gtr", 500, 20000, seed=7, tree_seed=3)
bd = lk.BeagleDivision(div, bg.library())
for i in range(3):
    bd.TouchAllTreeNodes(0); bd.LogLike(0); bd.AcceptMove(0)
# the cumulative buffer of chain 0 now holds the cycle trace in its first ints (block 0)
sc = bd.inst.get_scale_factors(bd.siteScalerIndex[0])
raw = np.rint(sc / np.log(2.0)).astype(np.int64)[:64]
print("raw ints", raw[:16])
PY
} 2>&1 | tee gpurun_out/ablate_walk.log
