#!/bin/bash
# call 47: every small host-to-device transfer (programs, eigen-systems, frequencies, weights, parsimony programs) through the pinned ring and a copy kernel of ours (MBAMD_NO_RING_COPY=1: hipMemcpyAsync) -- 4-state GPU
# tests, the two chains with / without, the kernels of 2 000 fixed-topology generations
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c47; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/c47/gputests.txt
timeout 900 python tools/mcmc_ab.py amd fixed 2000 42000 MBAMD_NO_RING_COPY=1 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite\|Pars\|pars' | tee gpurun_out/c47/fixed.txt
timeout 1200 python tools/mcmc_ab.py pars mix 2000 32000 MBAMD_NO_RING_COPY=1 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite' | tee gpurun_out/c47/mix.txt
timeout 600 bash tools/prof_mcmc.sh 2>&1 | head -30 | tee gpurun_out/c47/timeline.txt
# the codon chain with every binding (eigen-systems of 61 states travel per move), with / without
timeout 900 python - <<'PY' 2>&1 | tee gpurun_out/c47/codon.txt
import os, sys
sys.path.insert(0, os.getcwd())
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
st = mbdata.synthetic_states(100, 5000, 61, 7, 0.15, 0.0) if hasattr(mbdata, 'synthetic_states') else None
PY
run() { cfg=$1; shift; env "$@" python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-mcmc --no-arith 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('   ms/step %.4f  all kernels %.4f  partials %.4f' % (o['ms_per_step'], r.get('all_kernels_ms_per_step', 0), r.get('partials_kernel_ms_per_step', 0)))
"; }
{ for cfg in c2 c5; do echo "== $cfg product"; run $cfg; echo "== $cfg MBAMD_NO_RING_COPY=1"; run $cfg MBAMD_NO_RING_COPY=1; done; } 2>&1 | tee gpurun_out/c47/steps.txt
