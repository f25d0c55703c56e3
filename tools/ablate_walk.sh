#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { MBAMD_VERBOSE=1 timeout 300 python bench.py --config ${CFG:-c2} --steps 30 --warmup 3 --no-cpu-baseline --no-also --no-mcmc 2>&1 | python -c "
import sys,json
seen=False
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
    elif 'walk plan' in l and not seen: print(l.strip()[:200]); seen=True
"; }
{
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q 2>&1 | tail -3
for W in ${WAVES:-4 5 6 8}; do for cfg in c2 c4; do echo "== $cfg total waves=$W"; CFG=$cfg MBAMD_WALK_WAVES=$W run; done; done
} 2>&1 | tee gpurun_out/ablate_walk.log
