#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q 2>&1 | tail -3
for cfg in c2 c4; do
for W in ${WAVES:-4 8}; do
  for SL in ${SLOTS:-64}; do
    echo "== $cfg W=$W maxslots=$SL"
    MBAMD_WALK_WAVES=$W MBAMD_MAX_LDS_SLOTS=$SL timeout 300 python bench.py --config $cfg --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
    elif 'amdgpu.ids' not in l: print(l.strip()[:200])
"
  done
done
done 2>&1 | tee gpurun_out/sweep_walk.log
