#!/bin/bash
# round 5, call 26: 61 states with FOUR bins per workgroup and so few LDS slots that two workgroups share a CU (2 working waves per SIMD
# without the row split) -- W x slots sweep at C5
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "2 4" "4 1" "4 2" "4 3" "2 2" "2 1"; do
  set -- $v
  echo "== MBAMD_WALK_WAVES=$1 MBAMD_MAX_LDS_SLOTS=$2"
  MBAMD_WALK_WAVES=$1 MBAMD_MAX_LDS_SLOTS=$2 timeout 200 python bench.py --config c5 --steps 100 --no-cpu-baseline --no-also --no-mcmc | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f ms/step %.4f all_kernels %.4f partials %.4f frac %.3f' % (d['value'], d['ms_per_step'], r['all_kernels_ms_per_step'], r['partials_kernel_ms_per_step'], r['frac']))"
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c26.log
