#!/bin/bash
# round 5, call 31: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1) -- step times of the four workloads and the fixed-topology chain
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for v in 0 1; do
  for c in c2 c3 c5; do
    echo "== $c HIP_FORCE_DEV_KERNARG=$v"
    HIP_FORCE_DEV_KERNARG=$v timeout 200 python bench.py --config $c --steps 200 --no-cpu-baseline --no-also --no-mcmc | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f ms/step %.4f all_kernels %.4f partials %.4f frac %.3f' % (d['value'], d['ms_per_step'], r['all_kernels_ms_per_step'], r['partials_kernel_ms_per_step'], r['frac']))"
  done
  timeout 300 python tools/mcmc_walls.py 2000 42000 HIP_FORCE_DEV_KERNARG=$v
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c31.log
