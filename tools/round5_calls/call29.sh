#!/bin/bash
# round 5, call 29: every matrix-core kernel now written against device primitives (the host emulation runs them too), wave
# barriers at the intra-wave LDS turn-arounds of the level kernels -- the whole GPU suite, and the step times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5c29_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c29_pytest.log
tail -4 gpurun_out/r5c29_pytest.log
for c in c5 c3; do
  for v in "" 1; do
    echo "== $c MBAMD_NO_WALKG=$v"
    env ${v:+MBAMD_NO_WALKG=1} timeout 200 python bench.py --config $c --steps 100 --no-cpu-baseline --no-also --no-mcmc | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f ms/step %.4f all_kernels %.4f partials %.4f frac %.3f' % (d['value'], d['ms_per_step'], r['all_kernels_ms_per_step'], r['partials_kernel_ms_per_step'], r['frac']))"
  done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c29.log
timeout 300 python tools/f64_bench.py 2>&1 | tail -8 | tee -a gpurun_out/r5c29.log
