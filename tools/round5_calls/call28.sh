#!/bin/bash
# round 5, call 28: k_walkg / k_walkg2 moved to csrc/ (written against device primitives the host emulation also implements), the
# 16-pattern-tile branches removed -- GPU parity of the general-state tests, step time at C5 / C3 unchanged?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "codon or protein or general_state or state_counts or golden or matrices or config3 or config5 or bench_workloads or pair or covarion" 2>&1 | tail -3
for c in c5 c3; do
  for v in "" 1; do
    echo "== $c MBAMD_WALKG_PAIR=$v"
    env ${v:+MBAMD_WALKG_PAIR=1} timeout 200 python bench.py --config $c --steps 200 --no-cpu-baseline --no-also --no-mcmc | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f ms/step %.4f all_kernels %.4f partials %.4f frac %.3f' % (d['value'], d['ms_per_step'], r['all_kernels_ms_per_step'], r['partials_kernel_ms_per_step'], r['frac']))"
  done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c28.log
