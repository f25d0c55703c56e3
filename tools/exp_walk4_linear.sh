#!/bin/bash
# 4-state walk: would a buffer-major arena (all waves writing into one moving window) make the store stream faster?  (ablation: wrong results)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() {  # label, config, env...
  local label=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-also --no-mcmc 2>/tmp/exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-28s %s  value %9.0f  ms/step %.4f  partials %.4f  all %.4f  frac %.3f' % ('$label', '$cfg', d['value'], d['ms_per_step'], r['partials_kernel_ms_per_step'], r['all_kernels_ms_per_step'], r['frac']))
" || { echo "$label $cfg FAILED"; tail -3 /tmp/exp.err; }
}
{
python tools/hbm_probe.py 2>&1 | head -2
L=$PWD/build_x/libhmsbeagle_linear.so
for cfg in c4 c2; do for rep in 1 2; do run block_major $cfg X=1; run linear_stores $cfg MBAMD_LIBRARY=$L MBAMD_BENCH_NO_ASSERT=1; done; done
} 2>&1 | tee gpurun_out/exp_walk4_linear.log
