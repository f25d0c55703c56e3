#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
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
L=$PWD/build_x/libhmsbeagle_linexps.so
for rep in 1 2; do run buffer_major c4 X=1; run plus_linear_exps c4 MBAMD_LIBRARY=$L MBAMD_BENCH_NO_ASSERT=1; done
} 2>&1 | tee gpurun_out/exp_walk4_linexps.log
