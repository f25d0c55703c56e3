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
L=$PWD/build_x/libhmsbeagle_glinear.so
for cfg in c3 c5; do for rep in 1 2; do run tile_major $cfg X=1; run linear_stores $cfg MBAMD_LIBRARY=$L MBAMD_BENCH_NO_ASSERT=1; done; done
echo "== protein 200 x 160000 (tile-major, then linear stores)"
timeout 400 python tools/scale_time.py wag 200 160000; MBAMD_LIBRARY=$L timeout 400 python tools/scale_time.py wag 200 160000
} 2>&1 | tee gpurun_out/exp_walkg_linear.log
