#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() {  # label, config, env...
  local label=$1 cfg=$2; shift 2
  env "$@" MBAMD_BENCH_NO_ASSERT=1 timeout 300 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-mcmc 2>/tmp/exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-28s %s  value %9.0f  ms/step %.4f  partials %.4f  all %.4f' % ('$label', '$cfg', d['value'], d['ms_per_step'], r['partials_kernel_ms_per_step'], r['all_kernels_ms_per_step']))
" || { echo "$label $cfg FAILED"; tail -5 /tmp/exp.err; }
}
{
for cfg in c5 c3; do
  for v in tw32 tw32_nofetch tw32_nomfma tw32_nostore tw16 tw16_d2 tw16_nofetch tw16_nomfma tw16_nostore; do
    run $v $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_$v.so
  done
  run tw16_w1 $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_tw16.so MBAMD_WALK_WAVES=1
  run tw16_nofetch_w1 $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_tw16_nofetch.so MBAMD_WALK_WAVES=1
done
} 2>&1 | tee gpurun_out/exp_tw16_ablate.log
