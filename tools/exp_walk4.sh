#!/bin/bash
# GPU experiment round for the 4-state walk: variants (build_x/libhmsbeagle_<name>.so) x environment switches -> one line each
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
for cfg in c4 c2; do
  run base $cfg X=1
  for d in 4 8 16; do run tipahead$d $cfg MBAMD_WALK_TIP_AHEAD=$d; done
  for v in tip0 mat0 nostore tip0mat0; do
    [[ -f build_x/libhmsbeagle_$v.so ]] && run abl_$v $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_$v.so MBAMD_BENCH_NO_ASSERT=1
  done
done
} 2>&1 | tee gpurun_out/exp_walk4.log
