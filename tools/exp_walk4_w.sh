#!/bin/bash
# 4-state walk: waves per workgroup (tree parallelism) at C2 and C4 with the round's final kernel
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
run default c2 X=1
for w in 1 2 3 4 6; do run waves$w c2 MBAMD_WALK_WAVES=$w; done
for w in 4; do for sl in 6 8 12; do run waves${w}_slots$sl c2 MBAMD_WALK_WAVES=$w MBAMD_MAX_LDS_SLOTS=$sl; done; done
run default c4 X=1
for w in 2 3; do run waves$w c4 MBAMD_WALK_WAVES=$w; done
} 2>&1 | tee gpurun_out/exp_walk4_w.log
