#!/bin/bash
# buffer-major 4-state arena: the whole GPU suite, then C4 / C2 (and the partial-update latency) with the new layout
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() {  # label, config, env...
  local label=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-also --no-mcmc 2>/tmp/exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-28s %s  value %9.0f  ms/step %.4f  partials %.4f  all %.4f  frac %.3f  lnL %s' % ('$label', '$cfg', d['value'], d['ms_per_step'], r['partials_kernel_ms_per_step'], r['all_kernels_ms_per_step'], r['frac'], d['config']['lnL']))
" || { echo "$label $cfg FAILED"; tail -3 /tmp/exp.err; }
}
{
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do run buffer_major c4 X=1; done
for rep in 1 2; do run buffer_major c2 X=1; done
timeout 300 python tools/partial_time.py gtr 300
} 2>&1 | tee gpurun_out/exp_walk4_arena.log
