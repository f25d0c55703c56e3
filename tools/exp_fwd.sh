#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q 2>&1 | tail -3 | tee gpurun_out/pytest_fwd.log
run() {  # label, config, env...
  local label=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-also --no-mcmc --no-mpi 2>/tmp/exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-28s %s  value %9.0f  ms/step %.4f  partials %.4f  all %.4f  frac %.3f' % ('$label', '$cfg', d['value'], d['ms_per_step'], r['partials_kernel_ms_per_step'], r['all_kernels_ms_per_step'], r['frac']))
" || { echo "$label $cfg FAILED"; tail -3 /tmp/exp.err; }
}
{
for cfg in c4 c2; do
  run nofwd $cfg MBAMD_WALK_NO_FORWARD=1
  run fwd $cfg X=1
  run fwd_w2 $cfg MBAMD_WALK_WAVES=2
  run fwd_w3 $cfg MBAMD_WALK_WAVES=3
  run fwd_w4 $cfg MBAMD_WALK_WAVES=4
  run nofwd_w2 $cfg MBAMD_WALK_NO_FORWARD=1 MBAMD_WALK_WAVES=2
  run fwd_w1 $cfg MBAMD_WALK_WAVES=1
done
} 2>&1 | tee gpurun_out/exp_fwd.log
for m in gtr; do timeout 300 python tools/partial_time.py $m 2>&1 | tail -1; MBAMD_WALK_NO_FORWARD=1 timeout 300 python tools/partial_time.py $m 2>&1 | tail -1; done | tee gpurun_out/partial_fwd.log
