#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q > gpurun_out/pytest_tw16.log 2>&1; echo "pytest tw16 exit $?" | tee -a gpurun_out/pytest_tw16.log; tail -15 gpurun_out/pytest_tw16.log
run() {  # label, config, env...
  local label=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-mcmc 2>/tmp/exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-28s %s  value %9.0f  ms/step %.4f  partials %.4f  all %.4f' % ('$label', '$cfg', d['value'], d['ms_per_step'], r['partials_kernel_ms_per_step'], r['all_kernels_ms_per_step']))
" || { echo "$label $cfg FAILED"; tail -5 /tmp/exp.err; }
}
{
for cfg in c5 c3; do
  run tw16 $cfg X=1
  run tw16_w4 $cfg MBAMD_WALK_WAVES=4
  run tw16_w1 $cfg MBAMD_WALK_WAVES=1
  run tw32 $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_tw32.so
done
} 2>&1 | tee gpurun_out/exp_tw16.log
for m in wag m3; do timeout 300 python tools/partial_time.py $m 2>&1 | tail -1; done | tee gpurun_out/partial_tw16.log
for m in wag m3; do MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_tw32.so timeout 300 python tools/partial_time.py $m 2>&1 | tail -1; done | tee gpurun_out/partial_tw32.log
