#!/bin/bash
# k_walkg: exponent / tip-state bytes through the scalar path (product) against vector loads (variant), same box
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
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -2
for cfg in c3 c5; do
for rep in 1 2; do
run scalar_tiny $cfg X=1
run vector_tiny $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_vectiny.so
done
done
timeout 300 python tools/states_time.py 8 200 10000 | head -1
timeout 300 python tools/states_time.py 16 200 10000 | head -1
timeout 300 python tools/partial_time.py wag 300
timeout 300 python tools/partial_time.py m3 300
} 2>&1 | tee gpurun_out/exp_walkg5.log
