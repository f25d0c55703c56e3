#!/bin/bash
# 61-state walk: operand fetch two chunks ahead (one wave per SIMD) against the product's one chunk ahead (two per SIMD)
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
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -2
run base c5 X=1
run base_nospread c5 MBAMD_WALK_NO_SPREAD=1
L=$PWD/build_x/libhmsbeagle_d2.so
run d2 c5 MBAMD_LIBRARY=$L
run d2_nospread c5 MBAMD_LIBRARY=$L MBAMD_WALK_NO_SPREAD=1
run d2_w1 c5 MBAMD_LIBRARY=$L MBAMD_WALK_WAVES=1
run d2_w4 c5 MBAMD_LIBRARY=$L MBAMD_WALK_WAVES=4
run base c3 X=1
run base c2 X=1
for c in c5 c3; do bash tools/prof_one.sh $c > /dev/null 2>&1; head -5 gpurun_out/prof_${c}_summary.txt | cut -c1-170; done
} 2>&1 | tee gpurun_out/exp_walkg2.log
