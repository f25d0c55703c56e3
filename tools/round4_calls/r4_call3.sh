#!/bin/bash
# Round 4, GPU call 3: k_walkg_s with the rotated LDS layout (conflict-free tip gathers) and operand reads first; the leaner
# k_walk4_t against round 3's library on the same box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call3.log; : > $OUT
say() { echo "$@" | tee -a $OUT; }
run() {  # label, config, env...
  local label=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-also --no-mcmc 2>/tmp/exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-28s %s  value %9.0f  ms/step %.4f  partials %.4f  all %.4f  frac %.3f' % ('$label', '$cfg', d['value'], d['ms_per_step'], r['partials_kernel_ms_per_step'], r['all_kernels_ms_per_step'], r['frac']))
" | tee -a $OUT
  [ ${PIPESTATUS[0]} -ne 0 ] && { say "$label $cfg FAILED"; tail -5 /tmp/exp.err | tee -a $OUT; }
}
timeout 1500 python -m pytest tests/test_engine_gpu.py -q -m gpu --maxfail=10 > gpurun_out/r4_pytest_gpu3.log 2>&1; say "engine gpu tests exit $?"; tail -6 gpurun_out/r4_pytest_gpu3.log | tee -a $OUT
for m in m3 wag; do
  say "== trace $m"
  MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_wgs_trace.so timeout 300 python tools/trace_walkgs.py $m 2>&1 | tail -14 | tee -a $OUT
done
say "== k_walkg_s vs k_walkg (kernel ms per evaluation)"
for cfg in c5 c3; do
  timeout 300 python tools/ablate_walkg.py $cfg 2>&1 | tail -1 | tee -a $OUT
  MBAMD_WALKG_SHARED=0 timeout 300 python tools/ablate_walkg.py $cfg 2>&1 | tail -1 | sed 's/product/old kernel/' | tee -a $OUT
done
say "== 4-state walk: round 4 loop vs round 3 library"
for rep in 1 2; do for cfg in c4 c2; do
  run r4 $cfg X=1
  run r3 $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_r3.so
done; done
