#!/bin/bash
# Round 4, GPU call 1: k_walkg_s (transition tables staged in LDS, shared by the waves of a workgroup) -- parity first, then
# C3 / C5 timings of the variants (waves per workgroup, bins, prefetch distance) against k_walkg.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call1.log; : > $OUT
say() { echo "$@" | tee -a $OUT; }
run() {  # label, config, env...
  local label=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-also --no-mcmc 2>/tmp/exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-28s %s  value %9.0f  ms/step %.4f  partials %.4f  all %.4f  frac %.3f' % ('$label', '$cfg', d['value'], d['ms_per_step'], r['partials_kernel_ms_per_step'], r['all_kernels_ms_per_step'], r['frac']))
" | tee -a $OUT
  [ ${PIPESTATUS[0]} -ne 0 ] && { say "$label $cfg FAILED"; tail -5 /tmp/exp.err | tee -a $OUT; }
}
say "== parity: the GPU suite on the general-state paths (safe waits first, then the product)"
MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_wgs_safe.so timeout 900 python -m pytest tests/test_engine_gpu.py -q -m gpu -x -k "general or protein or codon or bench_work or states or golden" > gpurun_out/r4_pytest_safe.log 2>&1; say "safe-waits subset exit $?"; tail -3 gpurun_out/r4_pytest_safe.log | tee -a $OUT
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 > gpurun_out/r4_pytest_gpu.log 2>&1; say "full gpu suite exit $?"; tail -15 gpurun_out/r4_pytest_gpu.log | tee -a $OUT
say "== timings"
for cfg in c5 c3; do
  run old_kernel $cfg MBAMD_WALKG_SHARED=0
  run shared_default $cfg X=1
  run shared_G2 $cfg MBAMD_WALKG_G=2
  run shared_G4 $cfg MBAMD_WALKG_G=4
  run shared_d1 $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_wgs_d1.so
  for w in 1 2 3; do run shared_bins$w $cfg MBAMD_WALK_WAVES=$w; done
  run shared_G2_bins3 $cfg MBAMD_WALKG_G=2 MBAMD_WALK_WAVES=3
  run shared_default_again $cfg X=1
done
say "== partial updates (one branch), wall"
for m in wag m3; do
  timeout 300 python tools/partial_time.py $m 2>&1 | tail -2 | tee -a $OUT
  MBAMD_WALKG_SHARED=0 timeout 300 python tools/partial_time.py $m 2>&1 | tail -2 | tee -a $OUT
done
say "== kernel stats, c5 and c3 (rocprofv3)"
for cfg in c5 c3; do TIMELINE=12 bash tools/prof_one.sh $cfg > /dev/null 2>&1; head -8 gpurun_out/prof_${cfg}_summary.txt | cut -c1-170 | tee -a $OUT; done
