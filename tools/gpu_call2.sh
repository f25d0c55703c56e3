#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
run() {  # label, config, env...
  local label=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-mcmc 2>/tmp/exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-28s %s  value %9.0f  ms/step %.4f  partials %.4f  all %.4f  frac %.3f' % ('$label', '$cfg', d['value'], d['ms_per_step'], r['partials_kernel_ms_per_step'], r['all_kernels_ms_per_step'], r['frac']))
" || { echo "$label $cfg FAILED"; tail -3 /tmp/exp.err; }
}
{
for cfg in c5 c3; do
  run base $cfg X=1
  for w in 1 2 4 8; do run waves$w $cfg MBAMD_WALK_WAVES=$w; done
done
} 2>&1 | tee gpurun_out/exp_walkg.log
timeout 600 python - > gpurun_out/mcmc_quick.json 2>gpurun_out/mcmc_quick.err <<'PY'
import json, bench
print(json.dumps(bench.mcmc_gen_per_s(json.load(open("tests/golden/bench_c2.json")), quick=True)))
PY
cut -c1-1500 gpurun_out/mcmc_quick.json
