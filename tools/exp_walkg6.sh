#!/bin/bash
# k_walkg: workgroup -> (tile, category, list) mapping with the category slowest (co-resident workgroups share tables in L1)
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
L=$PWD/build_x/libhmsbeagle_remap.so
for cfg in c5 c3; do for rep in 1 2; do run base $cfg X=1; run remap $cfg MBAMD_LIBRARY=$L; done; done
echo "== large: codon 100 x 40000, protein 200 x 160000"
timeout 400 python tools/scale_time.py m3 100 40000; MBAMD_LIBRARY=$L timeout 400 python tools/scale_time.py m3 100 40000
timeout 400 python tools/scale_time.py wag 200 160000; MBAMD_LIBRARY=$L timeout 400 python tools/scale_time.py wag 200 160000
} 2>&1 | tee gpurun_out/exp_walkg6.log
