#!/bin/bash
# call 36: the remainder layer in ONE launch (two programs; workgroups carry max(W) waves, unused ones leave at once) -- GPU tests,
# DNA 1000 x 50 000 with / without (MBAMD_NO_TAIL_BLOCKS=1) and with 2 / 4 / 8 waves, the f64 walk
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c36; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "remainder or double or f64 or precision or counted or hazard" 2>&1 | tail -4 | tee gpurun_out/c36/tests.txt
run() { cfg=$1; shift; env "$@" python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-mcmc --no-arith 2>gpurun_out/c36/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('   ms/step %.4f  all kernels %.4f  partials %.4f  frac %.3f  value %.0f' % (o['ms_per_step'], r.get('all_kernels_ms_per_step', 0), r.get('partials_kernel_ms_per_step', 0), r['frac'], o['value']))
"; grep 'remainder\|pattern blocks' gpurun_out/c36/err.txt | sort | uniq -c | head -3; }
{
for rep in 1 2; do
echo "== c4 product"; run c4 MBAMD_VERBOSE=1
echo "== c4 MBAMD_NO_TAIL_BLOCKS=1"; run c4 MBAMD_NO_TAIL_BLOCKS=1
done
for w in 3 4; do echo "== c4 MBAMD_TAIL_WAVES=$w"; run c4 MBAMD_TAIL_WAVES=$w MBAMD_VERBOSE=1; done
} 2>&1 | tee gpurun_out/c36/ab.txt
