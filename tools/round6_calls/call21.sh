#!/bin/bash
# call 21: the parsimony scorer's sums as their own completion signal (A/B on the default mix with the device-parsimony binding);
# waves per workgroup of the general-state walk at protein 200 x 10 000 and codon 100 x 5 000 (as call 18 did for four states).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c21; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu -k "pars" 2>&1 | tail -3 | tee gpurun_out/c21/tests.txt
timeout 1200 python tools/mcmc_ab.py pars mix 2000 32000 MBAMD_NO_SUM_POLL=1 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite' | tee gpurun_out/c21/ab_mix.txt
run() { cfg=$1; shift; env "$@" python bench.py --config $cfg --steps 300 --warmup 30 --no-cpu-baseline --no-also --no-mcmc --no-arith 2>gpurun_out/c21/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('   ms/step %.4f  all kernels %.4f  partials %.4f  frac %.3f' % (o['ms_per_step'], r.get('all_kernels_ms_per_step', 0), r.get('partials_kernel_ms_per_step', 0), r['frac']))
"; grep 'tree walk\|walk plan' gpurun_out/c21/err.txt | sort | uniq -c | head -3; }
{
for cfg in c3 c5; do
for w in 1 2 4 8; do echo "== $cfg W=$w"; run $cfg MBAMD_WALK_WAVES=$w MBAMD_VERBOSE=1; done
echo "== $cfg default"; run $cfg MBAMD_VERBOSE=1
done
} 2>&1 | tee gpurun_out/c21/waves.txt
