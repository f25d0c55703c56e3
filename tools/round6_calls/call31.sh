#!/bin/bash
# call 31: did the exponent ring / the conditional exponent store cost the SCALE_WRITE walk anything?  The library of commit da17019 (three
# waves per workgroup, sums polled, neither of the two) against the product on one box, DNA 1000 x 50 000 and 500 x 20 000, twice each
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c31; export TMPDIR=/tmp
run() { cfg=$1; shift; env "$@" python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-mcmc --no-arith 2>gpurun_out/c31/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('   ms/step %.4f  all kernels %.4f  partials %.4f' % (o['ms_per_step'], r.get('all_kernels_ms_per_step', 0), r.get('partials_kernel_ms_per_step', 0)))
"; grep 'walk plan' gpurun_out/c31/err.txt | sort | uniq -c | head -2; }
{
for rep in 1 2; do for cfg in c4 c2; do
echo "== $cfg product"; run $cfg MBAMD_VERBOSE=1
echo "== $cfg da17019"; run $cfg MBAMD_VERBOSE=1 MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_prering.so
done; done
} 2>&1 | tee gpurun_out/c31/ab.txt
