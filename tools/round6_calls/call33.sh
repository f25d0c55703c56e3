#!/bin/bash
# call 33 (as call 32, after the exponent ring was taken out again): the product against the library of commit da17019 on one box
# library of commit da17019 on one box (SCALE_WRITE walk, DNA 1000 x 50 000 and 500 x 20 000), a full-tree evaluation under both schemes,
# the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c33; export TMPDIR=/tmp
run() { cfg=$1; shift; env "$@" python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-mcmc --no-arith 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('   ms/step %.4f  all kernels %.4f  partials %.4f' % (o['ms_per_step'], r.get('all_kernels_ms_per_step', 0), r.get('partials_kernel_ms_per_step', 0)))
"; }
{
for rep in 1 2; do for cfg in c4 c2; do
echo "== $cfg product"; run $cfg
echo "== $cfg da17019"; run $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_prering.so
done; done
timeout 600 python tools/scale_read_time.py bench_c2 200; timeout 900 python tools/scale_read_time.py bench_c4 60
} 2>&1 | tee gpurun_out/c33/ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c33/gputests.txt
