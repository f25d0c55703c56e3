#!/bin/bash
# call 30: the entry fee of an integration inside the walk -- every workgroup of k_walkg ends with a device-scope release and one atomic
# (ablation build MBAMD_WG_ABL_TAIL_FENCE; MBAMD_WG_TAIL_FENCE=1 switches it on): kernel time at protein 200 x 10 000 and codon 100 x 5 000
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c30; export TMPDIR=/tmp
run() { cfg=$1; shift; env "$@" python bench.py --config $cfg --steps 300 --warmup 30 --no-cpu-baseline --no-also --no-mcmc --no-arith 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('   ms/step %.4f  all kernels %.4f  partials %.4f' % (o['ms_per_step'], r.get('all_kernels_ms_per_step', 0), r.get('partials_kernel_ms_per_step', 0)))
"; }
{
for cfg in c3 c5; do for rep in 1 2; do
echo "== $cfg without"; run $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_tailfence.so
echo "== $cfg with the fence and the atomic"; run $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_tailfence.so MBAMD_WG_TAIL_FENCE=1
done; done
} 2>&1 | tee gpurun_out/c30/tail_fence.txt
