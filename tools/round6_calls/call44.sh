#!/bin/bash
# call 44: the matrix kernel at three workgroups per CU -- full GPU suite, protein and codon steps against the previous commit
# codon 100 x 5 000 (the product against the previous commit's library), HBM write counters of both workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c44; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/c44/gputests.txt
run() { cfg=$1; shift; env "$@" python bench.py --config $cfg --steps 300 --warmup 30 --no-cpu-baseline --no-also --no-mcmc --no-arith 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('   ms/step %.4f  all kernels %.4f  partials %.4f' % (o['ms_per_step'], r.get('all_kernels_ms_per_step', 0), r.get('partials_kernel_ms_per_step', 0)))
"; }
{
for rep in 1 2; do for cfg in c3 c5; do
echo "== $cfg product"; run $cfg
echo "== $cfg previous commit"; run $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_prev.so
done; done
} 2>&1 | tee gpurun_out/c44/ab.txt
