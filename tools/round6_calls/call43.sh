#!/bin/bash
# call 43: no-operation entries of k_walkg write one line per store instead of a KiB -- general-state GPU tests, protein 200 x 10 000 and
# codon 100 x 5 000 (the product against the previous commit's library), HBM write counters of both workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c43; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/c43/gputests.txt
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
} 2>&1 | tee gpurun_out/c43/ab.txt
for c in c3 c5; do PMC_PASSES="FETCH_SIZE;WRITE_SIZE" bash tools/pmc_walk.sh $c > /dev/null 2>&1; grep -A4 'WRITE_SIZE\|FETCH_SIZE' gpurun_out/pmc_walk_$c.log | grep 'walkg' | cut -c1-160; done | tee gpurun_out/c43/traffic.txt
