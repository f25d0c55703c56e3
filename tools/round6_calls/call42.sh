#!/bin/bash
# call 42: schedule-shape sweeps on the final code (environment switches only): the size below which a phase runs on one wave, LDS slots
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c42; export TMPDIR=/tmp
run() { cfg=$1; shift; env "$@" python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-mcmc --no-arith 2>gpurun_out/c42/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('   ms/step %.4f  all kernels %.4f  partials %.4f' % (o['ms_per_step'], r.get('all_kernels_ms_per_step', 0), r.get('partials_kernel_ms_per_step', 0)))
"; grep 'walk plan' gpurun_out/c42/err.txt | sort | uniq -c | head -1 | cut -c1-160; }
{
for cfg in c2 c3 c5; do
echo "== $cfg default"; run $cfg MBAMD_VERBOSE=1
for sp in 4 8 32 64; do echo "== $cfg MBAMD_WALK_SMALL_PHASE=$sp"; run $cfg MBAMD_VERBOSE=1 MBAMD_WALK_SMALL_PHASE=$sp; done
done
for s in 5 6 7 10; do echo "== c2 MBAMD_MAX_LDS_SLOTS=$s"; run c2 MBAMD_VERBOSE=1 MBAMD_MAX_LDS_SLOTS=$s; done
for s in 4 8; do echo "== c3 MBAMD_MAX_LDS_SLOTS=$s"; run c3 MBAMD_VERBOSE=1 MBAMD_MAX_LDS_SLOTS=$s; done
echo "== c2 default again"; run c2 MBAMD_VERBOSE=1
} 2>&1 | tee gpurun_out/c42/sweeps.txt
