#!/bin/bash
# call 18: k_walk4_t at DNA 500 x 20 000 -- waves per workgroup (subtree bins) against the load of the fullest SIMD.
# 1 252 workgroups on 256 CUs = 4.9 per CU; W = 2 puts 10 waves of 251 entries on the four SIMDs of most CUs (3 + 3 + 2 + 2).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c18; export TMPDIR=/tmp
run() { env "$@" python bench.py --config c2 --steps 300 --warmup 30 --no-cpu-baseline --no-also --no-mcmc --no-arith 2>gpurun_out/c18/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('   ms/step %.4f  all kernels %.4f  partials %.4f  frac %.3f' % (o['ms_per_step'], r.get('all_kernels_ms', 0), r.get('partials_kernel_ms', 0), r['frac']))
"; grep 'tree walk:\|walk plan' gpurun_out/c18/err.txt | sort | uniq -c | head -4; }
{
for rep in 1 2; do
for w in 2 3 4 5 6 8; do echo "== W=$w"; run MBAMD_WALK_WAVES=$w MBAMD_VERBOSE=1; done
done
echo "== W=4, slots 8 / 12 / 16"
for s in 8 12 16; do echo "slots $s"; run MBAMD_WALK_WAVES=4 MBAMD_MAX_LDS_SLOTS=$s; done
echo "== W=3, slots 8 / 12 / 16"
for s in 8 12 16; do echo "slots $s"; run MBAMD_WALK_WAVES=3 MBAMD_MAX_LDS_SLOTS=$s; done
} > gpurun_out/c18/waves.txt 2>&1
cat gpurun_out/c18/waves.txt
