#!/bin/bash
# call 24: why SCALE_READ evaluations are slower at DNA 500 x 20 000 (0.181 against 0.125 ms) -- ablation builds (timing only, the
# results of these builds are wrong): no exponent store to the scratch buffer, no vmcnt waits, no exponent DMA
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c24; export TMPDIR=/tmp
{
echo "== product"; timeout 600 python tools/scale_read_time.py bench_c2 200
for v in nostore nowait nostorenowait nodma; do echo "== $v"; MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_$v.so timeout 600 python tools/scale_read_time.py bench_c2 200; done
} 2>&1 | tee gpurun_out/c24/ablation.txt
