#!/bin/bash
# call 27: where k_path4_lnl's time goes -- ablation builds (timing only): no sibling loads, no result stores, neither
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c27; export TMPDIR=/tmp
{
echo "== product"; timeout 600 python tools/path_time.py bench_c2 300
for v in p4noload p4nostore p4neither; do echo "== $v"; MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_$v.so timeout 600 python tools/path_time.py bench_c2 300; done
echo "== product, k_path4 + k_integrate_lnl_s4 (MBAMD_NO_FUSE_PATH=1)"; MBAMD_NO_FUSE_PATH=1 timeout 600 python tools/path_time.py bench_c2 300
} 2>&1 | tee gpurun_out/c27/path_ablation.txt
