#!/bin/bash
# round 5, call 24: what bounds k_transition_matrices_mfma at C5 (32 us for 712 matrices)? ablations: no exp / no operand loads / no stores
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in product noexp noload nostore notab all3; do
  lib=""; [ $v != product ] && lib=$GRAFT_REPO_ROOT/build_x/libhmsbeagle_$v.so
  rm -rf /tmp/pv; (cd /tmp && env ${lib:+MBAMD_LIBRARY=$lib} timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pv -o v -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 30 --warmup 3 --no-cpu-baseline --no-also --no-mcmc > /tmp/pv.log 2>&1)
  db=$(find /tmp/pv -name "*.db" | head -1)
  echo "== $v"; python tools/rocpd_summary.py $db | grep 'k_transition\|k_integrate\|k_walkg' | cut -c1-170
done 2>&1 | tee gpurun_out/r5c24.log
