#!/bin/bash
# Round 4, GPU call 47: final check of the committed build -- all GPU tests, smoke, the fp64 kernel statistics
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/smoke.log
rm -f gpurun_out/prof_f64_summary.txt
for c in c4 c5; do
  rm -rf /tmp/pf64; (cd /tmp && F64_STEPS=10 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf64 -o x -- python $GRAFT_REPO_ROOT/tools/f64_bench.py $c > /tmp/pf64.log 2>&1)
  db=$(find /tmp/pf64 -name "*.db" | head -1); { echo "== tools/f64_bench.py $c (10 + 8 evaluations)"; tail -1 /tmp/pf64.log; python tools/rocpd_summary.py $db | grep -i "k64\|Name" | cut -c1-200; } >> gpurun_out/prof_f64_summary.txt
done
cat gpurun_out/prof_f64_summary.txt
timeout 600 python tools/f64_bench.py c5 c3 c4 c2 2>&1 | grep config
