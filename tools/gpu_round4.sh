#!/bin/bash
# Round-4 evidence run on the GPU box: GPU tests, smoke, the default bench line, rocprofv3 kernel statistics + timelines and the
# PMC passes of the four workloads -> gpurun_out/ (copied to profiles/r04_* by tools/collect_round4.sh afterwards)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/smoke.log
timeout 1500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench default exit $?"
for c in c4 c2 c3 c5; do TIMELINE=16 bash tools/prof_one.sh $c > /dev/null 2>&1; head -6 gpurun_out/prof_${c}_summary.txt | cut -c1-170; done
for c in c4 c2 c3 c5; do bash tools/pmc_walk.sh $c > /dev/null 2>&1; grep -c "PMC" gpurun_out/pmc_walk_$c.log; done
# the double-precision engine: kernel statistics of replayed evaluations (DNA 1000 x 50 000, codon M3 100 x 5 000)
for c in c4 c5; do
  rm -rf /tmp/pf64; (cd /tmp && F64_STEPS=10 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf64 -o x -- python $GRAFT_REPO_ROOT/tools/f64_bench.py $c > /tmp/pf64.log 2>&1)
  db=$(find /tmp/pf64 -name "*.db" | head -1); { echo "== tools/f64_bench.py $c (10 + 8 evaluations)"; tail -1 /tmp/pf64.log; python tools/rocpd_summary.py $db | grep -i "k64\|Name" | cut -c1-200; } >> gpurun_out/prof_f64_summary.txt
done
