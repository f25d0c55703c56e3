#!/bin/bash
# Run on the GPU box (through gpurun): GPU parity tests, smoke, bench lines, rocprof summaries, PMC passes -> gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
WHAT="${1:-all}"
if [[ "$WHAT" == all || "$WHAT" == tests ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log
fi
if [[ "$WHAT" == all || "$WHAT" == bench ]]; then
  timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default exit $?"
  for c in c3 c5; do
    timeout 900 python bench.py --config $c --no-also --no-mcmc > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "bench $c exit $?"
  done
fi
if [[ "$WHAT" == all || "$WHAT" == prof ]]; then
  for c in c2 c4 c3 c5; do TIMELINE=16 bash tools/prof_one.sh $c > /dev/null 2>&1; head -8 $OUT/prof_${c}_summary.txt | cut -c1-160; done
fi
if [[ "$WHAT" == all || "$WHAT" == pmc ]]; then
  for c in c2 c4 c3 c5; do bash tools/pmc_walk.sh $c > /dev/null 2>&1; grep -c "PMC" $OUT/pmc_walk_$c.log; done
fi
