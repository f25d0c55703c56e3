#!/bin/bash
# Run on the GPU box (through gpurun): GPU parity tests, smoke, bench, rocprof summaries -> gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
WHAT="${1:-all}"
if [[ "$WHAT" == all || "$WHAT" == tests ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
  tail -15 $OUT/pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log
fi
if [[ "$WHAT" == all || "$WHAT" == bench ]]; then
  for c in c2 c3 c5 c4; do
    extra="--no-cpu-baseline --no-also --no-mcmc --no-also --no-mcmc"; [[ $c == c2 ]] && extra=""
    timeout 900 python bench.py --config $c --steps 50 --warmup 5 $extra > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "bench $c exit $?"
    cat $OUT/bench_$c.json
  done
fi
if [[ "$WHAT" == all || "$WHAT" == prof ]]; then
  for c in c2 c3 c5; do bash tools/prof_one.sh $c; done
fi
