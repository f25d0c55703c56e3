#!/bin/bash
# Round 4, GPU call 28: the integration kernel itself announces its result (last workgroup writes the polled word) -- GPU tests, MCMC statistics with and without (MBAMD_NO_POLL=1)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call28.log; : > $OUT
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee -a $OUT
timeout 1200 python -m pytest tests/test_mrbayes_dropin.py tests/test_mpi_shim.py -x -q -m gpu 2>&1 | tail -2 | tee -a $OUT
run() { echo "== $*" | tee -a $OUT; env "$@" timeout 900 python tools/mcmc_stats.py 500 20000 12000 dynamic fixed 2>&1 | grep -i "^wall\|waiting for\|Analysis completed\|Analysis used" | tee -a $OUT; }
run X=1
run MBAMD_NO_POLL=1
timeout 600 python bench.py --no-also --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); m = d['mcmc_gen_per_s']
print('c4', d['value'], d['ms_per_step'], d['roofline']['frac'])
print({k: {a: round(b, 1) for a, b in v.items() if isinstance(b, float)} for k, v in m.items() if isinstance(v, dict)})" | tee -a $OUT
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_engine_gpu.py --deselect tests/test_mrbayes_dropin.py --deselect tests/test_mpi_shim.py --maxfail=5 2>&1 | tail -3 | tee -a $OUT
