#!/bin/bash
# GPU-box script for the device-parsimony row: parity tests, pass timings, whole-MCMC default mix with and without it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "parsimony" > $OUT/pytest_pars.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_pars.log
(timeout 300 python tools/pars_time.py 500 20000 4; timeout 300 python tests/pars_cpu_time.py 500 20000 4) > $OUT/pars_time_c2.txt 2>&1; cat $OUT/pars_time_c2.txt
(timeout 300 python tools/pars_time.py 1000 50000 4; timeout 300 python tests/pars_cpu_time.py 1000 50000 4) > $OUT/pars_time_c4.txt 2>&1; cat $OUT/pars_time_c4.txt
(timeout 300 python tools/pars_time.py 200 10000 20; timeout 300 python tests/pars_cpu_time.py 200 10000 20) > $OUT/pars_time_c3.txt 2>&1; cat $OUT/pars_time_c3.txt
timeout 1200 python - > $OUT/mcmc_pars.json 2> $OUT/mcmc_pars.err <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
with open(os.path.join(bench.GOLD, bench.CONFIGS["c2"][0] + ".json")) as fh:
    gold = json.load(fh)
print(json.dumps(bench.mcmc_gen_per_s(gold, quick=True)))
PY
cat $OUT/mcmc_pars.json; tail -3 $OUT/mcmc_pars.err
