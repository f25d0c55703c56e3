#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log; tail -6 gpurun_out/pytest_gpu.log
for k in m3 wag; do timeout 600 python tools/eigen_time.py $k 2>&1 | tail -5; done | tee gpurun_out/eigen_time.log
timeout 900 python - > gpurun_out/mcmc_quick.json 2>gpurun_out/mcmc_quick.err <<'PY'
import json, bench
print(json.dumps(bench.mcmc_gen_per_s(json.load(open("tests/golden/bench_c2.json")), quick=True)))
PY
python -c "
import json; d=json.load(open('gpurun_out/mcmc_quick.json')); print({k:(v if not isinstance(v,dict) else {a:b for a,b in v.items() if 'ngen' not in a}) for k,v in d.items() if k!='note'})"
tail -3 gpurun_out/mcmc_quick.err
