#!/bin/bash
# round 5, call 23: k_path4 forming the queued matrices itself -- parity, bits against the separate matrix kernel, MCMC rate
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_mrbayes_dropin.py -x -q -m gpu -k "not codon and not protein" > gpurun_out/r5c23_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c23_pytest.log
tail -4 gpurun_out/r5c23_pytest.log
{
for v in "" 1; do
  echo "== MBAMD_NO_PATH_MATRICES=$v"
  env ${v:+MBAMD_NO_PATH_MATRICES=1} timeout 300 python tools/partial_time.py gtr 400
  env ${v:+MBAMD_NO_PATH_MATRICES=1} timeout 300 python tools/mcmc_stats.py 500 20000 12000 dynamic fixed 2>&1 | grep 'wall\|UpdatePartials\|waiting\|Analysis used'
done
} 2>&1 | tee gpurun_out/r5c23_path4.log
