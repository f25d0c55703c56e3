#!/bin/bash
# round 5, call 34: the parsimony walk with the tree cut over the waves of a workgroup -- GPU parity, pass times for 1 / 2 / 4 / 8 waves,
# the default-mix chain with the device-parsimony binding
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "parsimony" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_mrbayes_dropin.py tests/test_fullsize_dropin.py -x -q -m gpu -k "pars" 2>&1 | tail -3
{
for wv in 1 2 4 8; do echo "== MBAMD_PARS_WAVES=$wv"; MBAMD_PARS_WAVES=$wv timeout 200 python tools/pars_time.py 500 20000 4 | head -6; done
echo "== default"; timeout 200 python tools/pars_time.py 500 20000 4 | head -6
timeout 200 python tools/pars_time.py 1000 50000 4 | head -6
timeout 400 python tools/mcmc_walls.py pars 2000 27000
timeout 400 python tools/mcmc_walls.py pars 2000 27000 MBAMD_PARS_WAVES=1
} 2>&1 | tee gpurun_out/r5c34.log
