#!/bin/bash
# round 5, call 36: parsimony walk over 8 waves + 4-wave score blocks + the linear scheduler: GPU parity, engine statistics, chain rate
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_mrbayes_dropin.py tests/test_fullsize_dropin.py -x -q -m gpu -k "pars" 2>&1 | tail -2
{
timeout 600 python tools/mcmc_pars_stats.py 2>&1 | grep -v '%'
timeout 400 python tools/mcmc_walls.py pars 2000 27000
} 2>&1 | tee gpurun_out/r5c36.log | tail -34
