#!/bin/bash
# round 5, call 37: the 4-state matrix kernel launched at beagleUpdateTransitionMatrices (one call per evaluation) instead of at the
# list that follows -- GPU parity of the MrBayes drop-in tests, fixed-topology and default-mix chains with / without
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_mrbayes_dropin.py -x -q -m gpu -k "not codon and not protein" 2>&1 | tail -2
{
timeout 300 python tools/mcmc_walls.py 2000 42000
timeout 300 python tools/mcmc_walls.py 2000 42000 MBAMD_NO_EAGER_MATRICES=1
timeout 300 python tools/mcmc_walls.py pars 2000 27000
timeout 300 python tools/mcmc_walls.py pars 2000 27000 MBAMD_NO_EAGER_MATRICES=1
} 2>&1 | tee gpurun_out/r5c37.log
