# round 6, call 13: the root-ward path and its log-likelihood as one launch (k_path4_lnl) -- GPU parity (engine + drop-in suites),
# whole-MCMC generations/s with and without (MBAMD_NO_FUSE_PATH=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c13; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_mrbayes_dropin.py tests/test_survey_anchors.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c13/gputests.txt
for v in "" 1; do
  echo "== MBAMD_NO_FUSE_PATH=$v"
  env ${v:+MBAMD_NO_FUSE_PATH=1} timeout 900 python bench.py --config c2 --steps 100 --no-cpu-baseline --no-also --no-mpi --no-arith 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['mcmc_gen_per_s']
print({k: {a: round(b,1) for a,b in v.items() if isinstance(b,float)} for k,v in m.items() if isinstance(v,dict)})"
done 2>&1 | tee gpurun_out/c13/mcmc.txt
