# round 6, call 15: integration kernels announce their own completion (mbd_signal_done) -- full GPU suite, then whole-MCMC rates and
# the four workloads' step times with and without (MBAMD_STREAM_FLAG=1: the stream writes the flag behind the kernel, as in round 5)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c15; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c15/gputests.txt
for v in "" 1; do
  echo "== MBAMD_STREAM_FLAG=$v"
  env ${v:+MBAMD_STREAM_FLAG=1} timeout 900 python bench.py --config c2 --steps 200 --no-cpu-baseline --no-mpi --no-arith 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['mcmc_gen_per_s']
print({k: {a: round(b,1) for a,b in v.items() if isinstance(b,float)} for k,v in m.items() if isinstance(v,dict)})
print({k: v for k, v in d['summary'].items() if k in ('c2','c3','c4','c5')})"
done 2>&1 | tee gpurun_out/c15/ab.txt
