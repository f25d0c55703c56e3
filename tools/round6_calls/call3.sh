# round 6, call 3: the general-state walk on the 16-bit matrix cores (bf16 x 3 pieces): GPU parity, per-site error and kernel time
# against round 5's library (build_x/libhmsbeagle_r5.so, fp32 MFMA) on the same box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c3; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c3/gputests.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f ms/step %.4f all_kernels %.4f partials %.4f frac %.3f lnL %r' % (d['value'], d['ms_per_step'], r['all_kernels_ms_per_step'], r['partials_kernel_ms_per_step'], r['frac'], d.get('lnL')))"; }
for lib in build_x/libhmsbeagle_r5.so mrbayes_amd/libhmsbeagle.so; do
  echo "== $lib"
  MBAMD_LIBRARY=$PWD/$lib timeout 300 python tools/site_error.py c3 c5
  for c in c5 c3; do
    for rep in 1 2; do
      echo "-- $c"; MBAMD_LIBRARY=$PWD/$lib timeout 300 python bench.py --config $c --steps 200 --no-cpu-baseline --no-also --no-mcmc | line
    done
  done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c3/ab.txt
