# round 6, call 5: k_walkg with delayed result stores (LDS now, HBM an entry later) -- parity, kernel time against call 4's numbers;
# how v_mfma_f32_32x32x16_bf16 rounds (probes in tools/microbench/bf16x3.hip)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c5; export TMPDIR=/tmp
(cd tools/microbench && timeout 300 ./bf16x3 2>&1 | sed -n '/^1\./,/^2\. /p' > ../../gpurun_out/c5/rounding.txt); cat gpurun_out/c5/rounding.txt
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c5/gputests.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f ms/step %.4f all_kernels %.4f partials %.4f frac %.3f' % (d['value'], d['ms_per_step'], r['all_kernels_ms_per_step'], r['partials_kernel_ms_per_step'], r['frac']))"; }
{
for c in c5 c3; do
  for lib in mrbayes_amd/libhmsbeagle.so build_x/libhmsbeagle_bf40.so build_x/libhmsbeagle_r5.so; do
    for geo in "0 0" "2 3" "2 2"; do
      set -- $geo
      [ "$lib" = build_x/libhmsbeagle_r5.so ] && [ "$1" != 0 ] && continue
      echo "-- $c $lib bins $1 slots $2"
      env MBAMD_LIBRARY=$PWD/$lib $( [ $1 != 0 ] && echo MBAMD_WALK_WAVES=$1 MBAMD_MAX_LDS_SLOTS=$2 ) timeout 300 python bench.py --config $c --steps 200 --no-cpu-baseline --no-also --no-mcmc | line
    done
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c5/ab.txt
