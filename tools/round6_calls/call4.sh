# round 6, call 4: after v_cvt_pk_bf16_f32 went through the compiler (wait states in front of the MFMA): GPU parity again, per-site
# error, and a geometry sweep (subtree bins x LDS slots) now that a second wave of a SIMD no longer pays the first one's MFMA time
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c4; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c4/gputests.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f ms/step %.4f all_kernels %.4f partials %.4f frac %.3f' % (d['value'], d['ms_per_step'], r['all_kernels_ms_per_step'], r['partials_kernel_ms_per_step'], r['frac']))"; }
{
for lib in mrbayes_amd/libhmsbeagle.so build_x/libhmsbeagle_bf40.so; do
  echo "== $lib"
  MBAMD_LIBRARY=$PWD/$lib timeout 300 python tools/site_error.py c3 c5
done
for c in c5 c3; do
  for lib in mrbayes_amd/libhmsbeagle.so build_x/libhmsbeagle_bf40.so build_x/libhmsbeagle_r5.so; do
    for geo in "0 0" "2 2" "2 3" "4 2" "4 3" "4 4" "8 2"; do
      set -- $geo
      [ "$lib" = build_x/libhmsbeagle_r5.so ] && [ "$1" != 0 ] && continue
      [ "$c" = c5 ] && [ "$lib" = build_x/libhmsbeagle_bf40.so ] && continue
      [ "$c" = c5 ] && [ "$1" = 8 ] && continue
      echo "-- $c $lib bins $1 slots $2"
      env MBAMD_LIBRARY=$PWD/$lib ${1:+$( [ $1 != 0 ] && echo MBAMD_WALK_WAVES=$1 MBAMD_MAX_LDS_SLOTS=$2 )} timeout 300 python bench.py --config $c --steps 200 --no-cpu-baseline --no-also --no-mcmc | line
    done
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c4/sweep.txt
