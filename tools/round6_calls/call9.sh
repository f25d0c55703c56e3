# round 6, call 9: k_walkb v3 (K-block chunks, windowed operand loads, four / two rotating operand sets) -- parity and kernel time
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c9; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c9/gputests.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f ms/step %.4f all_kernels %.4f partials %.4f frac %.3f' % (d['value'], d['ms_per_step'], r['all_kernels_ms_per_step'], r['partials_kernel_ms_per_step'], r['frac']))"; }
{
for spec in "build_x/libhmsbeagle_wb3.so 0 0 0" "build_x/libhmsbeagle_wb1.so 0 0 0" "build_x/libhmsbeagle_wb3.so 0 0 1" "build_x/libhmsbeagle_wb3.so 2 3 0" "build_x/libhmsbeagle_wb1.so 2 3 0" "build_x/libhmsbeagle_wb1.so 4 3 0" "build_x/libhmsbeagle_wb1.so 4 2 0" "build_x/libhmsbeagle_r5.so 0 0 0"; do
    set -- $spec
    echo "-- c5 $1 bins $2 slots $3 no_walkb $4"
    env MBAMD_LIBRARY=$PWD/$1 $( [ $2 != 0 ] && echo MBAMD_WALK_WAVES=$2 MBAMD_MAX_LDS_SLOTS=$3 ) $( [ $4 != 0 ] && echo MBAMD_NO_WALKB=1 ) timeout 300 python bench.py --config c5 --steps 200 --no-cpu-baseline --no-also --no-mcmc | line
done
MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_wb3.so timeout 300 python tools/site_error.py c5
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c9/ab.txt
