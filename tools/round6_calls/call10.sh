# round 6, call 10: k_walkg on bf16 tables with windowed operand loads (mbd_buf) against the same kernel with 64-bit addresses
# (build_x/libhmsbeagle_wb3.so, MBAMD_NO_WALKB=1) -- parity and kernel time
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c10; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c10/gputests.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f ms/step %.4f all_kernels %.4f partials %.4f frac %.3f' % (d['value'], d['ms_per_step'], r['all_kernels_ms_per_step'], r['partials_kernel_ms_per_step'], r['frac']))"; }
{
for rep in 1 2; do
for spec in "mrbayes_amd/libhmsbeagle.so 0" "build_x/libhmsbeagle_wb3.so 1" "build_x/libhmsbeagle_r5.so 0"; do
    set -- $spec
    echo "-- c5 $1 no_walkb $2"
    env MBAMD_LIBRARY=$PWD/$1 $( [ $2 != 0 ] && echo MBAMD_NO_WALKB=1 ) timeout 300 python bench.py --config c5 --steps 200 --no-cpu-baseline --no-also --no-mcmc | line
done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c10/ab.txt
