# round 6, call 7: k_walkb (the software-pipelined 60..63-state walk) against k_walkg on the same tables (MBAMD_NO_WALKB=1) and
# round 5's library; GPU parity of the whole engine suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c7; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c7/gputests.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f ms/step %.4f all_kernels %.4f partials %.4f frac %.3f' % (d['value'], d['ms_per_step'], r['all_kernels_ms_per_step'], r['partials_kernel_ms_per_step'], r['frac']))"; }
{
for c in c5; do
  for spec in "mrbayes_amd/libhmsbeagle.so 0 0 0" "mrbayes_amd/libhmsbeagle.so 0 0 1" "mrbayes_amd/libhmsbeagle.so 2 3 0" "mrbayes_amd/libhmsbeagle.so 2 2 0" "mrbayes_amd/libhmsbeagle.so 4 3 0" "mrbayes_amd/libhmsbeagle.so 4 2 0" "build_x/libhmsbeagle_r5.so 0 0 0"; do
    set -- $spec
    echo "-- $c $1 bins $2 slots $3 no_walkb $4"
    env MBAMD_LIBRARY=$PWD/$1 $( [ $2 != 0 ] && echo MBAMD_WALK_WAVES=$2 MBAMD_MAX_LDS_SLOTS=$3 ) $( [ $4 != 0 ] && echo MBAMD_NO_WALKB=1 ) timeout 300 python bench.py --config $c --steps 200 --no-cpu-baseline --no-also --no-mcmc | line
  done
done
MBAMD_LIBRARY=$PWD/mrbayes_amd/libhmsbeagle.so timeout 300 python tools/site_error.py c5
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c7/ab.txt
