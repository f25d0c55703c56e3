# round 6, call 6: where a wave's time goes in k_walkg (clock-stamp build, tools/round6_calls/walkg_stamps.diff; perturbs the kernel)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c6; export TMPDIR=/tmp
rm -f gpurun_out/c6/stamps.txt
for spec in "c5 stamps 2 3" "c5 stamps 2 2" "c3 stamps 0 0" "c3 stamps40 0 0"; do
  set -- $spec
  echo "== $1 lib $2 bins $3 slots $4" >> gpurun_out/c6/stamps.txt
  env MBAMD_WG_STAMPS=$PWD/gpurun_out/c6/stamps.txt MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_$2.so $( [ $3 != 0 ] && echo MBAMD_WALK_WAVES=$3 MBAMD_MAX_LDS_SLOTS=$4 ) timeout 300 python bench.py --config $1 --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-mcmc > /dev/null 2>&1
done
cat gpurun_out/c6/stamps.txt
