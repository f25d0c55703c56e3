# round 6, call 12: the whole default bench line as it stands + HBM traffic of the codon workload with bf16 tables
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c12; export TMPDIR=/tmp
timeout 1500 python bench.py > gpurun_out/c12/bench_default.json 2> gpurun_out/c12/bench_default.err
tail -c 3000 gpurun_out/c12/bench_default.json
PMC_PASSES="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES" bash tools/pmc_walk.sh c5 > /dev/null 2>&1
cp gpurun_out/pmc_walk_c5.log gpurun_out/c12/
grep -A3 '== PMC' gpurun_out/pmc_walk_c5.log | cut -c1-300
