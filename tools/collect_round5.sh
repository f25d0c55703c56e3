#!/bin/bash
# gpurun_out/ -> profiles/r05_* (run in the build container after tools/gpu_round5.sh came back)
cd "$(dirname "$0")/.."
cp gpurun_out/bench_default.json profiles/r05_bench_default.json
for c in c2 c3 c4 c5; do
  cp gpurun_out/prof_${c}_summary.txt profiles/r05_${c}_kernel_stats.txt
  cp gpurun_out/prof_${c}_timeline.txt profiles/r05_${c}_timeline.txt
  cp gpurun_out/pmc_walk_$c.log profiles/r05_${c}_pmc.txt
done
cp gpurun_out/prof_f64_summary.txt profiles/r05_f64_kernel_stats.txt
cp gpurun_out/mcmc_fixed_topology.txt profiles/r05_mcmc_fixed_topology.txt
python tools/pmc_traffic.py gpurun_out r05 > /dev/null
