#!/bin/bash
# gpurun_out/ -> profiles/r06_* (run in the build container after tools/gpu_round6.sh came back)
cd "$(dirname "$0")/.."
cp gpurun_out/bench_default.json profiles/r06_bench_default.json
for c in c2 c3 c4 c5; do
  cp gpurun_out/prof_${c}_summary.txt profiles/r06_${c}_kernel_stats.txt
  cp gpurun_out/prof_${c}_timeline.txt profiles/r06_${c}_timeline.txt
  cp gpurun_out/pmc_walk_$c.log profiles/r06_${c}_pmc.txt
done
cp gpurun_out/mcmc_fixed_topology.txt profiles/r06_mcmc_fixed_topology.txt
python tools/pmc_traffic.py gpurun_out r06 > /dev/null
cp gpurun_out/scale_read_and_path.txt profiles/r06_scale_read_and_path_final.txt
