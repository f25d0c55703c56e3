#!/bin/bash
# Round-6 evidence run on the GPU box: GPU tests, smoke, the default bench line, rocprofv3 kernel statistics + timelines and the
# PMC passes of the four workloads, the fixed-topology chain -> gpurun_out/ (copied to profiles/r06_* by tools/collect_round6.sh)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/smoke.log
# the PMC passes first: the bench line's roofline.traffic is read from profiles/pmc_traffic.json, regenerated here from THIS code's counters
for c in c4 c2 c3 c5; do bash tools/pmc_walk.sh $c > /dev/null 2>&1; grep -c "PMC" gpurun_out/pmc_walk_$c.log; done
python tools/pmc_traffic.py gpurun_out r06 > /dev/null 2>&1; cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
timeout 1500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench default exit $?"
for c in c4 c2 c3 c5; do TIMELINE=16 bash tools/prof_one.sh $c > /dev/null 2>&1; head -6 gpurun_out/prof_${c}_summary.txt | cut -c1-170; done
# a full-tree evaluation under both rescaling schemes, the time of a path + log-likelihood launch
{ for c in bench_c2 bench_c4 bench_c3 bench_c5; do timeout 600 python tools/scale_read_time.py $c 100; done; timeout 600 python tools/path_time.py bench_c2 300; } > gpurun_out/scale_read_and_path.txt 2>&1
# the fixed-topology chain of the unmodified binary: engine statistics of 42 000 generations, kernel timeline of 2 000
{
echo "# Round 6: where a generation of the real binary goes -- fixed topology (branch-length and parameter moves), DNA 500 x 20 000, one chain,"
echo "# unmodified oracle/_ref/mb_amd on one MI355X (tools/gpu_round6.sh).  (1) MBAMD_STATS=1, 12 000 generations; (2) rocprofv3 kernel trace of"
echo "# 2 000 generations (tools/prof_mcmc.sh; times in us); (3) walls of 2 000 / 42 000 generations, each twice (tools/mcmc_walls.py)."
echo; echo "== (1)"
timeout 300 python tools/mcmc_stats.py 500 20000 12000 dynamic fixed 2>&1 | grep 'mbamd\|Analysis used'
echo; echo "== (2)"
timeout 600 bash tools/prof_mcmc.sh 2>&1 | head -44
echo; echo "== (3)"
timeout 300 python tools/mcmc_walls.py 2000 42000
} > gpurun_out/mcmc_fixed_topology.txt 2>&1
tail -3 gpurun_out/mcmc_fixed_topology.txt
