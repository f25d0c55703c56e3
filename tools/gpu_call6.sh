#!/bin/bash
# after the device-header refactor: core GPU parity, kernel times, and scalar-cache counters of the 4-state walk
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q > gpurun_out/pytest_gpu_core.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu_core.log
for c in c4 c2 c5 c3; do
  timeout 300 python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-mcmc > gpurun_out/q_$c.json 2> gpurun_out/q_$c.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/q_$c.json") if l.startswith("{")][-1]); r=d["roofline"]
print("$c", "ms/step %.4f all %.4f partials %.4f frac %.3f" % (d["ms_per_step"], r["all_kernels_ms_per_step"], r["partials_kernel_ms_per_step"], r["frac"]))
PY
done
SQC="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE;SQC_DCACHE_BUSY_CYCLES SQC_DCACHE_INPUT_VALID_READYB SQC_TC_REQ SQC_TC_STALL;SQC_TC_DATA_READ_REQ SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES;SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES;GRBM_GUI_ACTIVE"
PMC_PASSES="$SQC" PMC_TAG=_sqc bash tools/pmc_walk.sh c4 > /dev/null 2>&1; grep -A1 "== PMC" gpurun_out/pmc_walk_c4_sqc.log | grep -v "^--" | cut -c1-400
MBAMD_WALK_WAVES=2 PMC_PASSES="$SQC" PMC_TAG=_sqc_w2 bash tools/pmc_walk.sh c4 > /dev/null 2>&1; grep -A1 "== PMC" gpurun_out/pmc_walk_c4_sqc_w2.log | grep -v "^--" | cut -c1-400
PMC_PASSES="$SQC" PMC_TAG=_sqc bash tools/pmc_walk.sh c5 > /dev/null 2>&1; grep -A3 "== PMC" gpurun_out/pmc_walk_c5_sqc.log | grep -v "^--" | cut -c1-400
