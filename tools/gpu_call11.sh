#!/bin/bash
# fp64 four-state walk: device time per evaluation (rocprofv3 kernel statistics) against the level kernels, slot budgets
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for sl in default; do
  if [[ $sl == default ]]; then unset MBAMD_F64_WALK_SLOTS; else export MBAMD_F64_WALK_SLOTS=$sl; fi
  bash tools/prof_tool.sh f64_$sl $PWD/tools/f64_time.py c4 > /dev/null 2>&1
  echo "== slots $sl"; grep -E "k64_walk4|k64_partials_fused|k64_integrate|k64_matrices" gpurun_out/prof_f64_${sl}_summary.txt | cut -c1-175
done
unset MBAMD_F64_WALK_SLOTS
MBAMD_F64_WALK_ALWAYS=1 bash tools/prof_tool.sh f64_c2 $PWD/tools/f64_time.py c2 > /dev/null 2>&1
echo "== c2"; grep -E "k64_walk4|k64_partials_fused" gpurun_out/prof_f64_c2_summary.txt | cut -c1-175
timeout 600 python tools/f64_time.py c2 c4 2>&1 | tee gpurun_out/f64_time.txt
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "double" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_mrbayes_dropin.py -m gpu -x -q -k "double" 2>&1 | tail -2
