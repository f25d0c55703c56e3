#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
CFG=${1:-c2}
{
cd /tmp
if [[ "${QUICK:-0}" == 1 ]]; then
  PASSES=("SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE")
else
  PASSES=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
            "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
            "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum")
fi
if [[ -n "${PMC_PASSES:-}" ]]; then IFS=';' read -r -a PASSES <<< "$PMC_PASSES"; fi     # experiments: own counter sets, ';'-separated
for pass in "${PASSES[@]}"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-mcmc --no-arith > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  echo "== PMC $pass"
  [[ -n "$f" ]] && python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if any(t in k for t in ("walk", "mfma", "tips", "transition", "integrate")):
        print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, 'dispatches', len(next(iter(v.values()))),
              'sum', {c: round(sum(x),1) for c,x in v.items() if c in ('FETCH_SIZE', 'WRITE_SIZE')})
PY
done
} 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/pmc_walk_$CFG${PMC_TAG:-}.log
