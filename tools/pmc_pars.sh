#!/bin/bash
# PMC passes of the parsimony kernels (k_pars_walk, k_pars_score) under tools/pars_time.py 500 20000 4 -> gpurun_out/pmc_pars.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
cd /tmp
PASSES=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
        "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
        "FETCH_SIZE" "WRITE_SIZE")
for pass in "${PASSES[@]}"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmcp_$tag
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmcp_$tag -o p -- python $GRAFT_REPO_ROOT/tools/pars_time.py 500 20000 4 > /tmp/pmcp_$tag.log 2>&1
  f=$(find /tmp/pmcp_$tag -name "*counter_collection.csv" | head -1)
  echo "== PMC $pass"
  [[ -n "$f" ]] && python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if "pars" in k:
        print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, 'dispatches', len(next(iter(v.values()))))
PY
done
} 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/pmc_pars.log
