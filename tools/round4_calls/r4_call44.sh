#!/bin/bash
# Round 4, GPU call 44: counters of the fp64 codon level kernels as they are now (wide levels: what holds the matrix cores at a third?)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call44.log; : > $OUT
cd /tmp
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  rm -rf /tmp/pm; F64_STEPS=3 timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/f64_bench.py c5 > /tmp/pm.log 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  echo "== PMC $pass" | tee -a $OUT
  if [[ -n "$f" ]]; then python - "$f" <<'PY' | tee -a $OUT
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "k64_partials" not in r['Kernel_Name']: continue
    key=(r['Kernel_Name'].split('(')[0][-32:], r.get('Grid_Size','?'))
    acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items(), key=lambda kv: -int(kv[0][1]) if kv[0][1].isdigit() else 0)[:4]:
    print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, 'dispatches', len(next(iter(v.values()))))
PY
  else tail -3 /tmp/pm.log | tee -a $OUT; fi
done
