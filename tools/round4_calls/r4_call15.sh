#!/bin/bash
# Round 4, GPU call 15: fp64 -- tip rows gathered in the four-state walk; operation lists queued and run as one (codon parts)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call15.log; : > $OUT
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64 or hazard or multi_partition" 2>&1 | tail -3 | tee -a $OUT
timeout 1200 python -m pytest tests/test_mrbayes_dropin.py -x -q -m gpu -k "double" 2>&1 | tail -3 | tee -a $OUT
timeout 600 python tools/f64_bench.py c4 c2 c5 c3 2>&1 | tail -4 | tee -a $OUT
cd /tmp
for c in c4 c5; do
rm -rf /tmp/pf; F64_STEPS=10 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o x -- python $GRAFT_REPO_ROOT/tools/f64_bench.py $c > /tmp/pf.log 2>&1
db=$(find /tmp/pf -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db | grep -i "k64\|Name" | cut -c1-200 | tee -a $GRAFT_REPO_ROOT/$OUT
done
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES" "WRITE_SIZE" "FETCH_SIZE"; do
  rm -rf /tmp/pm; F64_STEPS=5 timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/f64_bench.py c4 > /tmp/pm.log 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  echo "== PMC $pass" | tee -a $GRAFT_REPO_ROOT/$OUT
  [[ -n "$f" ]] && python - "$f" <<'PY' | tee -a $GRAFT_REPO_ROOT/$OUT
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if "k64_walk4" in k:
        print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, 'dispatches', len(next(iter(v.values()))))
PY
done
