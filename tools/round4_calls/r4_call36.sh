#!/bin/bash
# Round 4, GPU call 36: counters of the fp64 codon evaluation's level kernels (what bounds 1.2 - 2.2 us per operation?)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call36.log; : > $OUT
cd /tmp
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "WRITE_SIZE" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pm; F64_STEPS=3 timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/f64_bench.py c5 > /tmp/pm.log 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  echo "== PMC $pass" | tee -a $OUT
  if [[ -n "$f" ]]; then python - "$f" <<'PY' | tee -a $OUT
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "k64_partials" not in r['Kernel_Name']: continue
    key=(r['Kernel_Name'].split('(')[0][-30:], r.get('Grid_Size', r.get('Grid_Size_Y','?')))
    acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items(), key=lambda kv: -int(kv[0][1]) if kv[0][1].isdigit() else 0)[:6]:
    print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, 'dispatches', len(next(iter(v.values()))))
PY
  else tail -3 /tmp/pm.log | tee -a $OUT; fi
done
