#!/bin/bash
# Round 4, GPU call 11: where the fp64 four-state walk's time goes (ablation builds, counters)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call11.log; : > $OUT
for v in "" old nt nostore nobar nofetch nofetchstore; do
  L=""; [[ -n "$v" ]] && L=$PWD/build_x/libhmsbeagle_$v.so
  MBAMD_LIBRARY=$L timeout 300 python tools/f64_bench.py c4 2>&1 | tail -1 | tee -a $OUT
done
MBAMD_F64_NO_WALK=1 timeout 300 python tools/f64_bench.py c4 2>&1 | tail -1 | sed 's/^/levels: /' | tee -a $OUT
cd /tmp
for v in "" old; do
  L=""; [[ -n "$v" ]] && L=$GRAFT_REPO_ROOT/build_x/libhmsbeagle_$v.so
  rm -rf /tmp/pf; MBAMD_LIBRARY=$L F64_STEPS=10 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o x -- python $GRAFT_REPO_ROOT/tools/f64_bench.py c4 > /tmp/pf.log 2>&1
  db=$(find /tmp/pf -name "*.db" | head -1); echo "== kernel stats ${v:-new}" | tee -a $GRAFT_REPO_ROOT/$OUT
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db | grep -i "k64\|Name" | cut -c1-200 | tee -a $GRAFT_REPO_ROOT/$OUT
done
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ"; do
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
