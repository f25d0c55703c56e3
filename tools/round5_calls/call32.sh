#!/bin/bash
# round 5, call 32: k_path4's chunk (operations whose siblings are requested back to back before their chain runs): 8 / 16 / 24 / 32
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in product p4c8 p4c24 p4c32; do
  lib=""; [ $v != product ] && lib=$GRAFT_REPO_ROOT/build_x/libhmsbeagle_$v.so
  rm -rf /tmp/pv; (cd /tmp && env ${lib:+MBAMD_LIBRARY=$lib} timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pv -o v -- python $GRAFT_REPO_ROOT/tools/partial_time.py gtr 600 > /tmp/pv.log 2>&1)
  db=$(find /tmp/pv -name "*.db" | head -1)
  echo "== $v"; tail -1 /tmp/pv.log; python - $db <<'PY'
import sqlite3, sys, statistics
con = sqlite3.connect(sys.argv[1])
d = sorted(r[0] for r in con.execute('select duration from kernels where name like "%k_path4%"').fetchall())
print("k_path4 launches", len(d), "median us %.2f  mean %.2f  p10 %.2f  p90 %.2f" % (statistics.median(d) / 1e3, sum(d) / len(d) / 1e3, d[len(d) // 10] / 1e3, d[9 * len(d) // 10] / 1e3))
PY
done 2>&1 | tee gpurun_out/r5c32.log
