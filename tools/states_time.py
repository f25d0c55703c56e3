"""Timing experiment: full-tree evaluation of a random reversible model with S states (2: restriction sites, 8: covarion
nucleotides, 16: doublets) on the tree-walk kernel and -- MBAMD_NO_WALKG=1 in a second process -- on the level kernels.
usage: states_time.py S [ntaxa npat]"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ntaxa = int(sys.argv[2]) if len(sys.argv) > 2 else 200
npat = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
if os.environ.get("STATES_TIME_CHILD") is None:
    for tag, env in (("tree walk", {}), ("level kernels", {"MBAMD_NO_WALKG": "1"})):
        e = dict(os.environ, STATES_TIME_CHILD="1", **env)
        out = subprocess.run([sys.executable, __file__, str(S), str(ntaxa), str(npat)], env=e, capture_output=True, text=True)
        print("%-14s %s" % (tag, out.stdout.strip() or out.stderr.strip()[-300:]))
    sys.exit(0)
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import synthetic_division
div = synthetic_division("gen%d" % S, ntaxa, npat, seed=7, tree_seed=3, alpha=0.7, ncat=4)
bd = lk.BeagleDivision(div, bg.library(), scaling=lk.MB_BEAGLE_SCALE_ALWAYS)
lnl = bd.LogLike(0); bd.AcceptMove(0)
ts = []
bd.inst.kernel_timing(True); bd.inst.get_step_timing(reset=True)
for rep in range(60):
    bd.TouchAllTreeNodes(0)
    t0 = time.perf_counter()
    lnl = bd.LogLike(0)
    ts.append(time.perf_counter() - t0)
    bd.AcceptMove(0)
ms, spans = bd.inst.get_step_timing()
tm = "%.1f us" % (ms / max(spans, 1) * 1e3)
print("%d states %d x %d K=4: %s; evaluation median %.1f us through the Python twin, lnL %.4f%s"
      % (S, ntaxa, npat, bd.inst.details.implName.decode(), np.median(ts[10:]) * 1e6, lnl, ("; device time of all kernels %s" % (tm,)) if tm else ""))
