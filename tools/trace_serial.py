"""Timing experiment: in-kernel per-operation clock stamps of the serial general-state kernel (partial update)."""
import ctypes as C, os, sys
import numpy as np
os.environ["MBAMD_WALK_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import synthetic_division
model = sys.argv[1] if len(sys.argv) > 1 else "m3"
full = len(sys.argv) > 2 and sys.argv[2] == "full"        # trace a FULL evaluation run by the serial kernel
if full:
    os.environ["MBAMD_MFMA_SERIAL"] = "100000"
shape = {"wag": (200, 10000), "m3": (100, 5000)}[model]
div = synthetic_division(model, shape[0], shape[1], seed=7, tree_seed=3)
lib = bg.library()
bd = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
bd.LogLike(0); bd.AcceptMove(0)
t = div.tree
def depth(i):
    d = 0
    while t.anc[i] != -1 and t.anc[i] != t.root:
        i = t.anc[i]; d += 1
    return d
deep = max(range(t.ntaxa), key=depth)
for rep in range(3):
    if full:
        bd.TouchAllTreeNodes(0)
    else:
        t.length[deep] *= 1.1
        bd.TouchBranch(0, deep)
    bd.LogLike(0); bd.AcceptMove(0)
out = np.zeros((4096, 8, 3), dtype=np.int64)
ns, nw = C.c_int(0), C.c_int(0)
lib.lib.mbamdWalkTrace.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.lib.mbamdWalkTrace(bd.inst.id, out.ctypes.data, 4096, C.byref(ns), C.byref(nw))
ns, nw = ns.value, nw.value
tt = out[:ns, :nw, :].astype(np.float64)
print("depth", depth(deep), "ops", ns, "waves", nw, "total (shader clocks)", tt[-1, :, 2].max() - tt[0, :, 0].min())
if full:
    per = (tt[:, :, 2].max(axis=1) - tt[:, :, 0].min(axis=1))
    print("per-op ticks: min %.0f  p25 %.0f  median %.0f  p75 %.0f  max %.0f  sum %.0f" % (per.min(), np.percentile(per, 25), np.median(per), np.percentile(per, 75), per.max(), per.sum()))
    print(" ".join("%.0f" % x for x in per))
for s in range(0 if not full else ns, ns):
    nxt = tt[s + 1, :, 0] if s + 1 < ns else tt[s, :, 2]
    print(s, "op (start -> end)", " ".join("%6.0f" % x for x in tt[s, :, 2] - tt[s, :, 0]), "| to next start", " ".join("%5.0f" % x for x in nxt - tt[s, :, 2]))
