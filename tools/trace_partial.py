"""Timing experiment: in-kernel step trace of a PARTIAL update (one branch changed: path to the root)."""
import ctypes as C, os, sys
import numpy as np
os.environ["MBAMD_WALK_TRACE"] = "1"
os.environ["MBAMD_VERBOSE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import synthetic_division
scaling = lk.MB_BEAGLE_SCALE_DYNAMIC if (len(sys.argv) > 1 and sys.argv[1] == "dynamic") else lk.MB_BEAGLE_SCALE_ALWAYS
div = synthetic_division("gtr", 500, 20000, seed=7, tree_seed=3)
lib = bg.library()
bd = lk.BeagleDivision(div, lib, scaling=scaling)
bd.LogLike(0); bd.AcceptMove(0)
t = div.tree
def depth(i):
    d = 0
    while t.anc[i] != -1 and t.anc[i] != t.root:
        i = t.anc[i]; d += 1
    return d
deep = max(range(t.ntaxa), key=depth)
for rep in range(3):
    t.length[deep] *= 1.1
    bd.TouchBranch(0, deep)
    bd.LogLike(0); bd.AcceptMove(0)
out = np.zeros((4096, 8, 3), dtype=np.int64)
ns, nw = C.c_int(0), C.c_int(0)
lib.lib.mbamdWalkTrace.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.lib.mbamdWalkTrace(bd.inst.id, out.ctypes.data, 4096, C.byref(ns), C.byref(nw))
ns, nw = ns.value, nw.value
tt = out[:ns, :nw, :].astype(np.float64)
print("depth", depth(deep), "steps", ns, "waves", nw, "total cycles", tt[-1, :, 2].max() - tt[0, :, 0].min())
work = tt[:, :, 1] - tt[:, :, 0]; wait = tt[:, :, 2] - tt[:, :, 1]
for s in range(min(ns, 14)):
    print(s, " ".join("%6.0f" % x for x in work[s]), "| wait", " ".join("%6.0f" % x for x in wait[s]))

print("loader phases (cycles from step start): commit done, issue done, children done, end")
L = nw - 1
for s in range(min(ns, 14)):
    t0 = out[s, L, 0]
    print(s, out[s, 7, 0] - t0, out[s, 7, 1] - t0, out[s, 7, 2] - t0, out[s, L, 1] - t0)
