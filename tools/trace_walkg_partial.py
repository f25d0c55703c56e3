"""Timing experiment: in-kernel clock stamps of the 20/61-state tree-walk kernel on a PARTIAL update (one branch changed).
Needs a library built with -DMBAMD_WG_TRACE (MBAMD_LIBRARY=...).  python tools/trace_walkg_partial.py wag|m3"""
import ctypes as C, os, sys
import numpy as np
os.environ["MBAMD_WALK_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import synthetic_division
model = sys.argv[1] if len(sys.argv) > 1 else "wag"
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
    t.length[deep] *= 1.1
    bd.TouchBranch(0, deep)
    bd.LogLike(0); bd.AcceptMove(0)
out = np.zeros((4096, 8, 3), dtype=np.int64)
ns, nw = C.c_int(0), C.c_int(0)
lib.lib.mbamdWalkTrace.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.lib.mbamdWalkTrace(bd.inst.id, out.ctypes.data, 4096, C.byref(ns), C.byref(nw))
ns, nw = ns.value, nw.value
t0 = out[:ns, :nw, 0].astype(np.float64); t1 = out[:ns, :nw, 1].astype(np.float64); ctl = out[:ns, :nw, 2]
print("depth", depth(deep), "entries", ns, "waves", nw, "total ticks", t0[-1].max() - t0[0].min())
for w in range(nw):
    d = np.diff(t0[:, w]); jobs = (t1[:, w] - t0[:, w])[:-1]
    print("wave", w, "per entry (ticks, n = NOP):", " ".join("%d%s" % (x, "n" if c & 1 else "") for x, c in zip(d.astype(int), ctl[:, w])))
    print("        chunks phase:", " ".join("%d" % x for x in jobs.astype(int)))
