"""Timing experiment: per-step s_memtime stamps of workgroup 0 of the tree-walk kernel (MBAMD_WALK_TRACE)."""
import ctypes as C, os, sys
import numpy as np
os.environ["MBAMD_WALK_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import synthetic_division
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
shape = {"c2": (500, 20000, 7, 3), "c4": (1000, 50000, 6, 10)}[cfg]
div = synthetic_division("gtr", shape[0], shape[1], seed=shape[2], tree_seed=shape[3])
lib = bg.library()
bd = lk.BeagleDivision(div, lib)
for i in range(4):
    bd.TouchAllTreeNodes(0); bd.LogLike(0); bd.AcceptMove(0)
out = np.zeros((4096, 8, 3), dtype=np.int64)
ns, nw = C.c_int(0), C.c_int(0)
lib.lib.mbamdWalkTrace.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
rc = lib.lib.mbamdWalkTrace(bd.inst.id, out.ctypes.data, 4096, C.byref(ns), C.byref(nw))
ns, nw = ns.value, nw.value
t = out[:ns, :nw, :].astype(np.float64)
t0 = t[0, :, 0].min()
print("steps", ns, "waves", nw, "total cycles", t[-1, :, 2].max() - t0)
work = t[:, :, 1] - t[:, :, 0]      # top of step -> before barrier
wait = t[:, :, 2] - t[:, :, 1]      # in barrier
step = np.diff(t[:, 0, 0])
print("mean step period (cycles of s_memtime @100MHz?)", step.mean(), "min", step.min(), "max", step.max())
for w in range(nw):
    print("wave", w, "writer" if w == nw - 1 else "compute", "work mean %.0f  barrier-wait mean %.0f" % (work[:, w].mean(), wait[:, w].mean()))
print("first 12 steps, per wave work:")
for s in range(min(12, ns)):
    print(s, " ".join("%6.0f" % x for x in work[s]), "| wait", " ".join("%6.0f" % x for x in wait[s]))
