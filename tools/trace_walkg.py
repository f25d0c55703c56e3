"""Timing experiment: in-kernel clock stamps of the 20/61-state tree-walk kernel (one workgroup), full evaluation.

    python tools/trace_walkg.py wag|m3
"""
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
for rep in range(3):
    bd.TouchAllTreeNodes(0)
    bd.LogLike(0); bd.AcceptMove(0)
out = np.zeros((4096, 8, 3), dtype=np.int64)
ns, nw = C.c_int(0), C.c_int(0)
lib.lib.mbamdWalkTrace.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.lib.mbamdWalkTrace(bd.inst.id, out.ctypes.data, 4096, C.byref(ns), C.byref(nw))
ns, nw = ns.value, nw.value
t0 = out[:ns, :nw, 0].astype(np.float64)
t1 = out[:ns, :nw, 1].astype(np.float64)
ctl = out[:ns, :nw, 2]
print("entries", ns, "waves", nw, "total ticks", t0[-1].max() - t0[0].min())
kinds = {0: "int-int", 0x20: "tip-int", 0x40: "int-tip", 0x60: "tip-tip"}
for w in range(nw):
    d = np.diff(t0[:, w])
    jobs = (t1[:, w] - t0[:, w])[:-1]
    print("wave", w, "per-entry ticks: median %.0f mean %.0f max %.0f" % (np.median(d), d.mean(), d.max()))
    for name, sel in (("NOP", (ctl[:-1, w] & 1) == 1),) + tuple((v, ((ctl[:-1, w] & 1) == 0) & ((ctl[:-1, w] & 0x60) == k)) for k, v in kinds.items()):
        if sel.any():
            print("   %-8s n=%3d  entry median %.0f mean %.0f | chunks (fetch + MFMA) median %.0f | epilogue median %.0f" %
                  (name, sel.sum(), np.median(d[sel]), d[sel].mean(), np.median(jobs[sel]), np.median(d[sel] - jobs[sel])))
    print("   first 40:", " ".join("%d%s" % (x, "n" if c & 1 else ("b" if c & 2 else "")) for x, c in zip(d[:40].astype(int), ctl[:40, w])))
