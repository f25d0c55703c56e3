"""Timing experiment: in-kernel clock stamps of k_walkg_s (wave 0 of one workgroup of the first launch of a full evaluation).
Needs a library built with -DMBAMD_WGS_TRACE (tools/build_variants.py wgs_trace=MBAMD_WGS_TRACE; MBAMD_LIBRARY=build_x/...).

    python tools/trace_walkgs.py wag|m3
"""
import ctypes as C, os, sys
import numpy as np
os.environ["MBAMD_WALK_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import synthetic_division
model = sys.argv[1] if len(sys.argv) > 1 else "m3"
shape = {"wag": (200, 10000), "m3": (100, 5000)}[model]
NQ = {"wag": 2, "m3": 4}[model]
CH = NQ // 2
div = synthetic_division(model, shape[0], shape[1], seed=7, tree_seed=3)
lib = bg.library()
bd = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_ALWAYS)
for rep in range(3):
    bd.TouchAllTreeNodes(0)
    bd.LogLike(0); bd.AcceptMove(0)
out = np.zeros(4096 * 8 * 3, dtype=np.int64)
ns, nw = C.c_int(0), C.c_int(0)
lib.lib.mbamdWalkTrace.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
rc = lib.lib.mbamdWalkTrace(bd.inst.id, out.ctypes.data, 4096, C.byref(ns), C.byref(nw))
st = out.reshape(-1, 8)
n = 0
while n < len(st) and st[n, 0] != 0: n += 1
st = st[:n]
print("rc", rc, "chunks traced", n, "entries", n // NQ, "total ticks", (st[-1, 5] - st[0, 0]) if n else 0)
names = ["issue DMA/loads", "compute", "epilogue", "vmcnt wait", "barrier"]
rows = {}
for c in range(n):
    j, q = divmod(c, NQ)
    ctl = int(st[j * NQ, 6])
    ch = q // CH
    kind = "tip" if ctl & (0x40 if ch else 0x20) else ("mem" if ctl & (0x08 if ch else 0x04) else "int")
    key = (q, kind)
    d = [st[c, p + 1] - st[c, p] for p in range(5)]
    gap = (st[c + 1, 0] - st[c, 5]) if c + 1 < n else 0
    rows.setdefault(key, []).append(d + [gap, st[c, 5] - st[c, 0]])
print("%-10s %5s | %s | next-gap | chunk total   (median ticks)" % ("chunk", "n", " | ".join("%15s" % x for x in names)))
for key in sorted(rows):
    a = np.array(rows[key], dtype=np.float64)
    print("q%d %-7s %5d | %s | %8.0f | %8.0f" % (key[0], key[1], len(a), " | ".join("%15.0f" % np.median(a[:, i]) for i in range(5)), np.median(a[:, 5]), np.median(a[:, 6])))
tot = np.array([r for v in rows.values() for r in v], dtype=np.float64)
print("sum over all chunks: " + ", ".join("%s %.0f" % (names[i], tot[:, i].sum()) for i in range(5)) + ", gaps %.0f" % tot[:, 5].sum())
