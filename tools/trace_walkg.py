"""Timing experiment: where a wave of the general-state tree walk spends its cycles, by entry type and phase.  Needs a library built
with -DMBAMD_WG_STAMPS (tools/build_variants.py stamps=MBAMD_WG_STAMPS): python tools/trace_walkg.py c5|c3  (MBAMD_LIBRARY=...)"""
import ctypes as C, os, sys
import numpy as np
os.environ["MBAMD_WALK_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import division_from_golden
case = {"c3": "bench_c3", "c5": "bench_c5"}[sys.argv[1] if len(sys.argv) > 1 else "c5"]
div = division_from_golden(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"), case)
lib = bg.library()
bd = lk.BeagleDivision(div, lib, nchains=1, scaling=lk.MB_BEAGLE_SCALE_ALWAYS)
try:
    bd.LogLike(0)
except Exception as e:
    print("first evaluation:", str(e)[:80])
bd.AcceptMove(0)
evals = [lk.record_evaluation(bd), lk.record_evaluation(bd)]
for i in range(6):
    evals[i & 1].run()
bd.inst.kernel_timing(True)
bd.inst.get_kernel_timing(reset=True)
for i in range(20):
    evals[i & 1].run()
kms, kl = bd.inst.get_kernel_timing(reset=True)
print("kernel %.4f ms" % (kms / 20))
NT = 4096
out = np.zeros((NT, 8, 3), dtype=np.int64)
ns, nw = C.c_int(0), C.c_int(0)
lib.lib.mbamdWalkTrace.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
rc = lib.lib.mbamdWalkTrace(bd.inst.id, out.ctypes.data, NT, C.byref(ns), C.byref(nw))
flat = out.reshape(-1)
names = ["int-int", "tip-int", "tip-tip", "no-op"]
print(case, os.environ.get("MBAMD_LIBRARY", "product"), "rc", rc)
print("%-8s %6s | per entry: %8s %8s %8s %8s %8s | %9s" % ("wave", "n", "topwait", "child1", "child2", "epi+max", "rest", "total"))
for w in range(8):
    a = flat[w * 24:(w + 1) * 24].reshape(4, 6).astype(float)
    if a[:, :5].sum() == 0:
        continue
    tot = 0.0
    for q in range(4):
        n = a[q, 5]
        if n == 0:
            continue
        tot += a[q, :5].sum()
        print("w%d %-7s %4d | %8.0f %8.0f %8.0f %8.0f %8.0f | %9.0f" % ((w, names[q], n) + tuple(a[q, :5] / n) + (a[q, :5].sum() / n,)))
    print("w%d all cycles %.0f" % (w, tot))

# per workgroup and physical wave: begin / end (100 MHz), HW_ID, XCC_ID
sched = flat[1024:1024 + 4096 * 4].reshape(-1, 4)
rows = [(i // 4, i % 4, r) for i, r in enumerate(sched) if r[0] != 0]
if rows:
    t0 = min(r[2][0] for r in rows)
    from collections import defaultdict
    percu = defaultdict(list)
    for wg, w, r in rows:
        hw = int(r[2]); cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
        percu[(int(r[3]) & 15, se, sh, cu)].append((wg, w, simd, (r[0] - t0) / 100.0, (r[1] - t0) / 100.0))
    ends = [(r[2][1] - t0) / 100.0 for r in rows]
    durs = [(r[2][1] - r[2][0]) / 100.0 for r in rows]
    print("workgroup-waves %d, CUs used %d, kernel span %.1f us; wave duration min %.1f median %.1f max %.1f us; begin max %.1f us" % (
        len(rows), len(percu), max(ends), min(durs), float(np.median(durs)), max(durs), max((r[2][0] - t0) / 100.0 for r in rows)))
    hist = defaultdict(int)
    for k, v in percu.items():
        hist[len(set(x[0] for x in v))] += 1
    print("workgroups per CU -> CUs:", dict(hist))
    bysimd = defaultdict(int)
    for k, v in percu.items():
        for x in v:
            bysimd[x[2]] += 1
    print("waves per SIMD id:", dict(bysimd))
    for k in sorted(percu)[:3]:
        print(k, [(x[0], x[1], x[2], round(x[3], 1), round(x[4], 1)) for x in percu[k]])
    # duration by the number of workgroups on the wave's CU
    # phase sums of k_walkg2 (MBAMD_WG_STAMPS=2): [workgroup * 4 + wave][type][operands, top wait, factors, exchange, scale+keep, stores, n]
    ph = flat[1024 + 4096 * 4:].view(np.uint32)[:640 * 4 * 28].reshape(640 * 4, 4, 7).astype(float)
    if ph.sum() > 0:
        names = ["int-int", "tip-int", "tip-tip", "no-op"]
        shared = set()
        for k, v in percu.items():
            if len(set(x[0] for x in v)) > 1:
                shared.update(x[0] * 4 + x[1] for x in v)
        allw = [wg * 4 + w for wg, w, r in rows if wg < 640]
        for label, sel in (("waves on a CU of their own", [i for i in allw if i not in shared]), ("waves that share their SIMD", [i for i in allw if i in shared])):
            if not sel:
                continue
            a = ph[sel].sum(axis=0)
            print(label, "(%d): cycles per entry  %8s %8s %8s %8s %8s %8s | %8s" % (len(sel), "operands", "topwait", "factors", "exchange", "scale", "stores", "total"))
            tot = 0.0
            for q in range(4):
                if a[q, 6] > 0:
                    print("   %-8s n/wave %5.1f | %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f | %8.0f" % ((names[q], a[q, 6] / len(sel)) + tuple(a[q, :6] / a[q, 6]) + (a[q, :6].sum() / a[q, 6],)))
                    tot += a[q, :6].sum() / len(sel)
            print("   cycles per wave %.0f" % tot)
    bycount = defaultdict(list)
    for k, v in percu.items():
        n = len(set(x[0] for x in v))
        for x in v:
            bycount[n].append(x[4] - x[3])
    for n in sorted(bycount):
        print("CUs with %d workgroup(s): wave duration median %.1f us (min %.1f, max %.1f)" % (n, float(np.median(bycount[n])), min(bycount[n]), max(bycount[n])))
