#!/usr/bin/env python3
"""Timing experiment (needs an experiment build of the engine whose k_walk4_t writes clock stamps: not the product library): where does
a root-ward path walk spend its time?  Clock stamps of wave 0 of workgroup 0 per entry: start | rare section done | result computed |
scalar burst issued | stores issued.       MBAMD_WALK_TRACE=1 MBAMD_LIBRARY=build_x/libhmsbeagle_trace4.so python tools/trace_walk4.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk                    # noqa: E402
from mrbayes_amd.division import synthetic_division                        # noqa: E402

div = synthetic_division("gtr", 500, 20000, seed=7, tree_seed=3)
lib = bg.library()
bd = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
bd.LogLike(0)
bd.AcceptMove(0)
t = div.tree
rng = np.random.default_rng(5)
nodes = [i for i in range(len(t.anc)) if t.anc[i] != -1 and i != t.root]
raw = lib.lib if hasattr(lib, "lib") else lib._lib
raw.mbamdWalkTrace.argtypes = [C.c_int, C.POINTER(C.c_longlong), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
for rep in range(6):
    b = int(rng.choice(nodes))
    t.length[b] *= 1.07
    bd.TouchBranch(0, b)
    bd.LogLike(0)
    bd.AcceptMove(0)
    buf = (C.c_longlong * (4096 * 24))()
    steps, waves = C.c_int(0), C.c_int(0)
    rc = raw.mbamdWalkTrace(bd.inst.id, buf, 4096, C.byref(steps), C.byref(waves))
    if rc != 0:
        print("mbamdWalkTrace", rc)
        break
    a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 24)[:steps.value]
    a = a[a[:, 0] != 0]
    if rep < 2 or len(a) == 0:
        continue
    print("update %d: %d entries, total %d ticks (100 MHz clock: %.1f us)" % (rep, len(a), a[-1, 4] - a[0, 0], (a[-1, 4] - a[0, 0]) / 100.0))
    print(" entry  kind   rare  compute  burst  tail | gap to next     (ticks of 10 ns)")
    for j in range(len(a)):
        ctl = int(a[j, 6])
        kind = "PF " if ctl & 4 else ("nop" if ctl & 1 else "op ")
        gap = (a[j + 1, 0] - a[j, 4]) if j + 1 < len(a) else 0
        print("  %3d   %s  %5d  %6d  %5d  %5d | %5d" % (j, kind, a[j, 1] - a[j, 0], a[j, 2] - a[j, 1], a[j, 3] - a[j, 2], a[j, 4] - a[j, 3], gap))
bd.finalize()
