"""Device time of the partial updates of branch moves (k_path4_lnl: the root-ward path and the log-likelihood in one launch) at a BASELINE
shape, through the python twin of src/mbbeagle.c.  usage: path_time.py [case] [moves]   (MBAMD_LIBRARY selects the library)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from mrbayes_amd import beagle as bg, likelihood as lk
from tests.engine_checks import division_from_golden
case = sys.argv[1] if len(sys.argv) > 1 else "bench_c2"
moves = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lib = bg.BeagleLibrary()
div = division_from_golden(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"), case)
t = div.tree
bd = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
try:
    bd.LogLike(0)
    bd.AcceptMove(0)
    bd.inst.kernel_timing(True)
    rng = np.random.default_rng(5)
    nodes = [i for i in range(len(t.anc)) if t.anc[i] != -1 and i != t.root]
    ops = 0
    for rep in range(moves):
        b = int(rng.choice(nodes))
        t.length[b] *= 1.1 if rep % 2 else 0.9
        bd.TouchBranch(0, b)
        bd.LogLike(0)
        bd.AcceptMove(0)
    kms, kn = bd.inst.get_kernel_timing()
    sms, sn = bd.inst.get_step_timing()
    print("%s: %d moves: partials launches %d, %.2f us each; all kernels of an evaluation %.2f us; lists %s" %
          (case, moves, kn, kms / max(kn, 1) * 1e3, sms / max(sn, 1) * 1e3, bd.inst.get_list_counts()))
finally:
    bd.finalize()
