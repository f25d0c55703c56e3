"""Timing experiment: wall time of PARTIAL updates (one branch length changed -> the path to the root is
recomputed), the evaluation an MCMC generation actually costs.  usage: partial_time.py gtr|wag|m3 [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import synthetic_division

model = sys.argv[1] if len(sys.argv) > 1 else "wag"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
shape = {"gtr": (500, 20000), "wag": (200, 10000), "m3": (100, 5000)}[model]
div = synthetic_division(model, shape[0], shape[1], seed=7, tree_seed=3)
lib = bg.library()
bd = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
bd.LogLike(0); bd.AcceptMove(0)
t = div.tree
rng = np.random.default_rng(5)
nodes = [i for i in range(len(t.anc)) if t.anc[i] != -1 and i != t.root]
times = []
for rep in range(n):
    b = int(rng.choice(nodes))
    t.length[b] *= float(np.exp(0.2 * (rng.random() - 0.5)))
    bd.TouchBranch(0, b)
    t0 = time.perf_counter()
    lnl = bd.LogLike(0)
    times.append(time.perf_counter() - t0)
    bd.AcceptMove(0)
times = np.array(times[20:]) * 1e6
bd2 = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
full = bd2.LogLike(0)
print("%s %dx%d: partial update median %.1f us  p10 %.1f  p90 %.1f   lnL %.4f  fresh full evaluation %.4f  rel %.2e"
      % (model, shape[0], shape[1], np.median(times), np.percentile(times, 10), np.percentile(times, 90), lnl, full,
         abs(lnl - full) / abs(full)))
