"""Timing experiment: a substitution-model move (new rate matrices -> eigen-systems -> every transition matrix -> full
evaluation) with the eigen-systems from the host (numpy eigh + beagleSetEigenDecomposition, what MrBayes' UpDateCijk does
with its own solver) and from the device (mbamdSetRateMatrices).  usage: eigen_time.py wag|m3 [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk, model as mbmodel
from mrbayes_amd.division import synthetic_division

kind = sys.argv[1] if len(sys.argv) > 1 else "m3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
shape = {"gtr": (500, 20000), "wag": (200, 10000), "m3": (100, 5000)}[kind]
div = synthetic_division(kind, shape[0], shape[1], seed=7, tree_seed=3)
lib = bg.library()
for device in (False, True):
    bd = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC, device_eigen=device)
    bd.LogLike(0); bd.AcceptMove(0)
    t_model, t_total = [], []
    for rep in range(n):
        t0 = time.perf_counter()
        if not device:                                # the host solver is part of the move
            div.eigen = [mbmodel.eigen_reversible(q, div.pi) for q in div.rate_matrices]
        t1 = time.perf_counter()
        bd.UpDateCijk(0)
        bd.TouchAllTreeNodes(0)
        lnl = bd.LogLike(0)
        t2 = time.perf_counter()
        bd.AcceptMove(0)
        t_model.append(t1 - t0); t_total.append(t2 - t0)
    print("%s %dx%d, eigen-systems on the %s: model move + full evaluation median %.1f us (host eigen-solver %.1f us), lnL %.4f"
          % (kind, shape[0], shape[1], "device" if device else "host", np.median(t_total[5:]) * 1e6, np.median(t_model[5:]) * 1e6, lnl))
    bd.finalize()

# the device solver runs one workgroup per matrix: a batch (several chains' models at once) costs what one matrix costs
import ctypes as C
bd = lk.BeagleDivision(div, lib, nchains=1)
S = div.nstates
for count in (1, 3, 8):
    if count > len(bd.div.eigen) * 8:
        break
    inst = bg.BeagleInstance(lib, 2, 2, 2, S, 64, count, 2, 1, 1)
    qs = np.stack([div.rate_matrices[i % len(div.rate_matrices)] for i in range(count)])
    inst.set_rate_matrices(0, qs, div.pi)
    inst.lib.mbamdSynchronize(inst.id)
    t0 = time.perf_counter()
    for rep in range(10):
        inst.set_rate_matrices(0, qs, div.pi)
    inst.lib.mbamdSynchronize(inst.id)
    dev = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for rep in range(10):
        es = [mbmodel.eigen_reversible(q, div.pi) for q in qs]
        for i, e in enumerate(es):
            inst.set_eigen_decomposition(i, e.evec, e.ivec, e.eval)
    inst.lib.mbamdSynchronize(inst.id)
    host = (time.perf_counter() - t0) / 10
    # warm start (mbamdSetRateMatricesFrom): a slightly perturbed model decomposed from the previous eigenvectors, like an MCMC move
    inst2 = bg.BeagleInstance(lib, 2, 2, 2, S, 64, 2 * count, 2, 1, 1)
    inst2.set_rate_matrices(0, qs, div.pi)
    rng = np.random.default_rng(1)
    pert = []
    for rep in range(12):
        f = np.exp(0.04 * (rng.random((S, S)) - 0.5)); f = 0.5 * (f + f.T)
        p2 = []
        for q in qs:
            e = q / np.asarray(div.pi)[None, :] * f
            q2 = e * np.asarray(div.pi)[None, :]
            np.fill_diagonal(q2, 0.0); np.fill_diagonal(q2, -q2.sum(axis=1))
            p2.append(q2)
        pert.append(np.stack(p2))
    cur = 0
    inst2.set_rate_matrices_from(count, pert[0], div.pi, 0); cur = count
    inst2.lib.mbamdSynchronize(inst2.id)
    t0 = time.perf_counter()
    for rep in range(1, 11):
        inst2.set_rate_matrices_from(count - cur, pert[rep], div.pi, cur); cur = count - cur
    t1 = time.perf_counter()
    inst2.lib.mbamdSynchronize(inst2.id)
    warm = (time.perf_counter() - t0) / 10
    print("%d eigen-systems of %d states: device %.0f us cold, %.0f us warm-started (host time of the asynchronous call %.0f us), host solver + beagleSetEigenDecomposition %.0f us"
          % (count, S, dev * 1e6, warm * 1e6, (t1 - t0) / 10 * 1e6, host * 1e6))
    inst.finalize()
    inst2.finalize()
