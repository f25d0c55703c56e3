"""Parity checks of the engine (through its C ABI) against the CPU oracle and the golden fixtures.

The same checks run in two harnesses:
  * tests/test_engine_gpu.py      (-m gpu)      the product library mrbayes_amd/libhmsbeagle.so on a MI355X
  * tests/test_engine_hostemu.py  (-m "not gpu") the host-emulation build of the same sources, which
                                                exercises the host logic and kernel indexing without a GPU
Tolerances (stated once, used everywhere):
  * log-likelihood vs the reference's fp64 build: relative 2e-6; vs its FMA build: relative 1e-5
    (the reference's own fp32 self-consistency band is 8e-9 .. 8.4e-7);
  * log-likelihood vs the oracle on the same inputs: relative 2e-6; vs the fp64 reference also
    |diff| <= 2 x (the reference's own |fp32 - fp64| on that data set) + 1e-3;
  * partials / transition matrices element-wise: 2e-6 absolute + 2e-5 relative (fp32 round-off with
    a different summation order / FMA contraction than the scalar reference);
  * parsimony state sets and lengths: bit-exact (integer work; lengths are sums of integer-valued weights).
"""
import json
import math
import os

import numpy as np
import pytest

from mrbayes_amd import beagle as bg
from mrbayes_amd import likelihood as lk
from mrbayes_amd import tree as mbtree
from mrbayes_amd.division import build_division, division_from_golden, synthetic_division

REL_FP64 = 2e-6
REL_FMA = 1e-5


def close_partials(a, b):
    return np.allclose(a, b, rtol=2e-5, atol=2e-6)


def gold(golden_dir, case):
    with open(os.path.join(golden_dir, case + ".json")) as fh:
        return json.load(fh)


# ------------------------------------------------------------------------------------------------
def check_transition_matrices(lib, oracle, div):
    bd = lk.BeagleDivision(div, lib)
    try:
        bd.UpDateCijk(0)
        bd.TreeTiProbs_Beagle(0)
        t = div.tree
        for p in t.all_down_pass[:: max(1, len(t.all_down_pass) // 7)]:
            length = min(max(t.length[p], lk.BRLENS_MIN), lk.BRLENS_MAX)
            ref = oracle.tiprobs(div, length)                 # [K or parts][S][S]
            for i in range(div.n_cijk_parts):
                got = bd.inst.get_transition_matrix(bd.tiProbsIndex[0][p] + i)     # [ncat][S][S]
                want = ref if div.n_cijk_parts == 1 else ref[i:i + 1]
                assert got.shape == want.shape
                assert np.allclose(got, want, rtol=1e-5, atol=2e-7), (p, i, np.abs(got - want).max())
                assert got.min() >= 0.0
                assert np.allclose(got.sum(axis=2), 1.0, atol=1e-5)
    finally:
        bd.finalize()


def check_closed_form_matrices(lib, oracle):
    """nst=2 (HKY) and nst=1 (Jukes-Cantor): the reference's native path uses the closed forms TiProbs_Hky /
    TiProbs_JukesCantor (src/likelihood.c:9709, 9846); at the BEAGLE seam MrBayes sends an eigen-system for every
    nst, so the engine's eigen path must reproduce the closed forms (short, ordinary and very long branches)."""
    tr = mbtree.random_tree(12, seed=5, brlen=None, brlen_mean=0.08)
    tr.length[2], tr.length[5], tr.length[7] = 1e-8, 2.5, 100.0
    w = np.ones(8)
    tips = [np.zeros(8, dtype=np.int32) for _ in range(12)]
    for kappa, pi in ((2.0, [0.35, 0.25, 0.15, 0.25]), (7.5, [0.1, 0.4, 0.3, 0.2]), (1.0, [0.25] * 4)):
        div = build_division("gtr", tr, w, tips, [None] * 12, revmat=[1.0, kappa, 1.0, 1.0, kappa, 1.0], pi=pi, alpha=0.7, ncat=4)
        bd = lk.BeagleDivision(div, lib)
        try:
            bd.UpDateCijk(0)
            bd.TreeTiProbs_Beagle(0)
            for p in tr.all_down_pass:
                if p == tr.root:
                    continue
                idx = bd.tiProbsIndex[0][p]
                length = min(max(tr.length[p], lk.BRLENS_MIN), lk.BRLENS_MAX)
                got = bd.inst.get_transition_matrix(idx)
                want = oracle.tiprobs_hky(kappa, pi, length, div.cat_rates)
                assert np.allclose(got, want, rtol=1e-5, atol=2e-7), (kappa, p, np.abs(got - want).max())
                if kappa == 1.0:
                    assert np.allclose(got, oracle.tiprobs_jc(length, div.cat_rates), rtol=1e-5, atol=2e-7)
        finally:
            bd.finalize()


def check_device_eigen(lib, kind):
    """mbamdSetRateMatrices: eigen-systems computed on the device by a Jacobi iteration on the symmetrised
    rate matrix.  The transition matrices built from them must be exp(Q t) -- checked against scipy's Pade expm, which shares
    nothing with either eigen-solver -- and the tree likelihood must be the one the host's eigen-systems give.  Also the
    exchangeability form (Q built and normalised on the device)."""
    from scipy.linalg import expm
    from mrbayes_amd.division import synthetic_division
    from mrbayes_amd import model as mbmodel
    div = synthetic_division(kind, 14, 60, seed=17, tree_seed=18, p_gap=0.02)
    host = lk.BeagleDivision(div, lib)
    dev = lk.BeagleDivision(div, lib, device_eigen=True)
    try:
        assert dev.device_eigen
        a, b = host.LogLike(0), dev.LogLike(0)
        assert abs(a - b) <= 1e-9 * abs(a), (a, b)
        t = div.tree
        S = div.nstates
        for p in t.all_down_pass[:6]:
            if p == t.root:
                continue
            length = min(max(t.length[p], lk.BRLENS_MIN), lk.BRLENS_MAX)
            for part, q in enumerate(div.rate_matrices):
                got = dev.inst.get_transition_matrix(dev.tiProbsIndex[0][p] + part)
                for k, r in enumerate(div.cat_rates):
                    want = expm(q * (length * r))
                    assert np.allclose(got[k], want, rtol=2e-5, atol=3e-7), (kind, p, part, k, np.abs(got[k] - want).max())
        if kind == "wag":                            # exchangeabilities in, Q built on the device
            exch = div.rate_matrices[0] / np.asarray(div.pi)[None, :]
            np.fill_diagonal(exch, 0.0)
            exch = 0.5 * (exch + exch.T) * 3.7       # any scale: the device normalises to one substitution per unit time
            dev.inst.set_rate_matrices(dev.cijkIndex[0], exch[None], div.pi, exchangeabilities=True)
            idx = dev.tiProbsIndex[0][t.all_down_pass[0]]
            length = min(max(t.length[t.all_down_pass[0]], lk.BRLENS_MIN), lk.BRLENS_MAX)
            dev.inst.set_category_rates(div.cat_rates)
            dev.inst.update_transition_matrices(dev.cijkIndex[0], np.asarray([idx], dtype=np.int32), [length])
            got = dev.inst.get_transition_matrix(idx)
            want = expm(div.rate_matrices[0] * (length * div.cat_rates[0]))
            assert np.allclose(got[0], want, rtol=2e-5, atol=3e-7), np.abs(got[0] - want).max()
    finally:
        host.finalize()
        dev.finalize()


def check_device_eigen_warm_start(lib, nstates, seed=21):
    """mbamdSetRateMatricesFrom: a chain of slightly perturbed reversible rate matrices, each decomposed from the eigenvectors
    of the one before (alternating between two eigen buffers like MrBayes' FlipCijkSpace), must give exp(Q t) every time
    -- scipy's Pade expm again -- and the shield makes the next beagleSetEigenDecomposition a no-op, once."""
    from scipy.linalg import expm
    rng = np.random.default_rng(seed)
    S = nstates
    inst = bg.BeagleInstance(lib, 2, 4, 2, S, 64, 2, 2, 1, 2)
    try:
        pi = rng.random(S) + 0.2
        pi /= pi.sum()
        ex = rng.random((S, S)) + 0.1
        ex = 0.5 * (ex + ex.T)
        inst.set_category_rates([1.0])

        def qmat(e):
            q = e * pi[None, :]
            np.fill_diagonal(q, 0.0)
            np.fill_diagonal(q, -q.sum(axis=1))
            return q / -(pi * np.diag(q)).sum()

        cur = 0
        inst.set_rate_matrices(cur, qmat(ex)[None], pi)              # cold
        for step in range(70):                                      # (> 64: one cold restart inside)
            ex = ex * np.exp(0.05 * (rng.random((S, S)) - 0.5))
            ex = 0.5 * (ex + ex.T)
            q = qmat(ex)
            inst.set_rate_matrices_from(1 - cur, q[None], pi, cur, shield=(step == 3))
            cur = 1 - cur
            if step == 3:                                           # what a client's own code path sends afterwards: ignored once
                junk = np.eye(S)
                inst.set_eigen_decomposition(cur, junk, junk, np.zeros(S))
            if step in (0, 3, 4, 33, 64, 65, 69):
                inst.update_transition_matrices(cur, np.asarray([0], dtype=np.int32), [0.37])
                got = inst.get_transition_matrix(0)[0]
                want = expm(q * 0.37)
                assert np.allclose(got, want, rtol=2e-5, atol=3e-7), (S, step, np.abs(got - want).max())
        junk = np.eye(S)
        inst.set_eigen_decomposition(cur, junk, junk, np.zeros(S))      # not shielded any more: identity eigen-system
        inst.update_transition_matrices(cur, np.asarray([0], dtype=np.int32), [0.37])
        assert np.allclose(inst.get_transition_matrix(0)[0], np.eye(S), atol=1e-7)
    finally:
        inst.finalize()


def check_root_equals_edge(lib, oracle, div, scaling=lk.MB_BEAGLE_SCALE_ALWAYS):
    """beagleCalculateRootLogLikelihoods (rooted / clock trees, reference src/mbbeagle.c:1251-1257) against the edge form
    MrBayes uses for unrooted trees: fold the root branch into one more partials operation (identity matrix for the
    interior node, the branch's matrix for the root tip) and integrate the result at the root.  Same lnL, same per-site
    values; and the oracle's value."""
    bd = lk.BeagleDivision(div, lib, scaling=scaling)
    try:
        want = bd.LogLike(0)
        site_edge = bd.inst.get_site_log_likelihoods()
        inst, t = bd.inst, div.tree
        S, K = div.nstates, div.ncat
        p = t.root_left
        step = bd.step
        parents = [bd.condLikeIndex[0][p] + i for i in range(step)]
        children = [bd.condLikeIndex[0][t.root]] * step
        mats = [bd.tiProbsIndex[0][p] + i for i in range(step)]
        wts = frq = [bd.cijkIndex[0] + i for i in range(step)]
        cums = [bd.siteScalerIndex[0] + i for i in range(step)]
        ident = np.broadcast_to(np.eye(S), (K, S, S)).copy()
        roots = []
        for n in range(step):                              # the scratch set of node p is free between evaluations
            inst.set_transition_matrix(bd.tiProbsScratchIndex[p] + n, ident)
            dst = bd.condLikeScratchIndex[p] + n
            inst.update_partials(np.array([[dst, -1, -1, parents[n], bd.tiProbsScratchIndex[p] + n, children[n], mats[n]]],
                                          dtype=np.int32), bg.BEAGLE_OP_NONE)
            roots.append(dst)
        rc, got = inst.calculate_root_log_likelihoods(roots, wts, frq, cums)
        assert rc == 0
        site_root = inst.get_site_log_likelihoods()
        assert abs(got - want) <= 1e-10 * abs(want), (got, want)
        assert np.allclose(site_root, site_edge, rtol=1e-12, atol=1e-9)
        ref = oracle.tree_loglike(div, use_shortcuts=False)
        assert abs(got - ref) / abs(ref) < REL_FP64
    finally:
        bd.finalize()


def _dense_tip(states, nstates, ncat):
    P = len(states)
    cl = np.zeros((ncat, P, nstates), dtype=np.float32)
    for c, s in enumerate(states):
        if s >= nstates:
            cl[:, c, :] = 1.0
        else:
            cl[:, c, s] = 1.0
    return cl


def check_single_operations(lib, oracle, nstates, ncat, npat, seed=1):
    """One beagleUpdatePartials operation per child-kind combination, no scaling, then with scaling:
    compares the destination partials element-wise with CondLikeDown of the oracle."""
    rng = np.random.default_rng(seed)
    S, K, P = nstates, ncat, npat
    inst = bg.BeagleInstance(lib, 2, 8, 2, S, P, 1, 4, K, 4)
    try:
        # random reversible-ish transition matrices: rows sum to one
        def rand_ti():
            m = rng.random((K, S, S)) + 0.05
            return m / m.sum(axis=2, keepdims=True)
        ti1, ti2 = rand_ti(), rand_ti()
        inst.set_transition_matrix(0, ti1)
        inst.set_transition_matrix(1, ti2)
        assert np.allclose(inst.get_transition_matrix(0), ti1.astype(np.float32), atol=1e-7)
        st1 = rng.integers(0, S + 1, size=P).astype(np.int32)       # includes the missing code S
        st2 = rng.integers(0, S + 1, size=P).astype(np.int32)
        inst.set_tip_states(0, st1)
        inst.set_tip_states(1, st2)
        pa = (rng.random((K, P, S)) * 0.9 + 0.05)
        pb = (rng.random((K, P, S)) * 0.9 + 0.05) * 1e-3
        inst.set_partials(2, pa)
        inst.set_partials(3, pb)
        assert np.allclose(inst.get_partials(2), pa.astype(np.float32), rtol=1e-7)
        t1 = np.ascontiguousarray(ti1, dtype=np.float32)
        t2 = np.ascontiguousarray(ti2, dtype=np.float32)
        fa = np.ascontiguousarray(pa, dtype=np.float32)
        fb = np.ascontiguousarray(pb, dtype=np.float32)
        combos = [("states,states", 0, 1, None, st1, None, st2),
                  ("states,partials", 0, 3, None, st1, fb, None),
                  ("partials,states", 2, 1, fa, None, None, st2),
                  ("partials,partials", 2, 3, fa, None, fb, None)]
        for name, c1, c2, cl1, s1, cl2, s2 in combos:
            want = oracle.condlike_down(S, K, P, cl1, s1, t1, cl2, s2, t2)
            # the engine treats a compact tip as the dense 0/1 vector (what the reference's SIMD kernels
            # do); for a missing state that is the row sum instead of the scalar short-cut's literal 1.0
            inst.update_partials(np.array([[4, -1, -1, c1, 0, c2, 1]], dtype=np.int32), bg.BEAGLE_OP_NONE)
            got = inst.get_partials(4)
            assert close_partials(got, want), (name, np.abs(got - want).max())
            # with rescaling: got2 * 2^e == unscaled.  The 4-state path rescales every (pattern, category) column on its
            # own (max over the states in [0.5, 1)), the general-state path every pattern (max over categories and states)
            inst.reset_scale_factors(1)
            inst.update_partials(np.array([[5, 0, -1, c1, 0, c2, 1]], dtype=np.int32), 1)
            got2 = inst.get_partials(5)
            e = inst.get_scale_exponents(0).astype(np.float64)              # [K][P]
            assert np.array_equal(e, inst.get_scale_exponents(1))
            lnsc = inst.get_scale_factors(0)
            assert np.allclose(lnsc, e.max(axis=0) * math.log(2.0), atol=1e-9)
            assert np.allclose(inst.get_scale_factors(1), lnsc)
            assert np.allclose(got2 * np.exp2(e)[:, :, None], got, rtol=1e-6, atol=0)
            per_category = "tree-walk" in inst.details.implName.decode()
            mx = got2.max(axis=2) if per_category else np.broadcast_to(got2.max(axis=(0, 2)), (K, P))
            assert np.all((mx >= 0.5 - 1e-6) & (mx < 1.0 + 1e-6))
            # "divide by the factors already stored" (dynamic scaling's no-rescale pass)
            inst.update_partials(np.array([[6, -1, 0, c1, 0, c2, 1]], dtype=np.int32), bg.BEAGLE_OP_NONE)
            got3 = inst.get_partials(6)
            assert np.array_equal(got3, got2)
            # remove / accumulate are exact inverses
            inst.remove_scale_factors([0], 1)
            assert np.all(inst.get_scale_exponents(1) == 0)
            inst.accumulate_scale_factors([0, 0], 1)
            assert np.array_equal(inst.get_scale_exponents(1), 2 * inst.get_scale_exponents(0))
            # Reset + Accumulate back to back (how MrBayes rebuilds a cumulative buffer) is one launch that stores: the old
            # content must be gone; a reset followed by anything else is a plain reset; a reset of ANOTHER buffer does not leak
            e0 = inst.get_scale_exponents(0)
            inst.reset_scale_factors(1)
            inst.accumulate_scale_factors([0], 1)
            assert np.array_equal(inst.get_scale_exponents(1), e0)
            inst.reset_scale_factors(1)
            inst.accumulate_scale_factors([0, 0, 0], 1)
            assert np.array_equal(inst.get_scale_exponents(1), 3 * e0)
            inst.reset_scale_factors(1)
            assert np.all(inst.get_scale_exponents(1) == 0)
            inst.accumulate_scale_factors([0], 1)
            inst.reset_scale_factors(2)
            inst.accumulate_scale_factors([0], 1)
            assert np.array_equal(inst.get_scale_exponents(1), 2 * e0)
            assert np.all(inst.get_scale_exponents(2) == 0)
    finally:
        inst.finalize()


def check_final_pass(lib, nstates, ncat, npat, seed=7):
    """mbamdUpdateFinalPartials / mbamdGetScaledPartials against a numpy restatement of the reference's CondLikeUp_Gen /
    CondLikeUp_NUC4 (src/likelihood.c:4655-4795) on random data: a three-node path top -> a -> b, the top node with a compact
    root tip, with a partials root tip and without one (rooted); scaled read-out against the exponents the engine reports."""
    rng = np.random.default_rng(seed)
    S, K, P = nstates, ncat, npat
    inst = bg.BeagleInstance(lib, 2, 12, 2, S, P, 1, 4, K, 4)
    try:
        def rand_ti():
            m = rng.random((K, S, S)) + 0.05
            return m / m.sum(axis=2, keepdims=True)
        ti = [rand_ti() for _ in range(3)]
        for q in range(3):
            inst.set_transition_matrix(q, ti[q])
        t32 = [np.ascontiguousarray(t, dtype=np.float32) for t in ti]
        st = rng.integers(0, S + 1, size=P).astype(np.int32)
        inst.set_tip_states(0, st)
        tipvec = np.zeros((P, S), dtype=np.float32)
        for c in range(P):
            tipvec[c] = 1.0 if st[c] >= S else 0.0
            if st[c] < S:
                tipvec[c, st[c]] = 1.0
        down = [(rng.random((K, P, S)) * 0.9 + 0.05) * 10.0 ** (-q) for q in range(3)]       # down partials of top, a, b
        amb = rng.random((K, P, S)) * 0.9 + 0.05                                              # a root tip given as partials
        for q in range(3):
            inst.set_partials(2 + q, down[q])
        inst.set_partials(5, amb)
        d32 = [np.ascontiguousarray(d, dtype=np.float32) for d in down]

        def up(fa, d, t):                                # one CondLikeUp step, float32 like the reference's CLFlt
            s = np.einsum("kai,kci->kca", t, d).astype(np.float32)
            with np.errstate(divide="ignore", invalid="ignore"):
                u = np.where(s != 0, fa / s, 0).astype(np.float32)
            return (np.einsum("kci,kai->kca", u, t).astype(np.float32) * d).astype(np.float32)

        for name, root_tip, factor in (("compact root tip", 0, np.einsum("kaj,cj->kca", t32[0], tipvec)),
                                       ("partials root tip", 5, np.einsum("kaj,kcj->kca", t32[0], amb.astype(np.float32))),
                                       ("rooted", -1, None)):
            top = d32[0] if factor is None else (d32[0] * factor.astype(np.float32)).astype(np.float32)
            fa = up(top, d32[1], t32[1])
            fb = up(fa, d32[2], t32[2])
            inst.update_final_partials(np.array([[6, -1, 2, 0, root_tip], [7, 6, 3, 1, -1], [8, 7, 4, 2, -1]], dtype=np.int32))
            for buf, want in ((6, top), (7, fa), (8, fb)):
                # (the pass keeps the top node's columns in [0.5, 1) by their own powers of two; the scaled read-out reports them)
                got, ln = inst.get_scaled_partials(buf)
                true = got.astype(np.float64) * np.exp(ln.astype(np.float64))[None, :, None]
                assert np.allclose(true, want, rtol=3e-5, atol=1e-30), (name, buf, np.abs(true / want - 1).max())
            raw = inst.get_partials(6)
            assert raw.max(axis=2).min() >= 0.5 - 1e-6 and raw.max() < 1.0 + 1e-6          # every column of the top node normalised
        # MrBayes' dynamic rescaling scheme leaves the down pass unscaled until a likelihood underflows: on a 60-taxon tree its values
        # are 1e-30 ... 1e-44 floats.  The final pass must survive them (it runs in double and normalises the top node): posteriors
        # against a float64 restatement, no zero columns, nothing that is not a number.
        tiny = [np.ascontiguousarray((rng.random((K, P, S)) * 0.9 + 0.05) * 10.0 ** (-e), dtype=np.float32) for e in (36, 28, 18)]
        for q in range(3):
            inst.set_partials(2 + q, tiny[q].astype(np.float64))
        inst.update_final_partials(np.array([[6, -1, 2, 0, 0], [7, 6, 3, 1, -1], [8, 7, 4, 2, -1]], dtype=np.int32))
        t64 = [t.astype(np.float64) for t in t32]
        d64 = [t.astype(np.float64) for t in tiny]

        def up64(fa, d, t):
            s = np.einsum("kai,kci->kca", t, d)
            return np.einsum("kci,kai->kca", fa / s, t) * d
        f0 = d64[0] * np.einsum("kaj,cj->kca", t64[0], tipvec.astype(np.float64))
        f2 = up64(up64(f0, d64[1], t64[1]), d64[2], t64[2])
        got, ln = inst.get_scaled_partials(8)
        assert np.all(np.isfinite(got)) and np.all(np.isfinite(ln))
        post = got.astype(np.float64).sum(axis=0)
        post /= post.sum(axis=1, keepdims=True)
        want = f2.sum(axis=0)
        want /= want.sum(axis=1, keepdims=True)
        assert np.allclose(post, want, rtol=2e-4, atol=1e-7), np.abs(post - want).max()
        # the scaled read-out: categories brought to the largest exponent of the pattern
        inst.reset_scale_factors(1)
        inst.update_partials(np.array([[9, 0, -1, 2, 0, 3, 1]], dtype=np.int32), 1)          # rescaled, exponents also in buffer 1
        e = inst.get_scale_exponents(1).astype(np.int64)                                      # [K][P]
        raw = inst.get_partials(9)
        got, ln = inst.get_scaled_partials(9, 1)
        emax = e.max(axis=0)
        assert np.allclose(ln, emax * math.log(2.0), rtol=1e-6)
        assert np.array_equal(got, (raw * np.exp2(e - emax[None, :])[:, :, None]).astype(np.float32))
    finally:
        inst.finalize()


def check_hazard_lists(lib, nstates, ncat, npat, seed=5, double_precision=False):
    """Operation lists MrBayes never issues but the API allows: a destination that an earlier operation of the same list
    read (write-after-read) or wrote (write-after-write), an exponent buffer written by one operation and read by a
    later one, two independent lists back to back.  One call must give what one call per operation gives."""
    rng = np.random.default_rng(seed)
    S, K, P = nstates, ncat, npat

    def run(split):
        inst = bg.BeagleInstance(lib, 2, 10, 2, S, P, 1, 4, K, 6,
                                 preference_flags=bg.BEAGLE_FLAG_PRECISION_DOUBLE if double_precision else 0)
        try:
            for m in range(4):
                ti = rng2.random((K, S, S)) + 0.05
                inst.set_transition_matrix(m, ti / ti.sum(axis=2, keepdims=True))
            inst.set_tip_states(0, st1)
            inst.set_tip_states(1, st2)
            inst.set_partials(2, pa)
            ops = np.array([[3, 0, -1, 0, 0, 1, 1],       # 3 = f(tip0, tip1), writes exponents 0
                            [4, 1, -1, 3, 2, 2, 3],       # 4 = f(3, 2), writes exponents 1
                            [5, -1, 0, 3, 1, 4, 0],       # 5 = f(3, 4) divided by the exponents 0 (written above)
                            [3, 2, -1, 4, 2, 5, 3],       # 3 overwritten after it was read (WAR) and written (WAW)
                            [6, -1, 1, 3, 0, 5, 1],       # reads the new 3, divides by exponents 1
                            [4, 0, -1, 6, 1, 3, 2]],      # 4 overwritten; exponents 0 written a second time
                           dtype=np.int32)
            inst.reset_scale_factors(5)
            if split:
                for o in ops:
                    inst.update_partials(o[None, :], 5)
            else:
                inst.update_partials(ops, 5)
            return [inst.get_partials(b) for b in (3, 4, 5, 6)] + [inst.get_scale_exponents(i) for i in (0, 1, 2, 5)]
        finally:
            inst.finalize()

    st1 = rng.integers(0, S + 1, size=P).astype(np.int32)
    st2 = rng.integers(0, S + 1, size=P).astype(np.int32)
    pa = rng.random((K, P, S)) * 0.9 + 0.05
    rng2 = np.random.default_rng(seed + 1)
    one = run(False)
    rng2 = np.random.default_rng(seed + 1)
    many = run(True)
    for a, b in zip(one, many):
        assert np.array_equal(a, b)


def check_sharded_instance(lib, oracle, div, monkeypatch, shards=3):
    """Site-pattern sharding inside one instance: MBAMD_SHARD=<g> (or a resource list with several GPUs)
    splits the patterns over g child engines -- here g children on the same device.  Every read-out equals the unsharded
    instance's: lnL to the last bit of the per-pattern values (the sum is formed per child), partial updates, rejects."""
    base = lk.BeagleDivision(div, lib)
    want = base.LogLike(0)
    want_sites = base.inst.get_site_log_likelihoods()
    base.finalize()
    monkeypatch.setenv("MBAMD_SHARD", str(shards))
    for scaling in (lk.MB_BEAGLE_SCALE_ALWAYS, lk.MB_BEAGLE_SCALE_DYNAMIC):
        bd = lk.BeagleDivision(div, lib, scaling=scaling)
        try:
            assert bd.inst.child_count() == min(shards, (div.npatterns + 63) // 64)
            got = bd.LogLike(0)
            sites = bd.inst.get_site_log_likelihoods()
            if scaling == lk.MB_BEAGLE_SCALE_ALWAYS:
                assert np.array_equal(sites, want_sites)
            assert abs(got - want) <= 1e-10 * abs(want)
            assert abs(sites.sum() * 0 + (sites * div.weights).sum() - got) <= 1e-9 * abs(got)
            bd.AcceptMove(0)
            t = div.tree
            node = t.int_down_pass[0]
            old = t.length[node]
            t.length[node] = old * 1.9
            bd.TouchBranch(0, node)
            moved = bd.LogLike(0)
            ref = oracle.tree_loglike(div, use_shortcuts=False)
            assert abs(moved - ref) / abs(ref) < REL_FP64
            t.length[node] = old
            bd.ResetFlips(0)
            assert abs(bd.LogLike(0) - got) <= 1e-12 * abs(got)
        finally:
            bd.finalize()
    monkeypatch.delenv("MBAMD_SHARD")


def check_multi_partition_instance(lib, oracle, div_a, div_b, double_precision=False):
    """BEAGLE v3 multi-partition mode (reference src/mbbeagle.c:1500-3010): ONE instance holds the patterns of two data
    divisions with their own eigen-systems, category rates, branch lengths and operation lists; per-partition
    log-likelihoods must equal two separate single-division instances (and the oracle)."""
    assert div_a.nstates == div_b.nstates and div_a.ncat == div_b.ncat and div_a.ntaxa == div_b.ntaxa
    S, K, N = div_a.nstates, div_a.ncat, div_a.ntaxa
    divs = [div_a, div_b]
    want = [engine_lnl(lib, d, double_precision=double_precision) for d in divs]
    Pa, Pb = div_a.npatterns, div_b.npatterns
    P = Pa + Pb
    nInt, nNodes = N - 2, 2 * N - 2
    # one chain; buffer indices as MrBayes lays them out: tips 0..N-1, interior N..; matrices / eigen offset by division
    inst = bg.BeagleInstance(lib, N, N + nInt, N, S, P, 2, 2 * nNodes, K, nInt + 2,
                             preference_flags=bg.BEAGLE_FLAG_PRECISION_DOUBLE if double_precision else bg.BEAGLE_FLAG_PRECISION_SINGLE)
    try:
        assert bool(inst.details.flags & bg.BEAGLE_FLAG_PRECISION_DOUBLE) == double_precision
        for t in range(N):
            inst.set_tip_states(t, np.concatenate([div_a.tip_states[t], div_b.tip_states[t]]).astype(np.int32))
        inst.set_pattern_weights(np.concatenate([div_a.weights, div_b.weights]))
        inst.set_pattern_partitions(2, np.concatenate([np.zeros(Pa, dtype=np.int32), np.ones(Pb, dtype=np.int32)]))
        assert inst.child_count() == 2
        ops = []
        eig_idx, rate_idx, prob_idx, lengths = [], [], [], []
        for d, dv in enumerate(divs):
            es = dv.eigen[0]
            inst.set_eigen_decomposition(d, es.evec, es.ivec, es.eval)
            inst.set_state_frequencies(d, dv.pi)
            inst.set_category_weights(d, dv.category_weights(0))
            inst.set_category_rates_with_index(d, dv.cat_rates)
            tr = dv.tree
            for p in tr.all_down_pass:
                if p == tr.root:
                    continue
                eig_idx.append(d); rate_idx.append(d); prob_idx.append(d * nNodes + p)
                lengths.append(min(max(tr.length[p], lk.BRLENS_MIN), lk.BRLENS_MAX))
            for p in tr.int_down_pass:
                l, r = tr.left[p], tr.right[p]
                ops.append([p, p - N, -1, l, d * nNodes + l, r, d * nNodes + r, d, nInt + d])
        inst.update_transition_matrices_with_multiple_models(eig_idx, rate_idx, prob_idx, lengths)
        for d in range(2):
            inst.reset_scale_factors_by_partition(nInt + d, d)
        # the two divisions' operations interleaved, as MrBayes' operationsAll is (src/mbbeagle.c:2262-2290)
        ops_a, ops_b = ops[:nInt], ops[nInt:]
        mixed = [x for pair in zip(ops_a, ops_b) for x in pair]
        inst.update_partials_by_partition(np.array(mixed, dtype=np.int32))
        parents = [divs[d].tree.root_left for d in range(2)]
        children = [divs[d].tree.root for d in range(2)]
        probs = [d * nNodes + divs[d].tree.root_left for d in range(2)]
        rc, by, total = inst.calculate_edge_log_likelihoods_by_partition(parents, children, probs, [0, 1], [0, 1],
                                                                         [nInt, nInt + 1], [0, 1], 1)
        assert rc == 0
        for d in range(2):
            assert abs(by[d] - want[d]) <= 1e-10 * abs(want[d]), (d, by[d], want[d])
            ref = oracle.tree_loglike(divs[d], use_shortcuts=False)
            assert abs(by[d] - ref) / abs(ref) < REL_FP64
        assert abs(total - sum(want)) <= 1e-10 * abs(total)
        sites = inst.get_site_log_likelihoods()
        assert abs((sites[:Pa] * div_a.weights).sum() - want[0]) <= 1e-9 * abs(want[0])
        # only partition 1 re-evaluated (its branch lengths doubled): partition 0's value must not move
        only = [1]
        rc, by1, _ = inst.calculate_edge_log_likelihoods_by_partition([parents[1]], [children[1]], [probs[1]], [1], [1], [nInt + 1], only, 1)
        assert abs(by1[0] - want[1]) <= 1e-10 * abs(want[1])
    finally:
        inst.finalize()


def engine_lnl(lib, div, scaling=lk.MB_BEAGLE_SCALE_ALWAYS, nchains=1, chain=0, double_precision=False):
    bd = lk.BeagleDivision(div, lib, nchains=nchains, scaling=scaling, double_precision=double_precision)
    try:
        return bd.LogLike(chain)
    finally:
        bd.finalize()


def check_golden_case(lib, oracle, golden_dir, case, scaling=lk.MB_BEAGLE_SCALE_ALWAYS):
    g = gold(golden_dir, case)
    div = division_from_golden(golden_dir, case)
    lnl = engine_lnl(lib, div, scaling)
    ref64, reffma = g["lnL"]["fp64"], g["lnL"]["fma"]
    assert abs(lnl - ref64) / abs(ref64) < REL_FP64, (case, lnl, ref64)
    assert abs(lnl - reffma) / abs(reffma) < REL_FMA, (case, lnl, reffma)
    want = oracle.tree_loglike(div, use_shortcuts=False)
    assert abs(lnl - want) / abs(want) < REL_FP64, (case, lnl, want)
    # and it must sit inside (twice) the reference's own fp32-vs-fp64 band on this data set
    band = max(abs(reffma - ref64), abs(g["lnL"]["scalar"] - ref64))
    assert abs(lnl - ref64) <= 2.0 * band + 1e-3, (case, lnl, ref64, band)
    return lnl


def check_generic_states(lib, oracle, nstates, ntaxa, npat, ncat=4, p_gap=0.05):
    """A full tree of a random reversible model with `nstates` states (restriction sites 2, covarion nucleotides 8, ...):
    engine == oracle for both scaling schemes, per site too, and a partial update + reject behaves."""
    from mrbayes_amd.division import synthetic_division
    div = synthetic_division("gen%d" % nstates, ntaxa, npat, seed=11, tree_seed=5, alpha=0.7, ncat=ncat, p_gap=p_gap)
    want = oracle.tree_loglike(div, use_shortcuts=False)
    for scaling in (lk.MB_BEAGLE_SCALE_ALWAYS, lk.MB_BEAGLE_SCALE_DYNAMIC):
        lnl = engine_lnl(lib, div, scaling)
        assert abs(lnl - want) / abs(want) < REL_FP64, (nstates, scaling, lnl, want)
    check_site_likelihoods(lib, oracle, div)
    check_partial_update_and_reject(lib, oracle, div, lk.MB_BEAGLE_SCALE_DYNAMIC)


def check_site_likelihoods(lib, oracle, div):
    bd = lk.BeagleDivision(div, lib)
    try:
        lnl = bd.LogLike(0)
        site = bd.inst.get_site_log_likelihoods()
        want_lnl, want_site = oracle.tree_loglike(div, use_shortcuts=False, want_sites=True)
        if div.pinvar == 0.0:
            assert np.allclose(site, want_site, rtol=2e-6, atol=2e-5)
            assert abs(float((site * div.weights).sum()) - lnl) < 1e-6 * abs(lnl)
        assert abs(lnl - want_lnl) / abs(want_lnl) < REL_FP64
        # a client that reads the site values gets them through pinned host memory from the second evaluation on
        # (beagleGetSiteLogLikelihoods): same numbers either way, and they follow the state
        bd.AcceptMove(0)
        bd.TouchAllTreeNodes(0)
        assert bd.LogLike(0) == lnl
        assert np.array_equal(bd.inst.get_site_log_likelihoods(), site)
        bd.AcceptMove(0)
        t = div.tree
        deep = max(range(t.ntaxa), key=lambda i: _depth(t, i))
        old = t.length[deep]
        t.length[deep] = old * 2.5
        bd.TouchBranch(0, deep)
        lnl2 = bd.LogLike(0)
        site2 = bd.inst.get_site_log_likelihoods()
        t.length[deep] = old
        assert lnl2 != lnl and not np.array_equal(site2, site)
        if div.pinvar == 0.0:
            assert abs(float((site2 * div.weights).sum()) - lnl2) < 1e-6 * abs(lnl2)
    finally:
        bd.finalize()


def check_partial_update_and_reject(lib, oracle, div, scaling=lk.MB_BEAGLE_SCALE_ALWAYS):
    """An MCMC-style sequence on one chain: full evaluation; change one branch (only the path to the
    root is recomputed); reject (flips undone) ; change another branch and accept; every value must
    equal a from-scratch evaluation of the same state."""
    import copy
    t = div.tree
    bd = lk.BeagleDivision(div, lib, scaling=scaling)
    try:
        lnl0 = bd.LogLike(0)
        bd.AcceptMove(0)
        assert abs(lnl0 - oracle.tree_loglike(div, use_shortcuts=False)) / abs(lnl0) < REL_FP64
        deep = max(range(t.ntaxa), key=lambda i: _depth(t, i))
        old = t.length[deep]
        t.length[deep] = old * 3.7
        bd.TouchBranch(0, deep)
        lnl1 = bd.LogLike(0)
        want1 = oracle.tree_loglike(div, use_shortcuts=False)
        assert abs(lnl1 - want1) / abs(want1) < REL_FP64, (lnl1, want1)
        assert abs(lnl1 - lnl0) > 1e-6 * abs(lnl0)
        # reject: restore the branch, undo the flips; a no-op evaluation must give lnl0 again
        t.length[deep] = old
        bd.ResetFlips(0)
        lnl_back = bd.LogLike(0)
        assert abs(lnl_back - lnl0) <= 1e-9 * abs(lnl0), (lnl_back, lnl0)
        bd.AcceptMove(0)
        # second move on an interior branch, accepted
        inner = t.int_down_pass[0]
        old2 = t.length[inner]
        t.length[inner] = old2 * 0.3 + 0.01
        bd.TouchBranch(0, inner)
        lnl2 = bd.LogLike(0)
        want2 = oracle.tree_loglike(div, use_shortcuts=False)
        assert abs(lnl2 - want2) / abs(want2) < REL_FP64, (lnl2, want2)
        bd.AcceptMove(0)
        t.length[inner] = old2
    finally:
        bd.finalize()


def check_general_state_path_kernel(lib, oracle, golden_dir, monkeypatch, cases=("avian_wag_g4", "synth_codon_m3")):
    """k_pathg (a move's root-ward path at 20 / 61 states on two waves: the sibling factors formed ahead of the chain) against
    k_walkg walking the same lists (MBAMD_NO_PATHG=1): a sequence of branch moves with accepts and rejects under both scaling
    schemes -- log-likelihoods and per-site values bit for bit, and the oracle's value for every state."""
    import copy
    for case in cases:
        for scaling in (lk.MB_BEAGLE_SCALE_DYNAMIC, lk.MB_BEAGLE_SCALE_ALWAYS):
            runs = []
            for no_path in (False, True):
                if no_path:
                    monkeypatch.setenv("MBAMD_NO_PATHG", "1")
                else:
                    monkeypatch.delenv("MBAMD_NO_PATHG", raising=False)
                div = division_from_golden(golden_dir, case)
                t = div.tree
                bd = lk.BeagleDivision(div, lib, scaling=scaling)
                try:
                    seq = [bd.LogLike(0)]
                    bd.AcceptMove(0)
                    rng = np.random.default_rng(17)
                    nodes = [i for i in range(len(t.anc)) if t.anc[i] != -1 and i != t.root]
                    for rep in range(7):
                        b = int(rng.choice(nodes))
                        old = t.length[b]
                        t.length[b] = old * float(np.exp(0.8 * (rng.random() - 0.5)))
                        bd.TouchBranch(0, b)
                        lnl = bd.LogLike(0)
                        site = bd.inst.get_site_log_likelihoods()
                        if not no_path and rep < 3:
                            want = oracle.tree_loglike(div, use_shortcuts=False)
                            assert abs(lnl - want) / abs(want) < REL_FP64, (case, scaling, rep, lnl, want)
                        seq.append(lnl)
                        seq.append(site.copy())
                        if rep % 3 == 1:                    # reject: the branch back, the flips undone
                            t.length[b] = old
                            bd.ResetFlips(0)
                            seq.append(bd.LogLike(0))
                        bd.AcceptMove(0)
                finally:
                    bd.finalize()
                runs.append(seq)
            monkeypatch.delenv("MBAMD_NO_PATHG", raising=False)
            assert len(runs[0]) == len(runs[1])
            for x, y in zip(runs[0], runs[1]):
                if isinstance(x, np.ndarray):
                    assert np.array_equal(x, y), (case, scaling)
                else:
                    assert x == y, (case, scaling, x, y)


def check_path_kernels_at_bench_shape(lib, golden_dir, monkeypatch, case, switch):
    """The partial-update kernels (k_path4: MBAMD_NO_PATH4; k_pathg: MBAMD_NO_PATHG) at the BASELINE shape `case` -- hundreds of
    tiles x categories over every XCD, paths of 20-30 operations (k_path4's 24-operation chunk boundary, k_pathg's four-deep factor
    ring wrapping several times).  Three engines step through the same branch moves (deep tips, branches next to the root, random
    ones; accepts and rejects mixed): the path kernel, the whole-tree walk kernel on the same lists (`switch` set when the instance is
    made), and the double-precision engine (itself pinned to the reference's fp64 build).  After every move: log-likelihood and
    per-site values of the first two bit for bit, and the log-likelihood within REL_FP64 of the fp64 engine's.  Both scaling schemes."""
    div = division_from_golden(golden_dir, case)
    t = div.tree
    base_len = list(t.length)
    movable = [i for i in range(len(t.anc)) if t.anc[i] != -1 and i != t.root]
    by_depth = sorted(movable, key=lambda i: _depth(t, i))
    rng = np.random.default_rng(23)
    moves = [by_depth[-1], by_depth[0], by_depth[-2], by_depth[1]] + [int(x) for x in rng.choice(movable, 4, replace=False)]
    assert _depth(t, moves[0]) >= 12 and _depth(t, moves[1]) <= 1, (case, _depth(t, moves[0]), _depth(t, moves[1]))
    with open(os.path.join(golden_dir, case + ".json")) as fh:
        gold = json.load(fh)["lnL"]
    for scaling in (lk.MB_BEAGLE_SCALE_DYNAMIC, lk.MB_BEAGLE_SCALE_ALWAYS):
        t.length[:] = base_len
        monkeypatch.delenv(switch, raising=False)
        path = lk.BeagleDivision(div, lib, scaling=scaling)
        monkeypatch.setenv(switch, "1")
        walk = lk.BeagleDivision(div, lib, scaling=scaling)
        monkeypatch.delenv(switch, raising=False)
        f64 = lk.BeagleDivision(div, lib, scaling=scaling, double_precision=True)
        engines = (path, walk, f64)
        try:
            start = [bd.LogLike(0) for bd in engines]
            for bd in engines:
                bd.AcceptMove(0)
            assert start[0] == start[1] and abs(start[0] - gold["fp64"]) <= REL_FP64 * abs(gold["fp64"])
            assert abs(start[2] - gold["fp64"]) <= 1e-9 * abs(gold["fp64"]), (case, start[2], gold["fp64"])
            last = start[0]
            for rep, b in enumerate(moves):
                old = t.length[b]
                t.length[b] = old * (2.3 if rep % 2 == 0 else 0.45)
                vals = []
                for bd in engines:
                    bd.TouchBranch(0, b)
                    vals.append(bd.LogLike(0))
                sites = [bd.inst.get_site_log_likelihoods() for bd in engines[:2]]
                assert vals[0] == vals[1], (case, scaling, rep, b, vals)
                assert np.array_equal(sites[0], sites[1]), (case, scaling, rep, b)
                assert abs(vals[0] - vals[2]) <= REL_FP64 * abs(vals[2]), (case, scaling, rep, b, vals)
                assert vals[0] != last
                if rep in (1, 4, 6):                        # reject: the branch back, the flips undone, the old value again
                    t.length[b] = old
                    back = []
                    for bd in engines:
                        bd.ResetFlips(0)
                        back.append(bd.LogLike(0))
                    assert back[0] == last and back[1] == last, (case, scaling, rep, back, last)
                else:
                    last = vals[0]
                for bd in engines:
                    bd.AcceptMove(0)
        finally:
            for bd in engines:
                bd.finalize()
            t.length[:] = base_len


def check_fused_path_and_likelihood(lib, oracle, monkeypatch, golden_dir=None, case=None, ntaxa=60, npat=700):
    """A 4-state branch move: the root-ward path and the log-likelihood over its last result run as ONE launch (k_path4_lnl: the path is
    held until the next call).  Against the same moves with MBAMD_NO_FUSE_PATH=1 (k_path4, then k_integrate_lnl_s4): log-likelihoods
    and per-site values bit for bit, through accepts and rejects, under both scaling schemes, with a call between the list and its
    log-likelihood that forces the path out (a partials read-back) and one that does not (new category weights)."""
    for scaling in (lk.MB_BEAGLE_SCALE_DYNAMIC, lk.MB_BEAGLE_SCALE_ALWAYS):
        runs = []
        for off in (False, True):
            if off:
                monkeypatch.setenv("MBAMD_NO_FUSE_PATH", "1")
            else:
                monkeypatch.delenv("MBAMD_NO_FUSE_PATH", raising=False)
            div = division_from_golden(golden_dir, case) if case else synthetic_division("gtr", ntaxa, npat, seed=41, tree_seed=42, p_gap=0.03)
            t = div.tree
            bd = lk.BeagleDivision(div, lib, scaling=scaling)
            try:
                seq = [bd.LogLike(0)]
                bd.AcceptMove(0)
                if not off and oracle is not None:
                    want = oracle.tree_loglike(div, use_shortcuts=False)
                    assert abs(seq[0] - want) / abs(want) < REL_FP64
                rng = np.random.default_rng(13)
                nodes = [i for i in range(len(t.anc)) if t.anc[i] != -1 and i != t.root]
                for rep in range(9):
                    b = int(rng.choice(nodes))
                    old = t.length[b]
                    t.length[b] = old * (1.8 if rep % 2 else 0.55)
                    bd.TouchBranch(0, b)
                    lnl = bd.LogLike(0)
                    seq.append(lnl)
                    seq.append(bd.inst.get_site_log_likelihoods().copy())
                    if not off and oracle is not None and rep < 3:
                        want = oracle.tree_loglike(div, use_shortcuts=False)
                        assert abs(lnl - want) / abs(want) < REL_FP64, (scaling, rep, lnl, want)
                    if rep % 3 == 1:
                        t.length[b] = old
                        bd.ResetFlips(0)
                        seq.append(bd.LogLike(0))
                    bd.AcceptMove(0)
                seq.append(bd.LogLike(0))               # nothing touched: no list, the plain integration
            finally:
                bd.finalize()
            runs.append(seq)
        monkeypatch.delenv("MBAMD_NO_FUSE_PATH", raising=False)
        assert len(runs[0]) == len(runs[1])
        for x, y in zip(runs[0], runs[1]):
            if isinstance(x, np.ndarray):
                assert np.array_equal(x, y), scaling
            else:
                assert x == y, (scaling, x, y)


def check_forked_paths(lib, oracle, monkeypatch, golden_dir=None, case=None, ntaxa=60, npat=700, ncat=4):
    """The lists of topology moves -- two root-ward paths that join (NNI / SPR / TBR leave two dirty branches) -- run on the path kernel
    as ARMS (round 6; a third dirty branch whose arm would need a second saved result goes to the tree walk as before).  Against the same
    moves with MBAMD_NO_FORK_PATH=1 (the tree-walk kernel on the same lists): log-likelihoods and per-site values bit for bit through
    accepts and rejects under both scaling schemes, the oracle / the fp64 engine within REL_FP64; the engine's list counters say that
    the forked programs ran (and ran together with their log-likelihood as one launch)."""
    for scaling in (lk.MB_BEAGLE_SCALE_DYNAMIC, lk.MB_BEAGLE_SCALE_ALWAYS):
        runs, launches = [], []
        for off in (False, True):
            if off:
                monkeypatch.setenv("MBAMD_NO_FORK_PATH", "1")
            else:
                monkeypatch.delenv("MBAMD_NO_FORK_PATH", raising=False)
            div = division_from_golden(golden_dir, case) if case else synthetic_division("gtr", ntaxa, npat, seed=43, tree_seed=44, p_gap=0.03, ncat=ncat)
            t = div.tree
            bd = lk.BeagleDivision(div, lib, scaling=scaling)
            f64 = lk.BeagleDivision(div, lib, scaling=scaling, double_precision=True) if (case and not off) else None
            try:
                seq = [bd.LogLike(0)]
                bd.AcceptMove(0)
                if f64 is not None:
                    f64.LogLike(0)
                    f64.AcceptMove(0)
                rng = np.random.default_rng(17)
                nodes = [i for i in range(len(t.anc)) if t.anc[i] != -1 and i != t.root]
                by_depth = sorted(nodes, key=lambda i: _depth(t, i))
                sets = [(by_depth[-1], by_depth[0]), (by_depth[-1], by_depth[-3]), (by_depth[1], by_depth[0])]
                sets += [tuple(int(x) for x in rng.choice(nodes, 2, replace=False)) for _ in range(5)]
                sets += [tuple(int(x) for x in rng.choice(nodes, 3, replace=False)) for _ in range(2)]
                sets.append((by_depth[-1], t.anc[by_depth[-1]]))          # a branch and the one above it: one path
                for rep, bs in enumerate(sets):
                    old = [t.length[b] for b in bs]
                    for q, b in enumerate(bs):
                        t.length[b] = old[q] * (1.7 if (rep + q) % 2 else 0.6)
                        bd.TouchBranch(0, b)
                    lnl = bd.LogLike(0)
                    seq.append(lnl)
                    seq.append(bd.inst.get_site_log_likelihoods().copy())
                    if not off and oracle is not None and rep < 4:
                        want = oracle.tree_loglike(div, use_shortcuts=False)
                        assert abs(lnl - want) / abs(want) < REL_FP64, (scaling, rep, lnl, want)
                    if f64 is not None:
                        for b in bs:
                            f64.TouchBranch(0, b)
                        want = f64.LogLike(0)
                        assert abs(lnl - want) <= REL_FP64 * abs(want), (scaling, rep, lnl, want)
                    if rep % 3 == 1:
                        for q, b in enumerate(bs):
                            t.length[b] = old[q]
                        bd.ResetFlips(0)
                        seq.append(bd.LogLike(0))
                        if f64 is not None:
                            f64.ResetFlips(0)
                            f64.LogLike(0)
                    bd.AcceptMove(0)
                    if f64 is not None:
                        f64.AcceptMove(0)
                launches.append(bd.inst.get_list_counts())
            finally:
                bd.finalize()
                if f64 is not None:
                    f64.finalize()
            runs.append(seq)
        monkeypatch.delenv("MBAMD_NO_FORK_PATH", raising=False)
        assert len(runs[0]) == len(runs[1])
        for x, y in zip(runs[0], runs[1]):
            if isinstance(x, np.ndarray):
                assert np.array_equal(x, y), scaling
            else:
                assert x == y, (scaling, x, y)
        on, offc = launches
        assert on[2] >= 4 and offc[2] == 0 and offc[4] >= on[4] + on[2], launches      # forked paths ran; without: walks
        assert on[3] >= on[2] or ncat > 8, launches                                    # ... together with their log-likelihood (up to eight categories)


def _depth(t, i):
    d = 0
    while t.anc[i] != -1 and t.anc[i] != t.root:
        i = t.anc[i]
        d += 1
    return d


def check_multi_chain(lib, oracle, div, nchains=3):
    """All local chains live in one instance (as in MrBayes); their buffers must not interfere."""
    bd = lk.BeagleDivision(div, lib, nchains=nchains)
    try:
        want = oracle.tree_loglike(div, use_shortcuts=False)
        vals = [bd.LogLike(c) for c in range(nchains)]
        for v in vals:
            assert abs(v - want) / abs(want) < REL_FP64
        assert max(vals) - min(vals) <= 1e-9 * abs(want)
        # re-evaluating chain 0 after the others ran still gives the same number
        bd.TouchAllTreeNodes(0)
        assert abs(bd.LogLike(0) - vals[0]) <= 1e-9 * abs(want)
    finally:
        bd.finalize()


def check_dynamic_rescaling_state_machine(lib, oracle, div):
    """A tree deep enough to underflow fp32 without scaling: the first (no-rescale) pass must report
    BEAGLE_ERROR_FLOATING_POINT and the rescale-all retry must land on the right value."""
    bd = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
    try:
        bd.UpDateCijk(0)
        bd.TreeTiProbs_Beagle(0)
        bd.TreeCondLikes_Beagle_No_Rescale(0)
        rc, lnl = bd.TreeLikelihood_Beagle(0)
        assert rc == bg.BEAGLE_ERROR_FLOATING_POINT, (rc, lnl)
    finally:
        bd.finalize()
    lnl = engine_lnl(lib, div, scaling=lk.MB_BEAGLE_SCALE_DYNAMIC)
    want = oracle.tree_loglike(div, use_shortcuts=False)
    assert abs(lnl - want) / abs(want) < REL_FP64, (lnl, want)


# ---- Fitch parsimony on the device -----------------------------------------------------------
def check_parsimony(lib, ntaxa, npat, nstates, seed=3, words=1, gaps=0.05):
    """Down-pass, final pass and candidate lengths of a random tree against the oracle's restatement of GetFitchPartials /
    GetParsFP / the ParsSPR1 candidate loops -- sets word for word, lengths exactly (integer-valued weights)."""
    from mrbayes_amd import parsimony as mp
    from mrbayes_amd import tree as mbtree
    from tests import oracle_lib as ol
    rng = np.random.default_rng(seed)
    t = mbtree.random_tree(ntaxa, seed)
    nsets = t.n_nodes
    # tips: a random walk over the tree would be more realistic; independent states stress the empty-intersection branch
    base = rng.integers(0, nstates, size=npat)
    states = np.where(rng.random((ntaxa, npat)) < 0.3, rng.integers(0, nstates, size=(ntaxa, npat)), base[None, :])
    sets = np.zeros((nsets, npat * words), dtype=np.uint64)
    for i in range(ntaxa):
        for v in range(words):
            lo, hi = 64 * v, min(nstates, 64 * (v + 1))
            if hi <= lo:
                continue
            s = states[i]
            word = np.where((s >= lo) & (s < hi), np.left_shift(np.uint64(1), np.clip(s - lo, 0, 63).astype(np.uint64)), np.uint64(0))
            amb = rng.random(npat) < gaps
            full = np.uint64((1 << (hi - lo)) - 1) if hi - lo < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
            sets[i, v::words] = np.where(amb, full, word)
        # a partially ambiguous tip set now and then
        k = rng.integers(0, npat)
        sets[i, k * words] |= np.uint64(1) << np.uint64(rng.integers(0, min(nstates, 64)))
    # interior sets start as whatever an earlier move left: random subsets, uploaded like everything else
    for i in range(ntaxa, nsets):
        sets[i, :] = rng.integers(0, 1 << min(nstates, 62), size=npat * words, dtype=np.uint64) if words == 1 else 0
    w = rng.integers(1, 9, size=npat).astype(np.float32)
    inst = mp.ParsimonyInstance(nsets, npat, nstates if words == 1 else 64 * words, words, lib=lib)
    try:
        for i in range(nsets):
            inst.set_sets(i, sets[i])
        inst.set_pattern_weights(w)
        np.testing.assert_array_equal(inst.all_sets(), sets)
        ref = sets.copy()
        # GetParsDP(t, root->left) + GetParsFP(t, root->left)
        dops = mp.down_pass_ops(t, t.root_left)
        assert len(dops) == t.n_int_nodes
        total, node_len = ol.pars_down(ref, dops, w)
        got = inst.down_pass(dops)
        assert got == total, (got, total)
        np.testing.assert_array_equal(inst.all_sets(), ref)
        assert mp.GetParsimonyLength(inst, t) == total + ol.pars_score(ref, [[t.root_left, -1, t.root, -1]], w)[0]
        # per-node lengths as Likelihood_Pars keeps them: score tuples over the children
        np.testing.assert_array_equal(inst.score([[o[1], -1, o[2], -1] for o in dops]), node_len)
        fops = mp.final_pass_ops(t, t.root_left)
        ol.pars_final(ref, fops, npat)
        inst.final_pass(fops)
        np.testing.assert_array_equal(inst.all_sets(), ref)
        # a clipped subtree: down-pass without waiting for the length, final pass against a stale ancestor set
        v = next(n for n in t.int_down_pass if n != t.root_left)
        sub = mp.down_pass_ops(t, v)
        ol.pars_down(ref, sub, w)
        assert inst.down_pass(sub, want_length=False) is None
        subf = mp.final_pass_ops(t, v)
        ol.pars_final(ref, subf, npat)
        inst.final_pass(subf)
        np.testing.assert_array_equal(inst.all_sets(), ref)
        # a PARTIAL down pass (a big subtree: cut into bins over the waves of a workgroup) queued in front of a FULL final pass: steps
        # of the final pass that hang below that subtree's bins must not be moved ahead of the steps that make their ancestors' sets
        big = max((n for n in t.int_down_pass if n != t.root_left), key=lambda n: len(mp.down_pass_ops(t, n)))
        part = mp.down_pass_ops(t, big)
        ol.pars_down(ref, part, w)
        assert inst.down_pass(part, want_length=False) is None
        ol.pars_final(ref, fops, npat)
        inst.final_pass(fops)
        np.testing.assert_array_equal(inst.all_sets(), ref)
        # candidate positions: the three shapes ParsSPR1 uses and the four-set shape of ParsTBR1
        nodes = [n for n in t.all_down_pass if t.anc[n] >= 0]
        tuples = []
        for n in nodes[:40]:
            tuples += [[n, t.anc[n], v, -1], [v, -1, n, t.anc[n]], [n, t.anc[n], nodes[(n * 7) % len(nodes)], t.anc[nodes[(n * 7) % len(nodes)]]]]
        np.testing.assert_array_equal(inst.score(tuples), ol.pars_score(ref, tuples, w))
        # errors: out-of-range indices, bits beyond the declared width
        with pytest.raises(bg.BeagleError):
            inst.down_pass([[nsets, 0, 1, -1]])
        with pytest.raises(bg.BeagleError):
            inst.final_pass([[ntaxa, 0, 1, -1]])
        if words == 1 and nstates < 64:
            with pytest.raises(bg.BeagleError):
                inst.set_sets(0, np.full(npat, 1 << nstates, dtype=np.uint64))
    finally:
        inst.finalize()


# ---- double precision (BEAGLE_FLAG_PRECISION_DOUBLE; `set beagleprecision=double`) ---------------------------------
ABS_F64 = 2e-6          # the reference prints its log-likelihood with six (known-answer files: eight) decimals
REL_F64 = 2e-9          # how closely the model parameters themselves are reproduced outside the reference (its discrete-gamma
                        # quantiles and eigen-solver iterate to ~1e-6 .. 1e-9: seen as 7e-10 on the WAG case); the fp32 engine sits
                        # at 1e-7 .. 1e-6 on the same cases


def check_double_precision(lib, golden_dir, case):
    """The fp64 engine against the reference's double build (oracle/_ref/mb_fp64: CLFlt = double, src/bayes.h:110-112):
    log-likelihood to the printed digits for both scaling schemes, site values consistent with the sum, a partial update equal
    to a fresh evaluation, a reject restoring the value, several chains independent."""
    g = gold(golden_dir, case)
    ref64 = g["lnL"]["fp64"]
    div = division_from_golden(golden_dir, case)
    single = engine_lnl(lib, div)
    vals = []
    for scaling in (lk.MB_BEAGLE_SCALE_ALWAYS, lk.MB_BEAGLE_SCALE_DYNAMIC):
        bd = lk.BeagleDivision(div, lib, scaling=scaling, double_precision=True, nchains=2)
        try:
            assert bd.inst.details.flags & bg.BEAGLE_FLAG_PRECISION_DOUBLE and not bd.inst.details.flags & bg.BEAGLE_FLAG_PRECISION_SINGLE
            assert b"double-precision" in bd.inst.details.implName
            lnl = bd.LogLike(0)
            bd.AcceptMove(0)
            assert abs(lnl - ref64) <= ABS_F64 + REL_F64 * abs(ref64), (case, scaling, lnl, ref64)
            assert abs(lnl - ref64) <= abs(single - ref64) + ABS_F64, (case, lnl, single, ref64)
            vals.append(lnl)
            if div.pinvar == 0.0:                     # (with +I the host mixes the invariable-site term into the sum, src/mbbeagle.c:1322-1358)
                site = bd.inst.get_site_log_likelihoods()
                assert abs(float(np.dot(site, div.weights)) - lnl) <= 1e-9 * abs(lnl)
            assert abs(bd.LogLike(1) - lnl) <= 1e-12 * abs(lnl)            # the second chain: its own buffers, the same state
            bd.AcceptMove(1)
            # a partial update == a fresh evaluation of the new state; a reject restores the old value exactly
            t = div.tree
            deep = max(range(t.ntaxa), key=lambda i: _depth(t, i))
            old = t.length[deep]
            t.length[deep] = old * 2.5
            try:
                bd.TouchBranch(0, deep)
                moved = bd.LogLike(0)
                fresh = lk.BeagleDivision(div, lib, scaling=scaling, double_precision=True)
                try:
                    want = fresh.LogLike(0)
                finally:
                    fresh.finalize()
                assert abs(moved - want) <= 1e-12 * abs(want), (moved, want)
                assert abs(moved - lnl) > 1e-9 * abs(lnl)
            finally:
                t.length[deep] = old
            bd.ResetFlips(0)
            assert bd.LogLike(0) == lnl
            bd.AcceptMove(0)
        finally:
            bd.finalize()
    assert abs(vals[0] - vals[1]) <= 1e-11 * abs(vals[0]), vals


def check_double_precision_walk(lib, golden_dir, case, monkeypatch):
    """Four states in fp64: the tree walk (one launch per list) gives the bits of the level kernels (one launch per dependency
    level, MBAMD_F64_NO_WALK=1), for a full evaluation, a partial update and both scaling schemes -- per site, not just the sum."""
    div = division_from_golden(golden_dir, case)
    t = div.tree
    deep = max(range(t.ntaxa), key=lambda i: _depth(t, i))
    for scaling in (lk.MB_BEAGLE_SCALE_ALWAYS, lk.MB_BEAGLE_SCALE_DYNAMIC):
        got = {}
        for walk in (True, False):
            if walk:
                monkeypatch.delenv("MBAMD_F64_NO_WALK", raising=False)
                monkeypatch.setenv("MBAMD_F64_WALK_ALWAYS", "1")       # (mid-sized full evaluations default to the levels)
            else:
                monkeypatch.setenv("MBAMD_F64_NO_WALK", "1")
            bd = lk.BeagleDivision(div, lib, scaling=scaling, double_precision=True)
            try:
                bd.inst.get_kernel_timing(reset=True)
                lnl = bd.LogLike(0)
                _, launches = bd.inst.get_kernel_timing(reset=True)
                site = bd.inst.get_site_log_likelihoods().copy()
                bd.AcceptMove(0)
                old = t.length[deep]
                t.length[deep] = old * 1.7
                try:
                    bd.TouchBranch(0, deep)
                    moved = bd.LogLike(0)
                    site2 = bd.inst.get_site_log_likelihoods().copy()
                finally:
                    t.length[deep] = old
                got[walk] = (lnl, site, moved, site2, launches)
            finally:
                bd.finalize()
        monkeypatch.delenv("MBAMD_F64_NO_WALK", raising=False)
        monkeypatch.delenv("MBAMD_F64_WALK_ALWAYS", raising=False)
        assert got[True][0] == got[False][0] and got[True][2] == got[False][2], (case, scaling, got[True][0], got[False][0])
        assert np.array_equal(got[True][1], got[False][1]) and np.array_equal(got[True][3], got[False][3])
        assert got[True][4] <= 2 and got[False][4] > 2 * got[True][4], (got[True][4], got[False][4])   # a launch per list against one per level


F64_GENERAL_SWITCHES = ("MBAMD_F64_MFMA_NO_LDS", "MBAMD_F64_NO_TIPS_KERNEL", "MBAMD_F64_NO_CHAIN", "MBAMD_F64_NO_MATRIX_QUEUE", "MBAMD_F64_NO_RING")


def check_double_precision_general_paths(lib, golden_dir, case, monkeypatch):
    """16 ... 64 states in fp64: the kernels the engine picks by default -- operations on two tips from matrices parked in LDS, the
    contraction with its matrices in LDS (four or eight waves per workgroup), matrix updates of several calls as one launch, lists
    staged through the ring, a root-ward path as one launch with the running result in registers -- give the bits of the plain one-wave level
    kernel with everything switched off: per site, for a full evaluation and a partial update, both scaling schemes."""
    div = division_from_golden(golden_dir, case)
    t = div.tree
    deep = max(range(t.ntaxa), key=lambda i: _depth(t, i))
    for scaling in (lk.MB_BEAGLE_SCALE_ALWAYS, lk.MB_BEAGLE_SCALE_DYNAMIC):
        got = {}
        for plain in (False, True):
            for name in F64_GENERAL_SWITCHES:
                if plain:
                    monkeypatch.setenv(name, "1")
                else:
                    monkeypatch.delenv(name, raising=False)
            bd = lk.BeagleDivision(div, lib, scaling=scaling, double_precision=True)
            try:
                bd.inst.get_kernel_timing(reset=True)
                lnl = bd.LogLike(0)
                _, launches = bd.inst.get_kernel_timing(reset=True)
                site = bd.inst.get_site_log_likelihoods().copy()
                bd.AcceptMove(0)
                old = t.length[deep]
                t.length[deep] = old * 1.7
                try:
                    bd.TouchBranch(0, deep)
                    bd.inst.get_kernel_timing(reset=True)
                    moved = bd.LogLike(0)
                    _, launches2 = bd.inst.get_kernel_timing(reset=True)
                    site2 = bd.inst.get_site_log_likelihoods().copy()
                finally:
                    t.length[deep] = old
                got[plain] = (lnl, site, moved, site2, launches, launches2)
            finally:
                bd.finalize()
        for name in F64_GENERAL_SWITCHES:
            monkeypatch.delenv(name, raising=False)
        assert got[True][0] == got[False][0] and got[True][2] == got[False][2], (case, scaling, got[True][0], got[False][0])
        assert np.array_equal(got[True][1], got[False][1]) and np.array_equal(got[True][3], got[False][3])
        assert got[False][4] == got[True][4], (got[False][4], got[True][4])      # (a whole tree: a launch per dependency level either way)
        # the partial update is a root-ward path: ONE launch of the chain kernel per eigen part's list run together, a launch per level without it
        assert got[False][5] == 1 and got[True][5] >= 2, (got[False][5], got[True][5])


def check_double_precision_walk_categories(lib, monkeypatch, ntips=24, npat=200, seed=11):
    """The fp64 four-state walk keeps a pattern's categories in the lanes of ONE wave (category count rounded up to a power of two,
    the surplus lanes repeating the last category): every category count 1 ... 8 must give the bits of the level kernels --
    partials, exponents and the cumulative exponents of a whole tree, with one child held in memory (more nodes than LDS slots)."""
    rng = np.random.default_rng(seed)
    S, P = 4, npat
    nodes = 2 * ntips - 2                                    # tips 0 .. ntips-1, interior ntips .. nodes-1 (a caterpillar + cherries)
    for K in (1, 2, 3, 4, 5, 8):
        states = [rng.integers(0, S + 1, size=P).astype(np.int32) for _ in range(ntips)]
        mats = []
        for m in range(nodes):
            ti = rng.random((K, S, S)) + 0.02
            mats.append(ti / ti.sum(axis=2, keepdims=True) * 1e-3)       # (small values: the exponents are not all zero)
        # a random binary tree: repeatedly join two roots
        roots = list(range(ntips))
        ops = []
        nxt = ntips
        order = rng.permutation(ntips).tolist()
        roots = order
        while len(roots) > 1 and nxt < nodes + 1:
            i, j = sorted(rng.choice(len(roots), size=2, replace=False).tolist())
            a, b = roots[i], roots[j]
            ops.append([nxt, nxt - ntips, -1, a, a, b, b])
            del roots[j]
            roots[i] = nxt
            nxt += 1
        ops = np.array(ops, dtype=np.int32)
        nint = len(ops)
        got = {}
        for walk in (True, False):
            if walk:
                monkeypatch.delenv("MBAMD_F64_NO_WALK", raising=False)
                monkeypatch.setenv("MBAMD_F64_WALK_ALWAYS", "1")
                monkeypatch.setenv("MBAMD_F64_WALK_SLOTS", "2")      # (children fall out of LDS: the memory path runs too)
            else:
                monkeypatch.setenv("MBAMD_F64_NO_WALK", "1")
            inst = bg.BeagleInstance(lib, ntips, nint, ntips, S, P, 1, nodes + 1, K, nint + 1, preference_flags=bg.BEAGLE_FLAG_PRECISION_DOUBLE)
            try:
                for m in range(nodes):
                    inst.set_transition_matrix(m, mats[m])
                for i in range(ntips):
                    inst.set_tip_states(i, states[i])
                inst.reset_scale_factors(nint)
                inst.update_partials(ops, nint)
                got[walk] = [inst.get_partials(int(o[0])) for o in ops] + [inst.get_scale_exponents(i) for i in range(nint + 1)]
            finally:
                inst.finalize()
        for k in ("MBAMD_F64_NO_WALK", "MBAMD_F64_WALK_ALWAYS", "MBAMD_F64_WALK_SLOTS"):
            monkeypatch.delenv(k, raising=False)
        for x, y in zip(got[True], got[False]):
            assert np.array_equal(x, y), K
        assert any(np.any(e != 0) for e in got[True][nint:]), K


def check_double_precision_queue(lib, nstates=20, ncat=2, npat=150, seed=3):
    """The fp64 engine queues operation lists and runs them together (one launch per dependency level of ALL of them): a rejected
    list (bad index) is reported by the call that brought it and leaves nothing behind; lists split over several calls -- with a
    beagleRemoveScaleFactors on untouched buffers in between, as MrBayes does between the parts of a codon model -- give the bits
    of one call; reading a buffer runs what is queued."""
    rng = np.random.default_rng(seed)
    S, K, P = nstates, ncat, npat
    ntips, nint = 6, 5
    states = [rng.integers(0, S + 1, size=P).astype(np.int32) for _ in range(ntips)]
    mats = []
    for m in range(ntips + nint):
        ti = rng.random((K, S, S)) + 0.02
        mats.append(ti / ti.sum(axis=2, keepdims=True) * 1e-2)
    ops = np.array([[6, 0, -1, 0, 0, 1, 1], [7, 1, -1, 2, 2, 3, 3], [8, 2, -1, 4, 4, 5, 5], [9, 3, -1, 6, 6, 7, 7], [10, 4, -1, 9, 9, 8, 8]], dtype=np.int32)

    def make():
        inst = bg.BeagleInstance(lib, ntips, nint, ntips, S, P, 1, ntips + nint, K, nint + 3, preference_flags=bg.BEAGLE_FLAG_PRECISION_DOUBLE)
        for m in range(ntips + nint):
            inst.set_transition_matrix(m, mats[m])
        for i in range(ntips):
            inst.set_tip_states(i, states[i])
        return inst
    a = make()
    b = make()
    try:
        a.reset_scale_factors(5)
        a.update_partials(ops, 5)
        want = [a.get_partials(int(o[0])) for o in ops] + [a.get_scale_exponents(i) for i in range(6)]
        b.reset_scale_factors(5)
        b.reset_scale_factors(6)
        b.update_partials(ops[:2], 5)
        b.remove_scale_factors(np.array([7], dtype=np.int32), 6)        # buffers no queued operation touches: does not have to run the queue
        b.update_partials(ops[2:3], 5)
        bad = ops[3:].copy()
        bad[1, 3] = 99                                                     # child buffer out of range in the SECOND operation of the list
        with pytest.raises(bg.BeagleError):
            b.update_partials(bad, 5)
        b.update_partials(ops[3:], 5)                                      # (the rejected list left nothing queued: operation 9 is not done twice)
        got = [b.get_partials(int(o[0])) for o in ops] + [b.get_scale_exponents(i) for i in range(6)]
        for x, y in zip(want, got):
            assert np.array_equal(x, y)
        assert any(np.any(e != 0) for e in want[len(ops):])
    finally:
        a.finalize()
        b.finalize()


def check_double_precision_matrix_queue(lib, monkeypatch, nstates=20, ncat=2, seed=5):
    """The fp64 engine queues beagleUpdateTransitionMatrices calls and computes them in one launch when something else is asked of it:
    several calls (eigen parts, as a codon model makes them), a SECOND update of a matrix still in the queue (the later length wins),
    another category-rate vector between two calls, more calls than the staging ring holds (it wraps), and category weights / state
    frequencies set again to the same and to other values -- everything equal to the engine with the queue and the ring switched off."""
    rng = np.random.default_rng(seed)
    S, K, P = nstates, ncat, 70
    nmat, neig = 12, 2

    def system():
        a = rng.random((S, S))
        a = a + a.T
        np.fill_diagonal(a, 0.0)
        np.fill_diagonal(a, -a.sum(axis=1))
        lam, v = np.linalg.eigh(a)
        return v, v.T.copy(), lam
    systems = [system() for _ in range(neig)]
    states = [rng.integers(0, S + 1, size=P).astype(np.int32) for _ in range(2)]
    freqs = [np.full(S, 1.0 / S), rng.dirichlet(np.ones(S))]
    weights = [np.full(K, 1.0 / K), rng.dirichlet(np.ones(K))]

    def run(plain):
        for name in ("MBAMD_F64_NO_MATRIX_QUEUE", "MBAMD_F64_NO_RING"):
            if plain:
                monkeypatch.setenv(name, "1")
            else:
                monkeypatch.delenv(name, raising=False)
        inst = bg.BeagleInstance(lib, 2, 1, 2, S, P, neig, nmat, K, 2, preference_flags=bg.BEAGLE_FLAG_PRECISION_DOUBLE)
        out = []
        try:
            for e, (u, ui, lam) in enumerate(systems):
                inst.set_eigen_decomposition(e, u, ui, lam)
            for i in range(2):
                inst.set_tip_states(i, states[i])
            inst.set_category_rates(np.linspace(0.5, 1.5, K))
            inst.update_transition_matrices(0, np.arange(0, 6, dtype=np.int32), np.linspace(0.01, 0.3, 6))
            inst.update_transition_matrices(1, np.arange(6, 12, dtype=np.int32), np.linspace(0.02, 0.5, 6))
            inst.update_transition_matrices(1, np.array([3, 7], dtype=np.int32), np.array([0.9, 0.8]))       # 3 and 7 are still queued
            out += [inst.get_transition_matrix(m) for m in range(nmat)]
            inst.update_transition_matrices(0, np.array([0, 1], dtype=np.int32), np.array([0.11, 0.12]))
            inst.set_category_rates(np.linspace(0.2, 2.0, K))                                                  # (runs the queue: the old rates)
            inst.update_transition_matrices(0, np.array([2], dtype=np.int32), np.array([0.13]))
            out += [inst.get_transition_matrix(m) for m in range(3)]
            for rep in range(40):                                     # 40 x 12 jobs x 3 calls and the lists below: the ring wraps several times
                for f, w in ((0, 0), (0, 0), (1, 1), (0, 1)):
                    inst.set_state_frequencies(0, freqs[f])
                    inst.set_category_weights(0, weights[w])
                    if rep in (0, 39):
                        inst.update_transition_matrices(0, np.arange(nmat, dtype=np.int32), np.linspace(0.01, 0.4, nmat) * (1 + f))
                        inst.update_partials(np.array([[2, -1, -1, 0, 0, 1, 1 + w]], dtype=np.int32), bg.BEAGLE_OP_NONE)
                        out.append(np.array([inst.calculate_root_log_likelihoods(2, 0, 0, bg.BEAGLE_OP_NONE)[1]]))
                        out.append(inst.get_site_log_likelihoods().copy())
                    else:
                        inst.update_transition_matrices(rep & 1, np.arange(nmat, dtype=np.int32), np.linspace(0.01, 0.4, nmat) * (1 + 0.01 * rep))
        finally:
            inst.finalize()
            for name in ("MBAMD_F64_NO_MATRIX_QUEUE", "MBAMD_F64_NO_RING"):
                monkeypatch.delenv(name, raising=False)
        return out
    got, want = run(False), run(True)
    assert len(got) == len(want) and len(got) > nmat + 3
    for x, y in zip(got, want):
        assert np.array_equal(x, y)
    # four different (frequencies, weights) settings: four different likelihoods (a skipped upload would repeat one)
    lnls = [float(x[0]) for x in got[nmat + 3::2][:4]]
    assert lnls[0] == lnls[1] and len({lnls[1], lnls[2], lnls[3]}) == 3, lnls


def check_parsimony_model_golden(lib, golden_dir):
    """The device Fitch down-pass against the reference's OWN parsimony-model likelihood (Likelihood_Pars,
    src/likelihood.c:7593-7700; golden values written by tools/gen_golden_pars.py from oracle/_ref/mb with `lset parsmodel=yes`):
    lnL = -(tree length + number of characters) ln(number of states), to the printed digits."""
    from mrbayes_amd import data as mbdata
    from mrbayes_amd import parsimony as mp
    with open(os.path.join(golden_dir, "parsmodel.json")) as fh:
        cases = json.load(fh)["cases"]
    assert len(cases) >= 3
    for case in cases:
        st = mbdata.synthetic_states(case["ntaxa"], case["nsites"], case["nstates"], case["seed"], 0.15, case["p_gap"])
        tr = mbtree.random_tree(case["ntaxa"], case["tree_seed"], brlen=0.05)
        inst = mp.ParsimonyInstance(tr.n_nodes, st.shape[1], case["nstates"], lib=lib)
        try:
            sets = mp.tip_sets(st, case["nstates"])
            for i in range(st.shape[0]):
                inst.set_sets(i, sets[i])
            inst.set_pattern_weights(np.ones(st.shape[1], dtype=np.float32))
            length = mp.GetParsimonyLength(inst, tr)
            assert length == int(length)
            lnl = -(length + st.shape[1]) * math.log(case["nstates"])
            # The reference prints the value with 13 significant digits -- and adds the tree length up in CLFlt (float,
            # src/likelihood.c:7597-7672): beyond 2^24 steps its own sum is rounded (the configs[3] shape: 27 062 103 steps printed,
            # 27 062 126 counted; the device length is an exact integer, pinned word for word against GetParsDP / GetParsFP by
            # MBAMD_PARS_CHECK=1, tests/test_mrbayes_dropin.py).  There the golden pins to the float's resolution.
            tol = max(1e-6, 2e-12 * abs(lnl)) if length < 2 ** 24 else 4e-6 * abs(lnl)
            assert abs(lnl - case["lnL_reference"]) <= tol, (case["name"], lnl, case["lnL_reference"])
        finally:
            inst.finalize()


