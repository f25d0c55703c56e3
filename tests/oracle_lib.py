"""ctypes front-end of the CPU oracle (oracle/mb_oracle.c).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
_lib = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def build():
    so = os.path.join(ODIR, "liboracle.so")
    srcs = [os.path.join(ODIR, "mb_oracle.c"), os.path.join(ODIR, "pars_oracle.c")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(src) for src in srcs):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c99", "-o", so] + srcs + ["-lm"], cwd=ODIR)
    return so


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.mbo_tree_loglike.restype = C.c_int
        lib.mbo_likelihood.restype = C.c_int

    # ---- unit-level routines -------------------------------------------------------------
    def calc_cijk(self, es):
        n = es.eval.size
        out = np.empty(n * n * n)
        self.lib.mbo_calc_cijk(C.c_int(n), _p(np.ascontiguousarray(es.evec), _dp),
                               _p(np.ascontiguousarray(es.ivec), _dp), _p(out, _dp))
        return out

    def tiprobs(self, div, length, part=None):
        """float32 [K][S][S] for one branch; K = ncat (single eigen part) or nCijkParts (GenCov)."""
        n = div.nstates
        pi = np.ascontiguousarray(div.pi, dtype=np.float64)
        if div.n_cijk_parts == 1:
            out = np.empty((div.ncat, n, n), dtype=np.float32)
            ev = np.ascontiguousarray(div.eigen[0].eval)
            cijk = self.calc_cijk(div.eigen[0])
            rate = np.ascontiguousarray(div.cat_rates, dtype=np.float64)
            self.lib.mbo_tiprobs_gen(C.c_int(n), C.c_int(div.ncat), _p(ev, _dp), _p(cijk, _dp),
                                     C.c_double(length), _p(rate, _dp), _p(pi, _dp), _p(out, _fp))
            return out
        k = div.n_cijk_parts
        out = np.empty((k, n, n), dtype=np.float32)
        ev = np.ascontiguousarray(np.concatenate([e.eval for e in div.eigen]))
        cijk = np.ascontiguousarray(np.concatenate([self.calc_cijk(e) for e in div.eigen]))
        self.lib.mbo_tiprobs_gencov(C.c_int(n), C.c_int(k), _p(ev, _dp), _p(cijk, _dp),
                                    C.c_double(length * div.cat_rates[0]), _p(pi, _dp), _p(out, _fp))
        return out

    def tiprobs_hky(self, kappa, pis, length, rates):
        """TiProbs_Hky (nst=2), out[k][i][j]"""
        rates = np.ascontiguousarray(rates, dtype=np.float64)
        pis = np.ascontiguousarray(pis, dtype=np.float64)
        out = np.empty((len(rates), 4, 4), dtype=np.float32)
        self.lib.mbo_tiprobs_hky(C.c_int(len(rates)), C.c_double(kappa), _p(pis, _dp), C.c_double(length), _p(rates, _dp), _p(out, _fp))
        return out

    def tiprobs_jc(self, length, rates):
        """TiProbs_JukesCantor (nst=1), out[k][i][j]"""
        rates = np.ascontiguousarray(rates, dtype=np.float64)
        out = np.empty((len(rates), 4, 4), dtype=np.float32)
        self.lib.mbo_tiprobs_jc(C.c_int(len(rates)), C.c_double(length), _p(rates, _dp), _p(out, _fp))
        return out

    def condlike_down(self, n, K, P, clL, stL, tiL, clR, stR, tiR):
        out = np.empty((K, P, n), dtype=np.float32)
        self.lib.mbo_condlike_down(C.c_int(n), C.c_int(K), C.c_int(P), _p(clL, _fp), _p(stL, _ip), _p(tiL, _fp),
                                   _p(clR, _fp), _p(stR, _ip), _p(tiR, _fp), _p(out, _fp))
        return out

    def condlike_scaler(self, n, K, P, cl, ln_scaler):
        sc = np.empty(P, dtype=np.float32)
        self.lib.mbo_condlike_scaler(C.c_int(n), C.c_int(K), C.c_int(P), _p(cl, _fp), _p(sc, _fp), _p(ln_scaler, _fp))
        return sc

    def likelihood(self, n, K, P, cl, bs, catw, ln_scaler, nsites, pinvar=0.0, cl_invar=None):
        lnl = C.c_double(0.0)
        site = np.empty(P)
        rc = self.lib.mbo_likelihood(C.c_int(n), C.c_int(K), C.c_int(P), _p(cl, _fp), _p(bs, _dp), _p(catw, _dp),
                                     _p(ln_scaler, _fp), _p(nsites, _fp), C.c_double(pinvar), _p(cl_invar, _fp),
                                     C.byref(lnl), _p(site, _dp))
        return rc, lnl.value, site

    # ---- whole-tree evaluation (native flow of LaunchLogLikeForDivision) -------------------
    def tree_loglike(self, div, use_shortcuts=True, want_sites=False):
        n, P, N = div.nstates, div.npatterns, div.ntaxa
        t = div.tree
        left = np.asarray(t.left, dtype=np.int32)
        right = np.asarray(t.right, dtype=np.int32)
        length = np.asarray(t.length, dtype=np.float64)
        idp = np.asarray(t.int_down_pass, dtype=np.int32)
        tip_states = np.zeros((N, P), dtype=np.int32)
        is_partial = np.zeros(N, dtype=np.int32)
        any_partial = any(p is not None for p in div.tip_partials)
        tip_partials = np.zeros((N, P, n), dtype=np.float32) if any_partial else None
        for i in range(N):
            if div.tip_states[i] is not None:
                tip_states[i] = div.tip_states[i]
            else:
                is_partial[i] = 1
                tip_partials[i] = div.tip_partials[i]
        pi = np.ascontiguousarray(div.pi, dtype=np.float64)
        if div.n_cijk_parts == 1:
            K, n_eigen = div.ncat, 1
            ev = np.ascontiguousarray(div.eigen[0].eval)
            cijk = self.calc_cijk(div.eigen[0])
            rate = np.ascontiguousarray(div.cat_rates, dtype=np.float64)
            catw = np.full(K, (1.0 - div.pinvar) / K)
        else:
            assert div.ncat == 1
            K = n_eigen = div.n_cijk_parts
            ev = np.ascontiguousarray(np.concatenate([e.eval for e in div.eigen]))
            cijk = np.ascontiguousarray(np.concatenate([self.calc_cijk(e) for e in div.eigen]))
            rate = np.ascontiguousarray(div.cat_rates[:1], dtype=np.float64)
            catw = np.ascontiguousarray(div.part_weights, dtype=np.float64)
        w = np.ascontiguousarray(div.weights, dtype=np.float32)
        inv = None if div.inv_condlikes is None else np.ascontiguousarray(div.inv_condlikes, dtype=np.float32)
        lnl = C.c_double(0.0)
        site = np.empty(P) if want_sites else None
        rc = self.lib.mbo_tree_loglike(
            C.c_int(n), C.c_int(K), C.c_int(P), C.c_int(N), _p(left, _ip), _p(right, _ip), _p(length, _dp),
            _p(idp, _ip), C.c_int(t.root), C.c_int(t.root_left), _p(tip_states, _ip), _p(is_partial, _ip),
            _p(tip_partials, _fp), C.c_int(n_eigen), _p(ev, _dp), _p(cijk, _dp), _p(rate, _dp), _p(pi, _dp),
            _p(catw, _dp), C.c_double(div.pinvar), _p(inv, _fp), _p(w, _fp), C.c_int(1 if use_shortcuts else 0),
            C.byref(lnl), _p(site, _dp))
        if rc != 0:
            raise FloatingPointError("oracle: site likelihood below LIKE_EPSILON")
        return (lnl.value, site) if want_sites else lnl.value


def load():
    global _lib
    if _lib is None:
        _lib = Oracle(C.CDLL(build()))
    return _lib


# ---- Fitch parsimony (oracle/pars_oracle.c) ---------------------------------------------------------------------
_up = C.POINTER(C.c_uint64)


def pars_down(sets, ops, w):
    """sets uint64 [setCount][P*words] (modified in place); returns (total length, per-operation lengths)."""
    lib = load().lib
    lib.mbo_pars_down.restype = C.c_double
    ops = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 4)
    w = np.ascontiguousarray(w, dtype=np.float32)
    P = w.size
    words = sets.shape[1] // P
    node_len = np.zeros(ops.shape[0])
    total = lib.mbo_pars_down(_p(sets, _up), C.c_int(P), C.c_int(words), _p(ops, _ip), C.c_int(ops.shape[0]), _p(w, _fp), _p(node_len, _dp))
    return total, node_len


def pars_final(sets, ops, P):
    lib = load().lib
    ops = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 4)
    lib.mbo_pars_final(_p(sets, _up), C.c_int(P), C.c_int(sets.shape[1] // P), _p(ops, _ip), C.c_int(ops.shape[0]))


def pars_score(sets, tuples, w):
    lib = load().lib
    tuples = np.ascontiguousarray(tuples, dtype=np.int32).reshape(-1, 4)
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.zeros(tuples.shape[0])
    lib.mbo_pars_score(_p(sets, _up), C.c_int(w.size), C.c_int(sets.shape[1] // w.size), _p(tuples, _ip), C.c_int(tuples.shape[0]), _p(w, _fp), _p(out, _dp))
    return out
