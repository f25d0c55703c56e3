"""One data division as the likelihood path sees it: the subset of the reference's `ModelInfo`
(src/bayes.h:1252-1455) that `LaunchLogLikeForDivision` (src/likelihood.c:7851) actually reads --
tree, tip encodings, pattern weights, eigen-systems, category rates/weights, frequencies, pInvar.

`Division` is a plain container; building it from model parameters uses `mrbayes_amd.model`.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import data as mbdata
from . import model as mbmodel
from . import tree as mbtree
from .model import EigenSystem


@dataclass
class Division:
    nstates: int                       # numModelStates S
    ncat: int                          # numRateCats: BEAGLE categoryCount (rate categories)
    tree: mbtree.Tree
    weights: np.ndarray                # [P] numSitesOfPat
    tip_states: List[Optional[np.ndarray]]     # int32 [P] (value S = missing) for compact tips
    tip_partials: List[Optional[np.ndarray]]   # float64 [P][S] for partially ambiguous tips
    eigen: List[EigenSystem]           # nCijkParts eigen-systems (1, or one per omega class)
    pi: np.ndarray                     # [S] stationary frequencies
    cat_rates: np.ndarray              # [ncat] baseRate*catRate*correction (m->inRates, mbbeagle.c:1404)
    part_weights: np.ndarray           # [nCijkParts] mixture weight of each eigen part (omegaCatFreq), 1.0 if single
    pinvar: float = 0.0
    inv_condlikes: Optional[np.ndarray] = None   # [P][S] float32: invariable-site conditional likelihoods
    brlen_factor: float = 1.0          # GenCov: t = length*correction with no category rate
    rate_matrices: Optional[List[np.ndarray]] = None   # the Q behind each eigen part (device-side eigen: mbamdSetRateMatrices)

    @property
    def npatterns(self) -> int:
        return int(self.weights.shape[0])

    @property
    def ntaxa(self) -> int:
        return self.tree.ntaxa

    @property
    def n_cijk_parts(self) -> int:
        return len(self.eigen)

    def category_weights(self, part: int) -> np.ndarray:
        """What TreeLikelihood_Beagle passes to beagleSetCategoryWeights (src/mbbeagle.c:1184-1213)."""
        freq = (1.0 - self.pinvar) / self.ncat
        return np.full(self.ncat, freq * self.part_weights[part])


def _inv_condlikes(tip_states, tip_partials, nstates, npat) -> np.ndarray:
    """InitInvCondLikes (src/mcmc.c:6631): state i can be the invariable state of a pattern iff it is
    compatible with every tip."""
    inv = np.ones((npat, nstates), dtype=np.float32)
    for st, pt in zip(tip_states, tip_partials):
        if st is not None:
            onehot = np.zeros((npat, nstates), dtype=np.float32)
            miss = st >= nstates
            onehot[np.arange(npat)[~miss], st[~miss]] = 1.0
            onehot[miss] = 1.0
            inv *= onehot
        else:
            inv *= pt.astype(np.float32)
    return inv


def _tips_from_patterns(pat: mbdata.Patterns):
    tip_states, tip_partials = [], []
    for t in range(pat.ntaxa):
        if pat.is_part_ambig(t):
            tip_states.append(None)
            tip_partials.append(pat.tip_partials(t))
        else:
            tip_states.append(pat.tip_states(t))
            tip_partials.append(None)
    return tip_states, tip_partials


def _tips_from_states(states: np.ndarray):
    return [np.ascontiguousarray(states[t], dtype=np.int32) for t in range(states.shape[0])], [None] * states.shape[0]


def load_wag(golden_dir: str):
    with open(os.path.join(golden_dir, "aa_wag.json")) as fh:
        d = json.load(fh)
    pi = np.asarray(d["pi"], dtype=np.float64)
    return np.asarray(d["exchangeability"], dtype=np.float64), pi


def build_division(kind: str, tree, weights, tip_states, tip_partials, *, revmat=None, pi=None, alpha=None,
                   ncat=1, pinvar=0.0, omegas=None, omega_freqs=None, nst=1, wag=None) -> Division:
    """Assemble a Division from model parameters the way UpDateCijk + TreeTiProbs_Beagle do."""
    npat = len(weights)
    if kind == "gtr":
        s = 4
        pi = np.asarray(pi, dtype=np.float64)
        qmats = [mbmodel.gtr_q(revmat, pi)]
        eig = [mbmodel.eigen_reversible(qmats[0], pi)]
        corr = 1.0
        part_w = np.ones(1)
    elif kind == "wag":
        s = 20
        exch, wpi = wag
        pi = wpi / wpi.sum() if pi is None else np.asarray(pi, dtype=np.float64)
        qmats = [mbmodel.exchangeability_q(exch, pi)]
        eig = [mbmodel.eigen_reversible(qmats[0], pi)]
        corr = 1.0
        part_w = np.ones(1)
    elif kind == "m3":
        s = 61
        pi = np.full(61, 1.0 / 61) if (pi is None or isinstance(pi, str)) else np.asarray(pi, dtype=np.float64)
        qs = mbmodel.m3_qs(omegas, omega_freqs, pi, nst=nst, rates=revmat)
        qmats = list(qs)
        eig = [mbmodel.eigen_reversible(q, pi) for q in qs]
        corr = 3.0                                   # codon correction factor, src/mbbeagle.c:1385-1386
        part_w = np.asarray(omega_freqs, dtype=np.float64)
    elif kind.startswith("gen"):                     # any reversible model of int(kind[3:]) states: restriction sites (2),
        s = int(kind[3:])                            # covarion nucleotides (8), doublets (16), ... -- same arithmetic as "wag"
        rng = np.random.default_rng(4321 + s)
        ex = rng.gamma(1.0, 1.0, size=(s, s)); ex = ex + ex.T
        pi = rng.dirichlet(np.full(s, 5.0)) if pi is None else np.asarray(pi, dtype=np.float64)
        qmats = [mbmodel.exchangeability_q(ex, pi)]
        eig = [mbmodel.eigen_reversible(qmats[0], pi)]
        corr = 1.0
        part_w = np.ones(1)
    else:
        raise ValueError(kind)
    rates = mbmodel.discrete_gamma(alpha, ncat) if (alpha is not None and ncat > 1) else np.ones(ncat)
    base_rate = 1.0
    if pinvar > 0.0:
        base_rate /= (1.0 - pinvar)                  # src/mbbeagle.c:1393-1395
    inv = _inv_condlikes(tip_states, tip_partials, s, npat) if pinvar > 0.0 else None
    return Division(s, ncat, tree, np.asarray(weights, dtype=np.float64), tip_states, tip_partials, eig, pi,
                    base_rate * rates * corr, part_w, pinvar, inv, corr, qmats)


def division_from_golden(golden_dir: str, case: str) -> Division:
    with open(os.path.join(golden_dir, case + ".json")) as fh:
        g = json.load(fh)
    m = g["model"]
    tr = mbtree.parse_newick(g["newick"])
    if "synthetic" in g:
        sy = g["synthetic"]
        st = mbdata.synthetic_states(sy["ntaxa"], sy["nsites"], sy["nstates"], sy["seed"], sy["p_mut"], sy["p_gap"])
        st, w = mbdata.unique_columns(st)
        tip_states, tip_partials = _tips_from_states(st)
    else:
        z = np.load(os.path.join(golden_dir, case + ".npz"))
        bits = z["bits"]
        nst = {"dna": 4, "protein": 20, "codon": 61}[g["datatype"]]
        pat = mbdata.Patterns(nst, [[int(x) for x in row] for row in bits], z["weights"])
        w = pat.weights
        tip_states, tip_partials = _tips_from_patterns(pat)
    kw = {}
    if m["kind"] == "gtr":
        kw = dict(revmat=m["revmat"], pi=m["pi"], alpha=m["alpha"], ncat=m["ncat"], pinvar=m["pinvar"])
    elif m["kind"] == "wag":
        kw = dict(alpha=m["alpha"], ncat=m["ncat"], pinvar=m["pinvar"], wag=load_wag(golden_dir))
    elif m["kind"] == "m3":
        row = g["reference_params_row0"]
        kw = dict(omegas=[float(row["omega(%d)" % i]) for i in (1, 2, 3)],
                  omega_freqs=[float(row["pi(%d)" % i]) for i in (1, 2, 3)], nst=m["nst"], ncat=1)
    return build_division(m["kind"], tr, w, tip_states, tip_partials, **kw)


def synthetic_division(kind: str, ntaxa: int, npatterns: int, seed: int = 7, tree_seed: int = 3,
                       alpha: float = 1.0, ncat: int = 4, p_gap: float = 0.0, golden_dir: Optional[str] = None,
                       brlen: Optional[float] = 0.05) -> Division:
    """Synthetic inputs of the BASELINE shapes: every column is kept as its own pattern
    (weight 1) so P is exactly `npatterns`."""
    nstates = int(kind[3:]) if kind.startswith("gen") else {"gtr": 4, "wag": 20, "m3": 61}[kind]
    st = mbdata.synthetic_states(ntaxa, npatterns, nstates, seed, 0.15, p_gap)
    tr = mbtree.random_tree(ntaxa, tree_seed, brlen=brlen)
    tip_states, tip_partials = _tips_from_states(st)
    w = np.ones(npatterns)
    if kind == "gtr":
        return build_division("gtr", tr, w, tip_states, tip_partials, revmat=[0.10, 0.30, 0.05, 0.08, 0.40, 0.07],
                              pi=[0.35, 0.25, 0.15, 0.25], alpha=alpha, ncat=ncat)
    if kind == "wag":
        if golden_dir is not None and os.path.exists(os.path.join(golden_dir, "aa_wag.json")):
            wag = load_wag(golden_dir)
        else:   # any reversible 20-state matrix exercises the same arithmetic
            rng = np.random.default_rng(1234)
            ex = rng.gamma(1.0, 1.0, size=(20, 20)); ex = ex + ex.T
            p = rng.dirichlet(np.full(20, 5.0))
            wag = (ex, p)
        return build_division("wag", tr, w, tip_states, tip_partials, alpha=alpha, ncat=ncat, wag=wag)
    if kind.startswith("gen"):
        return build_division(kind, tr, w, tip_states, tip_partials, alpha=alpha, ncat=ncat)
    return build_division("m3", tr, w, tip_states, tip_partials, omegas=[0.1, 1.0, 3.0],
                          omega_freqs=[0.5, 0.3, 0.2], ncat=1)
