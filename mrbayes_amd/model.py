"""Host-side substitution-model numerics feeding the likelihood engine.

This is the part of the path the reference keeps on the host, executed once per Q-changing move
(`UpDateCijk`, reference src/likelihood.c:10476): build the instantaneous rate matrix, eigen-
decompose it, and hand (U, U^-1, lambda) to the engine through `beagleSetEigenDecomposition`
(src/likelihood.c:10636-10660).  Restated from the published model definitions, fp64 numpy:

  * GTR 4x4                      -- SetNucQMatrix, src/likelihood.c:8226-8267
  * empirical amino-acid models  -- SetProteinQMatrix, src/likelihood.c:8765 (exchangeabilities x pi,
                                    normalised to one expected substitution; src/model.c:18034-18260)
  * codon (Goldman-Yang/NY98) + M3 joint rescale -- src/likelihood.c:8543-8745, 10688-10712
  * discrete gamma (Yang 1994, category means)   -- DiscreteGamma, src/utils.c:10500-10534

Eigen-decomposition: the reference uses a general real Hessenberg-QR (GetEigens, src/utils.c:11201);
every model on the BEAGLE path is time-reversible, so we use the symmetric similarity transform
B = D^1/2 Q D^-1/2 and a symmetric eigen-solver, which is real by construction.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
from scipy import special, stats

NUC = "ACGT"
AA_ORDER = "ARNDCQEGHILKMFPSTWYV"     # MrBayes / PAML amino-acid state order

# universal genetic code, codons ordered with A<C<G<T at each position (src/model.c:18296-18370);
# '*' marks the three stop codons that are removed from the 61-state model.
_UNIVERSAL = ("KNKNTTTTRSRSIIMI" "QHQHPPPPRRRRLLLL" "EDEDAAAAGGGGVVVV" "*Y*YSSSS*CWCLFLF")


def sense_codons(code: str = _UNIVERSAL) -> Tuple[List[Tuple[int, int, int]], List[str]]:
    """(codonNucs, codonAAs) for the sense codons, in state order (src/model.c:18452-18466)."""
    nucs, aas = [], []
    for s1 in range(4):
        for s2 in range(4):
            for s3 in range(4):
                aa = code[s1 * 16 + s2 * 4 + s3]
                if aa != "*":
                    nucs.append((s1, s2, s3))
                    aas.append(aa)
    return nucs, aas


@dataclass
class EigenSystem:
    """What beagleSetEigenDecomposition receives: row-major U, U^-1, real eigenvalues."""
    evec: np.ndarray      # [S,S]
    ivec: np.ndarray      # [S,S]
    eval: np.ndarray      # [S]


def eigen_reversible(q: np.ndarray, pi: np.ndarray) -> EigenSystem:
    pi = np.asarray(pi, dtype=np.float64)
    if np.any(pi <= 0):
        w, v = np.linalg.eig(q)
        if np.max(np.abs(w.imag)) > 1e-12:
            raise ValueError("complex eigenvalues (the reference rejects these on the BEAGLE path, "
                             "src/likelihood.c:10623-10631)")
        v = v.real
        return EigenSystem(v, np.linalg.inv(v), w.real)
    d = np.sqrt(pi)
    b = (d[:, None] * q) / d[None, :]
    b = 0.5 * (b + b.T)
    w, v = np.linalg.eigh(b)
    evec = v / d[:, None]
    ivec = v.T * d[None, :]
    return EigenSystem(np.ascontiguousarray(evec), np.ascontiguousarray(ivec), w)


def gtr_q(rates: Sequence[float], pi: Sequence[float]) -> np.ndarray:
    """rates = (AC, AG, AT, CG, CT, GT).  src/likelihood.c:8226-8267."""
    pi = np.asarray(pi, dtype=np.float64)
    r = np.zeros((4, 4))
    idx = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    for (i, j), x in zip(idx, rates):
        r[i, j] = r[j, i] = x
    q = r * pi[None, :]
    np.fill_diagonal(q, 0.0)
    np.fill_diagonal(q, -q.sum(axis=1))
    scaler = -(pi * np.diag(q)).sum()
    return q / scaler


def exchangeability_q(s: np.ndarray, pi: Sequence[float]) -> np.ndarray:
    """General reversible model Q_ij = s_ij * pi_j, mean rate 1 (empirical AA models, aamodel GTR)."""
    pi = np.asarray(pi, dtype=np.float64)
    q = np.array(s, dtype=np.float64) * pi[None, :]
    np.fill_diagonal(q, 0.0)
    np.fill_diagonal(q, -q.sum(axis=1))
    scaler = -(pi * np.diag(q)).sum()
    return q / scaler


_GTR_PAIR = {(0, 1): 0, (0, 2): 1, (0, 3): 2, (1, 2): 3, (1, 3): 4, (2, 3): 5}


def codon_q_unscaled(omega: float, pi: Sequence[float], nst: int = 1,
                     rates: Optional[Sequence[float]] = None,
                     code: str = _UNIVERSAL) -> Tuple[np.ndarray, float, float]:
    """Unscaled codon rate matrix + (dN, dS) flux.  src/likelihood.c:8575-8745.
    nst=1: F81-like; nst=2: rates=(kappa,); nst=6: rates = 6 GTR rates."""
    nucs, aas = sense_codons(code)
    n = len(nucs)
    pi = np.asarray(pi, dtype=np.float64)
    q = np.zeros((n, n))
    dn = ds = 0.0
    for i in range(n):
        for j in range(i + 1, n):
            diff = [(nucs[i][k], nucs[j][k]) for k in range(3) if nucs[i][k] != nucs[j][k]]
            if len(diff) != 1:
                continue
            a, b = sorted(diff[0])
            mult = 1.0 if aas[i] == aas[j] else omega
            if nst == 2:
                if (a, b) in ((0, 2), (1, 3)):
                    mult *= rates[0]
            elif nst == 6:
                mult *= rates[_GTR_PAIR[(a, b)]]
            q[i, j] = pi[j] * mult
            q[j, i] = pi[i] * mult
            flux = pi[i] * q[i, j] + pi[j] * q[j, i]
            if aas[i] == aas[j]:
                ds += flux
            else:
                dn += flux
    np.fill_diagonal(q, -q.sum(axis=1))
    return q, dn, ds


def codon_q(omega: float, pi, nst: int = 1, rates=None) -> np.ndarray:
    q, dn, ds = codon_q_unscaled(omega, pi, nst, rates)
    return q / (dn + ds)


def m3_qs(omegas: Sequence[float], omega_freqs: Sequence[float], pi, nst: int = 1, rates=None) -> List[np.ndarray]:
    """One Q per omega class, jointly rescaled so the *mixture* has mean rate 1
    (src/likelihood.c:10688-10712)."""
    qs, scal = [], 0.0
    for w, f in zip(omegas, omega_freqs):
        q, dn, ds = codon_q_unscaled(w, pi, nst, rates)
        qs.append(q)
        scal += f * (dn + ds)
    return [q / scal for q in qs]


def discrete_gamma(alpha: float, k: int, median: bool = False) -> np.ndarray:
    """Mean (default) or median category rates of a Gamma(alpha, alpha).  src/utils.c:10500-10534."""
    beta = alpha
    if k == 1:
        return np.ones(1)
    if median:
        r = stats.gamma.ppf((2.0 * np.arange(k) + 1.0) / (2.0 * k), alpha, scale=1.0 / beta)
        return r * (k * alpha / beta) / r.sum()
    cut = stats.gamma.ppf(np.arange(1, k) / k, alpha, scale=1.0 / beta)
    cum = np.concatenate([special.gammainc(alpha + 1.0, cut * beta), [1.0]])
    r = np.diff(np.concatenate([[0.0], cum])) * (alpha / beta * k)
    return r


def transition_matrices(es: EigenSystem, t: float, cat_rates: Sequence[float]) -> np.ndarray:
    """fp64 P_k(t) = U diag(exp(lambda t r_k)) U^-1 -- test helper, not on the product path."""
    out = np.empty((len(cat_rates), es.eval.size, es.eval.size))
    for k, r in enumerate(cat_rates):
        out[k] = (es.evec * np.exp(es.eval * t * r)[None, :]) @ es.ivec
    return out
