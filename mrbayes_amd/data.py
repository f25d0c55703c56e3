"""Alignment -> site patterns -> tip encodings, as the likelihood path receives them.

The reference does this in setup code that is out of scope for the engine (`CompressData`,
src/model.c:2466; `SetUpTermState`, src/mcmc.c:18543; tip upload in `InitBeagleInstance`,
src/mbbeagle.c:116-168); this module produces the same *inputs* for tests, fixtures and the bench:

  * every tip is a [P] vector of state bit-sets (like `parsSets`);
  * a tip with only single states / fully-missing sites becomes a *compact* tip (int state codes,
    value S = missing) -> `beagleSetTipStates`;
  * a tip with any partially ambiguous site becomes a *partials* tip ([P][S] 0/1) ->
    `beagleSetTipPartials` (src/mbbeagle.c:98-108, 150-166).
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .model import AA_ORDER, NUC, sense_codons

_DNA_AMBIG: Dict[str, str] = {
    "A": "A", "C": "C", "G": "G", "T": "T", "U": "T",
    "R": "AG", "Y": "CT", "M": "AC", "K": "GT", "S": "CG", "W": "AT",
    "H": "ACT", "B": "CGT", "V": "ACG", "D": "AGT", "N": "ACGT", "-": "ACGT", "?": "ACGT",
}


def dna_bits(ch: str) -> int:
    ch = ch.upper()
    if ch not in _DNA_AMBIG:
        raise ValueError("bad nucleotide %r" % ch)
    b = 0
    for c in _DNA_AMBIG[ch]:
        b |= 1 << NUC.index(c)
    return b


def aa_bits(ch: str) -> int:
    ch = ch.upper()
    if ch in AA_ORDER:
        return 1 << AA_ORDER.index(ch)
    if ch == "B":   # Asn or Asp
        return (1 << AA_ORDER.index("N")) | (1 << AA_ORDER.index("D"))
    if ch == "Z":   # Gln or Glu
        return (1 << AA_ORDER.index("Q")) | (1 << AA_ORDER.index("E"))
    if ch in "X-?":
        return (1 << 20) - 1
    raise ValueError("bad amino acid %r" % ch)


def codon_bits(trip: str) -> int:
    """Bit-set over the 61 sense codons compatible with a (possibly ambiguous) nucleotide triplet."""
    nucs, _ = sense_codons()
    sets = [dna_bits(c) for c in trip]
    b = 0
    for s, (n1, n2, n3) in enumerate(nucs):
        if (sets[0] >> n1) & 1 and (sets[1] >> n2) & 1 and (sets[2] >> n3) & 1:
            b |= 1 << s
    if b == 0:
        raise ValueError("triplet %r is a stop codon" % trip)
    return b


@dataclass
class Patterns:
    """Compressed data of one division."""
    nstates: int
    bits: List[List[int]]            # [ntaxa][P] python ints (bit-sets over states)
    weights: np.ndarray              # [P] float64 number of sites of each pattern
    names: Optional[List[str]] = None

    @property
    def ntaxa(self) -> int:
        return len(self.bits)

    @property
    def npatterns(self) -> int:
        return len(self.bits[0])

    def is_part_ambig(self, taxon: int) -> bool:
        full = (1 << self.nstates) - 1
        return any((b != full) and (b & (b - 1)) != 0 for b in self.bits[taxon])

    def tip_states(self, taxon: int) -> np.ndarray:
        """int32 [P]; S = missing.  Only valid for compact tips."""
        full = (1 << self.nstates) - 1
        out = np.empty(self.npatterns, dtype=np.int32)
        for c, b in enumerate(self.bits[taxon]):
            out[c] = self.nstates if b == full else b.bit_length() - 1
        return out

    def tip_partials(self, taxon: int) -> np.ndarray:
        """float64 [P][S] of 0/1."""
        out = np.zeros((self.npatterns, self.nstates))
        for c, b in enumerate(self.bits[taxon]):
            for s in range(self.nstates):
                if (b >> s) & 1:
                    out[c, s] = 1.0
        return out


def compress(columns: Sequence[Tuple[int, ...]], nstates: int, names=None) -> Patterns:
    """Collapse identical site columns (tuples of per-taxon bit-sets) into weighted patterns,
    keeping first-occurrence order."""
    index: Dict[Tuple[int, ...], int] = {}
    weights: List[float] = []
    uniq: List[Tuple[int, ...]] = []
    for col in columns:
        k = index.get(col)
        if k is None:
            index[col] = len(uniq)
            uniq.append(col)
            weights.append(1.0)
        else:
            weights[k] += 1.0
    ntaxa = len(uniq[0])
    bits = [[col[t] for col in uniq] for t in range(ntaxa)]
    return Patterns(nstates, bits, np.asarray(weights), list(names) if names else None)


def patterns_from_sequences(seqs: Sequence[str], datatype: str, names=None) -> Patterns:
    n = len(seqs[0])
    if datatype == "dna":
        cols = [tuple(dna_bits(s[c]) for s in seqs) for c in range(n)]
        return compress(cols, 4, names)
    if datatype == "protein":
        cols = [tuple(aa_bits(s[c]) for s in seqs) for c in range(n)]
        return compress(cols, 20, names)
    if datatype == "codon":
        assert n % 3 == 0
        cols = [tuple(codon_bits(s[c:c + 3]) for s in seqs) for c in range(0, n, 3)]
        return compress(cols, 61, names)
    raise ValueError(datatype)


def read_nexus_matrix(path: str) -> Tuple[List[str], List[str], str]:
    """Tiny reader for the non-interleaved/interleaved `matrix` block of the example files:
    returns (names, sequences, datatype).  Test/tool use only."""
    names: List[str] = []
    seqs: Dict[str, List[str]] = {}
    datatype = "dna"
    matchchar = None
    in_matrix = False
    in_comment = 0
    with open(path) as fh:
        for raw in fh:
            line = ""
            for ch in raw:          # strip [comments]
                if ch == "[":
                    in_comment += 1
                elif ch == "]":
                    in_comment -= 1
                elif in_comment == 0:
                    line += ch
            s = line.strip()
            low = s.lower()
            if not in_matrix:
                if low.startswith("format"):
                    if "datatype=protein" in low.replace(" ", ""):
                        datatype = "protein"
                    elif "datatype=rna" in low.replace(" ", ""):
                        datatype = "dna"
                    mm = re.search(r"matchchar\s*=\s*(\S)", low)
                    if mm:
                        matchchar = mm.group(1)
                if low.startswith("matrix"):
                    in_matrix = True
                continue
            if s.startswith(";") or low.startswith("end"):
                break
            if not s:
                continue
            parts = s.split()
            name, chunk = parts[0], "".join(parts[1:]).rstrip(";")
            if name not in seqs:
                seqs[name] = []
                names.append(name)
            seqs[name].append(chunk)
            if s.endswith(";"):
                break
    full = ["".join(seqs[n]) for n in names]
    if matchchar:
        first = full[0]
        full = ["".join(first[i] if c == matchchar else c for i, c in enumerate(q)) for q in full]
    return names, full, datatype


# ---------------------------------------------------------------------------------------------
# Synthetic alignments: a running sequence, each taxon mutates it with p per site.
# ---------------------------------------------------------------------------------------------
def synthetic_states(ntaxa: int, nsites: int, nstates: int, seed: int, p_mut: float = 0.15,
                     p_gap: float = 0.0) -> np.ndarray:
    """int32 [ntaxa][nsites] state codes; value nstates = gap/missing.  With nstates^... columns the
    patterns are unique with high probability (callers may still compress)."""
    rng = np.random.default_rng(seed)
    cur = rng.integers(0, nstates, size=nsites, dtype=np.int32)
    out = np.empty((ntaxa, nsites), dtype=np.int32)
    for t in range(ntaxa):
        mut = rng.random(nsites) < p_mut
        new = rng.integers(0, nstates, size=nsites, dtype=np.int32)
        cur = np.where(mut, new, cur)
        out[t] = cur
        if p_gap > 0:
            gap = rng.random(nsites) < p_gap
            out[t] = np.where(gap, nstates, out[t])
    return out


def unique_columns(states: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Compress state-code columns: returns (states[:, unique], weights)."""
    _, first, counts = np.unique(states.T, axis=0, return_index=True, return_counts=True)
    order = np.argsort(first)
    return np.ascontiguousarray(states[:, first[order]]), counts[order].astype(np.float64)
