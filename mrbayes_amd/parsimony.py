"""Host side of the device parsimony scorer above its C ABI (include/libhmsbeagle/mbamd_parsimony.h): the Python twin of
the reference's parsimony helpers, same names and argument meaning, so that the parity tests read like the reference:

    InitParsSets (tip sets)          src/mcmc.c:6897-7040
    GetParsDP / GetFitchPartials     src/mcmc.c:4849-4876, 4794-4846
    GetParsFP                        src/mcmc.c:4881-4954
    GetParsimonyLength               src/mcmc.c:5016-5073
    candidate lengths of ParsSPR1    src/proposal.c:10783-10876

The real MrBayes binds the ABI through integration/mrbayes/mbamd_pars_glue.c; this module exists for
tests, tools and the benchmark.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import beagle as bg
from .tree import Tree

_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_ulonglong)

PARS_EXPORTS = ["mbamdParsCreateInstance", "mbamdParsFinalizeInstance", "mbamdParsSetSets", "mbamdParsGetSets",
                "mbamdParsSetPatternWeights", "mbamdParsDownPass", "mbamdParsFinalPass", "mbamdParsScore"]


def _declare(L) -> None:
    if getattr(L, "_mbamd_pars_declared", False):
        return
    for name in PARS_EXPORTS:
        if not hasattr(L, name):
            raise AttributeError("the engine library does not export %s" % name)
    L.mbamdParsCreateInstance.argtypes = [C.c_int] * 5
    L.mbamdParsFinalizeInstance.argtypes = [C.c_int]
    L.mbamdParsSetSets.argtypes = [C.c_int, C.c_int, _up]
    L.mbamdParsGetSets.argtypes = [C.c_int, C.c_int, _up]
    L.mbamdParsSetPatternWeights.argtypes = [C.c_int, _fp]
    L.mbamdParsDownPass.argtypes = [C.c_int, _ip, C.c_int, _dp]
    L.mbamdParsFinalPass.argtypes = [C.c_int, _ip, C.c_int]
    L.mbamdParsScore.argtypes = [C.c_int, _ip, C.c_int, _dp]
    L._mbamd_pars_declared = True


def tip_sets(states: np.ndarray, nstates: int) -> np.ndarray:
    """InitParsSets for one-word sets (src/mcmc.c:6931-6946): states [ntaxa][P] of state codes 0..nstates-1, or any
    value >= nstates / < 0 for missing/gap (all states possible) -> uint64 [ntaxa][P] with one bit per state."""
    s = np.asarray(states)
    out = np.left_shift(np.uint64(1), np.clip(s, 0, 63).astype(np.uint64))
    all_ambig = np.uint64((1 << nstates) - 1) if nstates < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    out[(s < 0) | (s >= nstates)] = all_ambig
    return out


class ParsimonyInstance:
    """The Fitch state sets of one division on the device (m->parsSets)."""

    def __init__(self, set_count: int, pattern_count: int, set_bits: int, words_per_set: int = 1,
                 lib: Optional[bg.BeagleLibrary] = None, likelihood_instance: int = -1):
        self.lib = lib or bg.library()
        _declare(self.lib.lib)
        self.P = pattern_count
        self.words = words_per_set
        self.set_count = set_count
        self.id = self.lib.lib.mbamdParsCreateInstance(set_count, pattern_count, words_per_set, set_bits, likelihood_instance)
        if self.id < 0:
            raise bg.BeagleError(self.id, "mbamdParsCreateInstance", self.lib.last_error())

    def _chk(self, code: int, where: str):
        if code != 0:
            raise bg.BeagleError(code, where, self.lib.last_error())

    def finalize(self):
        if self.id >= 0:
            self.lib.lib.mbamdParsFinalizeInstance(self.id)
            self.id = -1

    def __del__(self):
        try:
            self.finalize()
        except Exception:
            pass

    def set_sets(self, index: int, sets):
        a = np.ascontiguousarray(sets, dtype=np.uint64)
        assert a.size == self.P * self.words
        self._chk(self.lib.lib.mbamdParsSetSets(self.id, index, a.ctypes.data_as(_up)), "mbamdParsSetSets")

    def get_sets(self, index: int) -> np.ndarray:
        out = np.empty(self.P * self.words, dtype=np.uint64)
        self._chk(self.lib.lib.mbamdParsGetSets(self.id, index, out.ctypes.data_as(_up)), "mbamdParsGetSets")
        return out

    def all_sets(self) -> np.ndarray:
        return np.stack([self.get_sets(i) for i in range(self.set_count)])

    def set_pattern_weights(self, w):
        a = np.ascontiguousarray(w, dtype=np.float32)
        assert a.size == self.P
        self._chk(self.lib.lib.mbamdParsSetPatternWeights(self.id, a.ctypes.data_as(_fp)), "mbamdParsSetPatternWeights")

    def down_pass(self, ops: Sequence[Sequence[int]], want_length: bool = True) -> Optional[float]:
        a = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 4)
        out = C.c_double(0.0)
        self._chk(self.lib.lib.mbamdParsDownPass(self.id, a.ctypes.data_as(_ip), a.shape[0], C.byref(out) if want_length else None),
                  "mbamdParsDownPass")
        return out.value if want_length else None

    def final_pass(self, ops: Sequence[Sequence[int]]):
        a = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 4)
        self._chk(self.lib.lib.mbamdParsFinalPass(self.id, a.ctypes.data_as(_ip), a.shape[0]), "mbamdParsFinalPass")

    def score(self, tuples: Sequence[Sequence[int]]) -> np.ndarray:
        a = np.ascontiguousarray(tuples, dtype=np.int32).reshape(-1, 4)
        out = np.empty(a.shape[0], dtype=np.float64)
        self._chk(self.lib.lib.mbamdParsScore(self.id, a.ctypes.data_as(_ip), a.shape[0], out.ctypes.data_as(_dp)), "mbamdParsScore")
        return out


# ---- the reference's traversals as operation lists ---------------------------------------------------------------
def down_pass_ops(t: Tree, p: int) -> List[List[int]]:
    """GetParsDP(t, p, chain): post-order over the subtree of p, {node, left, right, -1} per interior node."""
    ops: List[List[int]] = []
    stack = [(p, False)]
    while stack:
        node, done = stack.pop()
        if t.left[node] < 0:
            continue
        if done:
            ops.append([node, t.left[node], t.right[node], -1])
            continue
        stack.append((node, True))
        stack.append((t.right[node], False))
        stack.append((t.left[node], False))
    return ops


def final_pass_ops(t: Tree, p: int) -> List[List[int]]:
    """GetParsFP(t, p, chain): pre-order (node, then left subtree, then right subtree), {node, left, right, anc}."""
    ops: List[List[int]] = []
    stack = [p]
    while stack:
        node = stack.pop()
        if t.left[node] < 0:
            continue
        ops.append([node, t.left[node], t.right[node], t.anc[node]])
        stack.append(t.right[node])
        stack.append(t.left[node])
    return ops


def GetParsDP(pars: ParsimonyInstance, t: Tree, p: int, want_length: bool = True) -> Optional[float]:
    return pars.down_pass(down_pass_ops(t, p), want_length)


def GetParsFP(pars: ParsimonyInstance, t: Tree, p: int) -> None:
    pars.final_pass(final_pass_ops(t, p))


def GetParsimonyLength(pars: ParsimonyInstance, t: Tree) -> float:
    """src/mcmc.c:5016-5073: the down-pass length plus the branch to the calculation root (a tip) of an unrooted tree."""
    length = GetParsDP(pars, t, t.root_left)
    return length + float(pars.score([[t.root_left, -1, t.root, -1]])[0])
