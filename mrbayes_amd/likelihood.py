"""Host side of the hot path above the C ABI: the Python twin of MrBayes' BEAGLE adapter.

Function-for-function mirror of the reference's src/mbbeagle.c (same names, same argument meaning, same
call protocol), so that the parity tests read like the reference and so that bench.py drives the
engine exactly the way an unmodified MrBayes does:

    InitBeagleInstance                       src/mbbeagle.c:60    (+ index tables of InitChainCondLikes,
                                                                   src/mcmc.c:5907-6253)
    TreeTiProbs_Beagle                       src/mbbeagle.c:1368
    TreeCondLikes_Beagle_No_Rescale          src/mbbeagle.c:783
    TreeCondLikes_Beagle_Rescale_All         src/mbbeagle.c:884
    TreeCondLikes_Beagle_Always_Rescale      src/mbbeagle.c:995
    TreeLikelihood_Beagle                    src/mbbeagle.c:1117
    LaunchBEAGLELogLikeForDivision           src/mbbeagle.c:400
    Flip*Space / ResetFlips                  src/likelihood.c:5614-5683, src/mcmc.c:15695-15766

The real MrBayes needs none of this -- it links libhmsbeagle.so directly; this module
exists for tests, the benchmark and the multi-GPU chain driver.
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np

from . import beagle as bg
from .division import Division

MB_BEAGLE_SCALE_ALWAYS = 0
MB_BEAGLE_SCALE_DYNAMIC = 1
BRLENS_MIN = 1e-8      # src/bayes.h
BRLENS_MAX = 100.0
BEAGLE_RESCALE_FREQ = 160   # src/bayes.h:77


class BeagleDivision:
    """ModelInfo-like state of one division driven through the BEAGLE ABI for `nchains` local chains."""

    def __init__(self, div: Division, lib: Optional[bg.BeagleLibrary] = None, nchains: int = 1,
                 scaling: int = MB_BEAGLE_SCALE_ALWAYS, resource: Optional[int] = None, device_eigen: bool = False,
                 double_precision: bool = False):
        self.div = div
        self.double_precision = double_precision        # `set beagleprecision=double` (src/command.c:6765-6766): a preference flag
        self.device_eigen = device_eigen and div.rate_matrices is not None and bool(np.all(np.asarray(div.pi) > 0))
        self.lib = lib or bg.library()
        self.nchains = nchains
        self.scaling = scaling
        t = div.tree
        self.N = N = t.ntaxa
        self.nInt = nInt = t.n_int_nodes
        self.nNodes = nNodes = t.n_nodes
        self.step = step = div.n_cijk_parts            # indexStep = nCijkParts, src/mcmc.c:5907-5910
        # --- index tables (InitChainCondLikes) -------------------------------------------------
        self.condLikeIndex = [[0] * nNodes for _ in range(nchains)]
        for c in range(nchains):
            for i in range(N):
                self.condLikeIndex[c][i] = i * step                      # tips shared by all chains
            for j in range(nInt):
                self.condLikeIndex[c][N + j] = (N + c * nInt + j) * step
        self.condLikeScratchIndex = [0] * nNodes
        for j in range(nInt):
            self.condLikeScratchIndex[N + j] = (N + nchains * nInt + j) * step
        self.numCondLikes = N + (nchains + 1) * nInt
        self.tiProbsIndex = [[(c * nNodes + i) * step for i in range(nNodes)] for c in range(nchains)]
        self.tiProbsScratchIndex = [(nchains * nNodes + i) * step for i in range(nNodes)]
        self.numTiProbs = (nchains + 1) * nNodes
        self.nodeScalerIndex = [[0] * nNodes for _ in range(nchains)]
        for c in range(nchains):
            for j in range(nInt):
                self.nodeScalerIndex[c][N + j] = (c * nInt + j) * step
        self.nodeScalerScratchIndex = [0] * nNodes
        for j in range(nInt):
            self.nodeScalerScratchIndex[N + j] = (nchains * nInt + j) * step
        base = (nchains + 1) * nInt
        self.siteScalerIndex = [(base + c) * step for c in range(nchains)]
        self.siteScalerScratchIndex = (base + nchains) * step
        self.numScalers = (nchains + 1) * (nInt + 1)
        self.cijkIndex = [c * step for c in range(nchains)]
        self.cijkScratchIndex = nchains * step
        # --- dirty flags ("touch" state), per chain ----------------------------------------------
        self.upDateCl = [[True] * nNodes for _ in range(nchains)]
        self.upDateTi = [[True] * nNodes for _ in range(nchains)]
        self.upDateAll = [True] * nchains
        self.upDateCijk = [True] * nchains
        self.flips: List[List[tuple]] = [[] for _ in range(nchains)]
        # dynamic rescaling state (src/mcmc.c:6220-6240)
        self.rescaleFreq = [max(1, BEAGLE_RESCALE_FREQ // div.nstates)] * nchains
        self.isScalerNode = [[False] * nNodes for _ in range(nchains)]
        self.successCount = [0] * nchains
        self.rescaleBeagleAll = False
        self.InitBeagleInstance(resource)

    # ------------------------------------------------------------------------------------------
    def InitBeagleInstance(self, resource=None):
        d = self.div
        part_ambig = [d.tip_states[i] is None for i in range(self.N)]
        num_part_ambig = sum(part_ambig)
        req = bg.BEAGLE_FLAG_SCALERS_LOG if self.scaling == MB_BEAGLE_SCALE_ALWAYS else 0   # src/mbbeagle.c:191-194
        self.inst = bg.BeagleInstance(
            self.lib, self.N, self.numCondLikes * self.step, self.N - num_part_ambig, d.nstates, d.npatterns,
            (self.nchains + 1) * self.step, self.numTiProbs * self.step, d.ncat, self.numScalers * self.step,
            resource=resource, requirement_flags=req,
            preference_flags=bg.BEAGLE_FLAG_PRECISION_DOUBLE if self.double_precision else bg.BEAGLE_FLAG_PRECISION_SINGLE)
        for i in range(self.N):                                           # src/mbbeagle.c:123-167
            if not part_ambig[i]:
                self.inst.set_tip_states(i * self.step, d.tip_states[i])
            else:
                self.inst.set_tip_partials(i * self.step, d.tip_partials[i])
        self.inst.set_pattern_weights(d.weights)                          # src/mcmc.c:6264
        for i in range(self.numScalers * self.step):                      # src/mcmc.c:6268-6271
            self.inst.reset_scale_factors(i)

    def finalize(self):
        self.inst.finalize()

    # ---- buffer flips (MH accept/reject for free) ---------------------------------------------
    def _flip(self, table, scratch, chain, node, tag):
        table[chain][node], scratch[node] = scratch[node], table[chain][node]
        self.flips[chain].append((tag, node))

    def FlipCondLikeSpace(self, chain, node):
        self._flip(self.condLikeIndex, self.condLikeScratchIndex, chain, node, "cl")

    def FlipNodeScalerSpace(self, chain, node):
        self._flip(self.nodeScalerIndex, self.nodeScalerScratchIndex, chain, node, "ns")

    def FlipTiProbsSpace(self, chain, node):
        self._flip(self.tiProbsIndex, self.tiProbsScratchIndex, chain, node, "ti")

    def FlipSiteScalerSpace(self, chain):
        self.siteScalerIndex[chain], self.siteScalerScratchIndex = self.siteScalerScratchIndex, self.siteScalerIndex[chain]
        self.flips[chain].append(("ss", -1))

    def FlipCijkSpace(self, chain):
        self.cijkIndex[chain], self.cijkScratchIndex = self.cijkScratchIndex, self.cijkIndex[chain]
        self.flips[chain].append(("cijk", -1))

    def ResetFlips(self, chain):
        """Undo every flip since the last accept (move rejected), src/mcmc.c:15695-15766."""
        for tag, node in reversed(self.flips[chain]):
            if tag == "cl":
                self.condLikeIndex[chain][node], self.condLikeScratchIndex[node] = self.condLikeScratchIndex[node], self.condLikeIndex[chain][node]
            elif tag == "ns":
                self.nodeScalerIndex[chain][node], self.nodeScalerScratchIndex[node] = self.nodeScalerScratchIndex[node], self.nodeScalerIndex[chain][node]
            elif tag == "ti":
                self.tiProbsIndex[chain][node], self.tiProbsScratchIndex[node] = self.tiProbsScratchIndex[node], self.tiProbsIndex[chain][node]
            elif tag == "ss":
                self.siteScalerIndex[chain], self.siteScalerScratchIndex = self.siteScalerScratchIndex, self.siteScalerIndex[chain]
            elif tag == "cijk":
                self.cijkIndex[chain], self.cijkScratchIndex = self.cijkScratchIndex, self.cijkIndex[chain]
        self.flips[chain] = []

    def AcceptMove(self, chain):
        self.flips[chain] = []

    # ---- touching --------------------------------------------------------------------------------
    def TouchAllTreeNodes(self, chain):
        """src/proposal.c:17682: everything dirty."""
        self.upDateCl[chain] = [True] * self.nNodes
        self.upDateTi[chain] = [True] * self.nNodes
        self.upDateAll[chain] = True

    def ClearTouches(self, chain):
        self.upDateCl[chain] = [False] * self.nNodes
        self.upDateTi[chain] = [False] * self.nNodes
        self.upDateAll[chain] = False
        self.upDateCijk[chain] = False

    def TouchBranch(self, chain, node):
        """A branch-length change on `node`: new Ti for the branch, new CLs on the path to the root."""
        t = self.div.tree
        self.upDateTi[chain][node] = True
        p = t.anc[node]
        while p >= self.N and p != -1:
            self.upDateCl[chain][p] = True
            if p == t.root_left:
                break
            p = t.anc[p]

    # ---- UpDateCijk (BEAGLE branch, src/likelihood.c:10636-10660, 10736-10757) ----------------------
    def UpDateCijk(self, chain):
        self.FlipCijkSpace(chain)
        if self.device_eigen:
            # extension: the eigen-systems are computed on the device from the rate matrices -- an
            # unmodified MrBayes has no call for this, it sends finished eigen-systems like the branch below
            self.inst.set_rate_matrices(self.cijkIndex[chain], np.stack(self.div.rate_matrices), self.div.pi)
        else:
            for i, es in enumerate(self.div.eigen):
                self.inst.set_eigen_decomposition(self.cijkIndex[chain] + i, es.evec, es.ivec, es.eval)
        self.upDateAll[chain] = True

    # ---- TreeTiProbs_Beagle ------------------------------------------------------------------------
    def TreeTiProbs_Beagle(self, chain):
        d, t = self.div, self.div.tree
        self.inst.set_category_rates(d.cat_rates)                          # src/mbbeagle.c:1404-1409
        idx, lens = [], []
        for p in t.all_down_pass:
            if self.upDateTi[chain][p]:
                self.FlipTiProbsSpace(chain, p)
                length = min(max(t.length[p], BRLENS_MIN), BRLENS_MAX)     # src/mbbeagle.c:1453-1456
                lens.append(length)
                idx.append(self.tiProbsIndex[chain][p])
        if idx:
            idx = np.asarray(idx, dtype=np.int32)
            for i in range(self.step):                                     # src/mbbeagle.c:1475-1486
                self.inst.update_transition_matrices(self.cijkIndex[chain] + i, idx + i, lens)

    # ---- operation builders ------------------------------------------------------------------------
    def _op(self, chain, p):
        t = self.div.tree
        l, r = t.left[p], t.right[p]
        return [self.condLikeIndex[chain][p], bg.BEAGLE_OP_NONE, bg.BEAGLE_OP_NONE,
                self.condLikeIndex[chain][l], self.tiProbsIndex[chain][l],
                self.condLikeIndex[chain][r], self.tiProbsIndex[chain][r]]

    def _expand_parts(self, op, chil1Step, chil2Step, j):
        """The j-th cijk-part twin of an operation (src/mbbeagle.c:845-863, 951-971, 1062-1075)."""
        o = list(op)
        o[0] += j
        if o[1] != bg.BEAGLE_OP_NONE:
            o[1] += j
        if o[2] != bg.BEAGLE_OP_NONE:
            o[2] += j
        o[3] += j * chil1Step
        o[4] += j
        o[5] += j * chil2Step
        o[6] += j
        return o

    def TreeCondLikes_Beagle_No_Rescale(self, chain):
        t = self.div.tree
        ops = []
        for p in t.int_down_pass:
            if not self.upDateCl[chain][p]:
                continue
            self.FlipCondLikeSpace(chain, p)
            op = self._op(chain, p)
            if self.isScalerNode[chain][p]:
                op[2] = self.nodeScalerIndex[chain][p]
            s1 = 0 if t.left[p] < self.N else 1
            s2 = 0 if t.right[p] < self.N else 1
            for j in range(self.step):
                ops.append(self._expand_parts(op, s1, s2, j))
        if ops:
            self.inst.update_partials(np.asarray(ops, dtype=np.int32), bg.BEAGLE_OP_NONE)

    def TreeCondLikes_Beagle_Rescale_All(self, chain):
        t = self.div.tree
        per_part: List[List[list]] = [[] for _ in range(self.step)]
        for p in t.int_down_pass:
            if not self.upDateCl[chain][p]:
                self.FlipCondLikeSpace(chain, p)
            op = self._op(chain, p)
            if self.isScalerNode[chain][p]:
                self.FlipNodeScalerSpace(chain, p)
                op[1] = self.nodeScalerIndex[chain][p]
            s1 = 0 if t.left[p] < self.N else 1
            s2 = 0 if t.right[p] < self.N else 1
            for j in range(self.step):
                per_part[j].append(self._expand_parts(op, s1, s2, j))
        for j in range(self.step):
            self.inst.update_partials(np.asarray(per_part[j], dtype=np.int32), self.siteScalerIndex[chain] + j)

    def TreeCondLikes_Beagle_Always_Rescale(self, chain):
        t = self.div.tree
        per_part: List[List[list]] = [[] for _ in range(self.step)]
        remove: List[List[int]] = [[] for _ in range(self.step)]
        for p in t.int_down_pass:
            if not self.upDateCl[chain][p]:
                continue
            if not self.upDateAll[chain]:                                  # remove old scalers
                for j in range(self.step):
                    remove[j].append(self.nodeScalerIndex[chain][p] + j)
            self.FlipCondLikeSpace(chain, p)
            self.FlipNodeScalerSpace(chain, p)
            op = self._op(chain, p)
            op[1] = self.nodeScalerIndex[chain][p]
            s1 = 0 if t.left[p] < self.N else 1
            s2 = 0 if t.right[p] < self.N else 1
            for j in range(self.step):
                per_part[j].append(self._expand_parts(op, s1, s2, j))
        for j in range(self.step):
            cum = self.siteScalerIndex[chain] + j
            if not self.upDateAll[chain] and remove[j]:
                self.inst.remove_scale_factors(remove[j], cum)
            if per_part[j]:
                self.inst.update_partials(np.asarray(per_part[j], dtype=np.int32), cum)

    # ---- TreeLikelihood_Beagle -----------------------------------------------------------------------
    def TreeLikelihood_Beagle(self, chain):
        """Returns (beagle return code, lnL)."""
        d, t = self.div, self.div.tree
        p = t.root_left
        if self.upDateCijk[chain]:
            for i in range(self.step):
                self.inst.set_state_frequencies(self.cijkIndex[chain] + i, d.pi)
        if d.n_cijk_parts > 1 or d.pinvar > 0.0 or self.upDateCijk[chain]:
            for i in range(self.step):
                self.inst.set_category_weights(self.cijkIndex[chain] + i, d.category_weights(i))
        buf = [self.condLikeIndex[chain][p] + i for i in range(self.step)]
        eig = [self.cijkIndex[chain] + i for i in range(self.step)]
        cum = [self.siteScalerIndex[chain] + i for i in range(self.step)]
        child = [self.condLikeIndex[chain][t.root]] * self.step
        prob = [self.tiProbsIndex[chain][p] + i for i in range(self.step)]
        rc, lnl = self.inst.calculate_edge_log_likelihoods(buf, child, prob, eig, eig, cum)
        if not math.isfinite(lnl):
            rc = bg.BEAGLE_ERROR_FLOATING_POINT
        if rc == bg.BEAGLE_ERROR_FLOATING_POINT:
            return rc, lnl
        self.successCount[chain] += 1
        if d.pinvar > 0.0:                                                 # src/mbbeagle.c:1326-1356
            site = self.inst.get_site_log_likelihoods()
            like_i = (d.inv_condlikes.astype(np.float64) * d.pi[None, :]).sum(axis=1)
            with np.errstate(divide="ignore"):
                ln_like_i = np.log(like_i * d.pinvar)
            diff = np.where(like_i != 0.0, ln_like_i - site, -1000.0)
            contrib = np.where(diff < -200.0, site,
                               np.where(diff > 200.0, ln_like_i, site + np.log1p(np.exp(np.clip(diff, -745, 700)))))
            lnl = float((contrib * d.weights).sum())
        return rc, lnl

    # ---- ResetScalersPartition (src/mcmc.c:15780) --------------------------------------------------
    def ResetScalersPartition(self, chain, rescaleFreq):
        t = self.div.tree
        unscaled = [0] * self.nNodes
        for p in t.int_down_pass:
            l, r = t.left[p], t.right[p]
            n = 1 + (unscaled[l] if l >= self.N else 0) + (unscaled[r] if r >= self.N else 0)
            if n >= rescaleFreq:
                self.isScalerNode[chain][p] = True
                unscaled[p] = 0
            else:
                self.isScalerNode[chain][p] = False
                unscaled[p] = n

    # ---- LaunchBEAGLELogLikeForDivision --------------------------------------------------------------
    def LaunchBEAGLELogLikeForDivision(self, chain):
        if self.upDateCijk[chain]:
            self.UpDateCijk(chain)                                         # src/likelihood.c:7864-7872
        if self.scaling == MB_BEAGLE_SCALE_ALWAYS:
            self.FlipSiteScalerSpace(chain)
            if self.upDateAll[chain]:
                for i in range(self.step):
                    self.inst.reset_scale_factors(self.siteScalerIndex[chain] + i)
            else:                                                          # CopySiteScalers, src/likelihood.c:8071-8107
                for i in range(self.step):
                    self.inst.copy_scale_factors(self.siteScalerIndex[chain] + i, self.siteScalerScratchIndex + i)
            self.TreeTiProbs_Beagle(chain)
            self.TreeCondLikes_Beagle_Always_Rescale(chain)
            rc, lnl = self.TreeLikelihood_Beagle(chain)
            return lnl
        # MB_BEAGLE_SCALE_DYNAMIC (src/mbbeagle.c:429-534), without the success-count heuristics
        self.rescaleBeagleAll = False
        self.TreeTiProbs_Beagle(chain)
        self.TreeCondLikes_Beagle_No_Rescale(chain)
        rc, lnl = self.TreeLikelihood_Beagle(chain)
        if rc == bg.BEAGLE_ERROR_FLOATING_POINT:
            rescaleFreqNew = self.rescaleFreq[chain]
            self.successCount[chain] = 0
            self.rescaleBeagleAll = True
            self.FlipSiteScalerSpace(chain)
            while True:
                self.ResetScalersPartition(chain, rescaleFreqNew)
                for i in range(self.step):
                    self.inst.reset_scale_factors(self.siteScalerIndex[chain] + i)
                self.TreeCondLikes_Beagle_Rescale_All(chain)
                rc, lnl = self.TreeLikelihood_Beagle(chain)
                if rc == bg.BEAGLE_ERROR_FLOATING_POINT and rescaleFreqNew > 1:
                    for p in self.div.tree.int_down_pass:
                        if self.isScalerNode[chain][p]:
                            self.FlipNodeScalerSpace(chain, p)
                    rescaleFreqNew -= rescaleFreqNew >> 3
                    rescaleFreqNew -= 1
                    rescaleFreqNew = max(rescaleFreqNew, 1)
                    continue
                break
            self.rescaleFreq[chain] = rescaleFreqNew
        return lnl

    def LogLike(self, chain=0):
        """One full evaluation of a chain from its current dirty flags; clears them afterwards."""
        lnl = self.LaunchBEAGLELogLikeForDivision(chain)
        self.ClearTouches(chain)
        return lnl


# ------------------------------------------------------------------------------------------------
# Call recording / replay.  MrBayes' host side is C: the handful of BEAGLE calls of one generation
# cost it microseconds.  The Python twin above spends milliseconds building the same arguments, so
# throughput measurements record the exact C-ABI call sequence of an evaluation once (function
# pointer + already marshalled ctypes arguments) and replay it: the engine sees byte-for-byte the
# calls an unmodified MrBayes makes, without the interpreter in the timed path.
# ------------------------------------------------------------------------------------------------
class _RecordingLib:
    def __init__(self, real, log):
        self._real, self._log = real, log

    def __getattr__(self, name):
        fn = getattr(self._real, name)

        def call(*args):
            self._log.append((fn, args))
            return fn(*args)
        return call


class RecordedEvaluation:
    """The C-ABI calls of one LogLike(chain), replayable with `run()` -> (return code, lnL)."""

    def __init__(self, calls):
        self.calls = calls
        self._out = None
        for fn, args in calls:
            if fn.__name__ in ("beagleCalculateEdgeLogLikelihoods", "beagleCalculateRootLogLikelihoods"):
                ref = args[-3] if fn.__name__ == "beagleCalculateEdgeLogLikelihoods" else args[-1]
                self._out = ref._obj
        if self._out is None:
            raise ValueError("the recorded evaluation contains no log-likelihood call")

    def run(self):
        rc = 0
        for fn, args in self.calls:
            r = fn(*args)
            if r != 0:
                rc = r
        return rc, self._out.value


def record_evaluation(bd: BeagleDivision, chain: int = 0, touch_all: bool = True) -> RecordedEvaluation:
    """Run bd.LogLike(chain) once while logging every C-ABI call it makes."""
    log = []
    real = bd.inst.lib
    bd.inst.lib = _RecordingLib(real, log)
    try:
        if touch_all:
            bd.TouchAllTreeNodes(chain)
        bd.LogLike(chain)
        bd.AcceptMove(chain)
    finally:
        bd.inst.lib = real
    return RecordedEvaluation(log)
