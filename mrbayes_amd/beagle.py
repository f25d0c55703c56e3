"""ctypes binding of the engine's C ABI (include/libhmsbeagle/beagle.h).

This is the Python twin of what an FFI client of libhmsbeagle does: plain pointers and sizes in,
status codes out.  It loads mrbayes_amd/libhmsbeagle.so -- the HIP library built by
`python -m mrbayes_amd.build` -- and raises if it is missing: there is no fallback implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libhmsbeagle.so")

BEAGLE_SUCCESS = 0
BEAGLE_ERROR_GENERAL = -1
BEAGLE_ERROR_OUT_OF_MEMORY = -2
BEAGLE_ERROR_UNINITIALIZED_INSTANCE = -4
BEAGLE_ERROR_OUT_OF_RANGE = -5
BEAGLE_ERROR_NO_RESOURCE = -6
BEAGLE_ERROR_NO_IMPLEMENTATION = -7
BEAGLE_ERROR_FLOATING_POINT = -8
BEAGLE_OP_NONE = -1

BEAGLE_FLAG_PRECISION_SINGLE = 1 << 0
BEAGLE_FLAG_PRECISION_DOUBLE = 1 << 1
BEAGLE_FLAG_SCALING_ALWAYS = 1 << 8
BEAGLE_FLAG_SCALERS_LOG = 1 << 10
BEAGLE_FLAG_PROCESSOR_GPU = 1 << 16
BEAGLE_FLAG_SCALING_DYNAMIC = 1 << 25


class BeagleInstanceDetails(C.Structure):
    _fields_ = [("resourceNumber", C.c_int), ("resourceName", C.c_char_p), ("implName", C.c_char_p),
                ("implDescription", C.c_char_p), ("flags", C.c_long)]


class BeagleResource(C.Structure):
    _fields_ = [("name", C.c_char_p), ("description", C.c_char_p), ("supportFlags", C.c_long),
                ("requiredFlags", C.c_long)]


class BeagleResourceList(C.Structure):
    _fields_ = [("list", C.POINTER(BeagleResource)), ("length", C.c_int)]


class BeagleOperation(C.Structure):
    _fields_ = [("destinationPartials", C.c_int), ("destinationScaleWrite", C.c_int),
                ("destinationScaleRead", C.c_int), ("child1Partials", C.c_int),
                ("child1TransitionMatrix", C.c_int), ("child2Partials", C.c_int),
                ("child2TransitionMatrix", C.c_int)]


EXPORTS = [
    "beagleGetVersion", "beagleGetCitation", "beagleGetResourceList", "beagleCreateInstance",
    "beagleFinalizeInstance", "beagleFinalize", "beagleSetTipStates", "beagleSetTipPartials", "beagleSetPartials",
    "beagleGetPartials", "beagleSetEigenDecomposition", "beagleSetStateFrequencies", "beagleSetCategoryWeights",
    "beagleSetCategoryRates", "beagleSetPatternWeights", "beagleUpdateTransitionMatrices",
    "beagleSetTransitionMatrix", "beagleGetTransitionMatrix", "beagleUpdatePartials", "beagleWaitForPartials",
    "beagleAccumulateScaleFactors", "beagleRemoveScaleFactors", "beagleResetScaleFactors", "beagleCopyScaleFactors",
    "beagleGetScaleFactors", "beagleCalculateRootLogLikelihoods", "beagleCalculateEdgeLogLikelihoods",
    "beagleGetSiteLogLikelihoods", "mbamdSynchronize", "mbamdGetLastError", "mbamdKernelTiming",
    "mbamdGetKernelTiming", "mbamdGetListCounts", "mbamdGetStepTiming", "mbamdUpdateFinalPartials", "mbamdGetScaledPartials", "mbamdSetKernelPath", "mbamdSetDeferredResult", "mbamdFetchLogLikelihood", "mbamdReduceLogLikelihood", "mbamdGetResourcePciBusId", "mbamdGetInstanceDevices",
    "mbamdGetScaleExponents", "mbamdGetChildCount", "mbamdSetRateMatrices", "mbamdSetRateMatricesFrom",
    # BEAGLE v3 surface (multi-partition instances, resource benchmark)
    "beagleGetBenchmarkedResourceList", "beagleSetCPUThreadCount", "beagleSetPatternPartitions",
    "beagleSetCategoryRatesWithIndex", "beagleUpdateTransitionMatricesWithMultipleModels", "beagleUpdatePartialsByPartition",
    "beagleAccumulateScaleFactorsByPartition", "beagleRemoveScaleFactorsByPartition", "beagleResetScaleFactorsByPartition",
    "beagleCalculateRootLogLikelihoodsByPartition", "beagleCalculateEdgeLogLikelihoodsByPartition",
]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class BeagleError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        super().__init__("%s failed with code %d%s" % (where, code, (": " + detail) if detail else ""))
        self.code = code


def _d(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


class BeagleLibrary:
    """The shared library, with argument types declared."""

    def __init__(self, path: Optional[str] = None):
        path = path or os.environ.get("MBAMD_LIBRARY") or DEFAULT_LIB
        if not os.path.exists(path):
            raise FileNotFoundError(
                "%s not found: build the HIP engine first (python -m mrbayes_amd.build). "
                "There is no CPU implementation to fall back to." % path)
        self.path = path
        self.lib = C.CDLL(path)
        for name in EXPORTS:
            if not hasattr(self.lib, name):
                raise AttributeError("%s does not export %s" % (path, name))
        L = self.lib
        L.beagleGetVersion.restype = C.c_char_p
        L.beagleGetCitation.restype = C.c_char_p
        L.mbamdGetLastError.restype = C.c_char_p
        L.beagleGetResourceList.restype = C.POINTER(BeagleResourceList)
        L.beagleCreateInstance.argtypes = [C.c_int] * 9 + [_ip, C.c_int, C.c_long, C.c_long,
                                                          C.POINTER(BeagleInstanceDetails)]
        L.beagleSetTipStates.argtypes = [C.c_int, C.c_int, _ip]
        L.beagleSetTipPartials.argtypes = [C.c_int, C.c_int, _dp]
        L.beagleSetPartials.argtypes = [C.c_int, C.c_int, _dp]
        L.beagleGetPartials.argtypes = [C.c_int, C.c_int, C.c_int, _dp]
        L.beagleSetEigenDecomposition.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
        L.beagleSetStateFrequencies.argtypes = [C.c_int, C.c_int, _dp]
        L.beagleSetCategoryWeights.argtypes = [C.c_int, C.c_int, _dp]
        L.beagleSetCategoryRates.argtypes = [C.c_int, _dp]
        L.beagleSetPatternWeights.argtypes = [C.c_int, _dp]
        L.beagleUpdateTransitionMatrices.argtypes = [C.c_int, C.c_int, _ip, _ip, _ip, _dp, C.c_int]
        L.beagleSetTransitionMatrix.argtypes = [C.c_int, C.c_int, _dp, C.c_double]
        L.beagleGetTransitionMatrix.argtypes = [C.c_int, C.c_int, _dp]
        L.beagleUpdatePartials.argtypes = [C.c_int, C.POINTER(BeagleOperation), C.c_int, C.c_int]
        L.beagleWaitForPartials.argtypes = [C.c_int, _ip, C.c_int]
        L.beagleAccumulateScaleFactors.argtypes = [C.c_int, _ip, C.c_int, C.c_int]
        L.beagleRemoveScaleFactors.argtypes = [C.c_int, _ip, C.c_int, C.c_int]
        L.beagleResetScaleFactors.argtypes = [C.c_int, C.c_int]
        L.beagleCopyScaleFactors.argtypes = [C.c_int, C.c_int, C.c_int]
        L.beagleGetScaleFactors.argtypes = [C.c_int, C.c_int, _dp]
        L.beagleCalculateRootLogLikelihoods.argtypes = [C.c_int, _ip, _ip, _ip, _ip, C.c_int, _dp]
        L.beagleCalculateEdgeLogLikelihoods.argtypes = [C.c_int, _ip, _ip, _ip, _ip, _ip, _ip, _ip, _ip, C.c_int,
                                                        _dp, _dp, _dp]
        L.beagleGetSiteLogLikelihoods.argtypes = [C.c_int, _dp]
        L.mbamdGetKernelTiming.argtypes = [C.c_int, _dp, C.POINTER(C.c_long), C.c_int]
        L.mbamdGetListCounts.argtypes = [C.c_int, C.POINTER(C.c_long)]
        L.mbamdGetStepTiming.argtypes = [C.c_int, _dp, C.POINTER(C.c_long), C.c_int]
        L.mbamdUpdateFinalPartials.argtypes = [C.c_int, C.c_void_p, C.c_int]
        L.mbamdGetScaledPartials.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.mbamdFetchLogLikelihood.argtypes = [C.c_int, _dp]
        L.mbamdReduceLogLikelihood.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.mbamdGetResourcePciBusId.argtypes = [C.c_int, C.c_char_p, C.c_int]
        L.mbamdGetInstanceDevices.argtypes = [C.c_int, _ip, C.c_int]

    def version(self) -> str:
        return self.lib.beagleGetVersion().decode()

    def last_error(self) -> str:
        return self.lib.mbamdGetLastError().decode()

    def pci_bus_id(self, resource: int) -> str:
        """PCI bus id of a resource ("0000:c1:00.0"): which physical GPU a rank / shard runs on."""
        buf = C.create_string_buffer(64)
        rc = self.lib.mbamdGetResourcePciBusId(resource, buf, 64)
        if rc != 0:
            raise BeagleError(rc, "mbamdGetResourcePciBusId", self.last_error())
        return buf.value.decode()

    def resources(self):
        rl = self.lib.beagleGetResourceList().contents
        return [(rl.list[i].name.decode(), rl.list[i].description.decode(), rl.list[i].supportFlags)
                for i in range(rl.length)]


_default: Optional[BeagleLibrary] = None


def library(path: Optional[str] = None) -> BeagleLibrary:
    global _default
    if path is not None:
        return BeagleLibrary(path)
    if _default is None:
        _default = BeagleLibrary()
    return _default


class BeagleInstance:
    """One engine instance; methods are the beagle* functions minus the instance argument."""

    def __init__(self, lib: BeagleLibrary, tip_count, partials_buffer_count, compact_buffer_count, state_count,
                 pattern_count, eigen_buffer_count, matrix_buffer_count, category_count, scale_buffer_count,
                 resource: Optional[int] = None, preference_flags=0, requirement_flags=0):
        self._L = lib
        self.lib = lib.lib
        self.details = BeagleInstanceDetails()
        rl = None
        rc = 0
        if resource is not None:
            arr = (C.c_int * 1)(resource)
            rl, rc = C.cast(arr, _ip), 1
        self.id = self.lib.beagleCreateInstance(tip_count, partials_buffer_count, compact_buffer_count, state_count,
                                                pattern_count, eigen_buffer_count, matrix_buffer_count,
                                                category_count, scale_buffer_count, rl, rc, preference_flags,
                                                requirement_flags, C.byref(self.details))
        if self.id < 0:
            raise BeagleError(self.id, "beagleCreateInstance", lib.last_error())
        self.state_count, self.pattern_count, self.category_count = state_count, pattern_count, category_count

    def _chk(self, code: int, where: str, allow=()):
        if code != BEAGLE_SUCCESS and code not in allow:
            raise BeagleError(code, where, self._L.last_error())
        return code

    def finalize(self):
        if self.id >= 0:
            self.lib.beagleFinalizeInstance(self.id)
            self.id = -1

    def __del__(self):
        try:
            self.finalize()
        except Exception:
            pass

    # ---- data -----------------------------------------------------------------------------------
    def set_tip_states(self, tip, states):
        a = _i(states)
        self._chk(self.lib.beagleSetTipStates(self.id, tip, a.ctypes.data_as(_ip)), "beagleSetTipStates")

    def set_tip_partials(self, tip, partials):
        a = _d(partials)
        self._chk(self.lib.beagleSetTipPartials(self.id, tip, a.ctypes.data_as(_dp)), "beagleSetTipPartials")

    def set_partials(self, idx, partials):
        a = _d(partials)
        self._chk(self.lib.beagleSetPartials(self.id, idx, a.ctypes.data_as(_dp)), "beagleSetPartials")

    def get_partials(self, idx) -> np.ndarray:
        out = np.empty((self.category_count, self.pattern_count, self.state_count))
        self._chk(self.lib.beagleGetPartials(self.id, idx, BEAGLE_OP_NONE, out.ctypes.data_as(_dp)), "beagleGetPartials")
        return out

    def set_eigen_decomposition(self, idx, evec, ivec, evals):
        a, b, c = _d(evec), _d(ivec), _d(evals)
        self._chk(self.lib.beagleSetEigenDecomposition(self.id, idx, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp),
                                                       c.ctypes.data_as(_dp)), "beagleSetEigenDecomposition")

    def set_rate_matrices(self, first, qs, pi, exchangeabilities=False):
        """Extension: eigen-systems of reversible rate matrices computed on the device (mbamdSetRateMatrices)."""
        a, b = _d(np.asarray(qs, dtype=np.float64)), _d(pi)
        n = a.size // (len(b) * len(b))
        self.lib.mbamdSetRateMatrices.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int]
        self._chk(self.lib.mbamdSetRateMatrices(self.id, first, n, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp),
                                                1 if exchangeabilities else 0), "mbamdSetRateMatrices")

    def set_rate_matrices_from(self, first, qs, pi, warm_first, exchangeabilities=False, shield=False):
        """Extension: the same, warm-started from the eigenvectors of eigen buffers warm_first ... (mbamdSetRateMatricesFrom)."""
        a, b = _d(np.asarray(qs, dtype=np.float64)), _d(pi)
        n = a.size // (len(b) * len(b))
        self.lib.mbamdSetRateMatricesFrom.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int]
        self._chk(self.lib.mbamdSetRateMatricesFrom(self.id, first, n, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp),
                                                    (1 if exchangeabilities else 0) | (2 if shield else 0), warm_first), "mbamdSetRateMatricesFrom")

    def set_state_frequencies(self, idx, f):
        a = _d(f)
        self._chk(self.lib.beagleSetStateFrequencies(self.id, idx, a.ctypes.data_as(_dp)), "beagleSetStateFrequencies")

    def set_category_weights(self, idx, w):
        a = _d(w)
        self._chk(self.lib.beagleSetCategoryWeights(self.id, idx, a.ctypes.data_as(_dp)), "beagleSetCategoryWeights")

    def set_category_rates(self, r):
        a = _d(r)
        self._chk(self.lib.beagleSetCategoryRates(self.id, a.ctypes.data_as(_dp)), "beagleSetCategoryRates")

    def set_pattern_weights(self, w):
        a = _d(w)
        self._chk(self.lib.beagleSetPatternWeights(self.id, a.ctypes.data_as(_dp)), "beagleSetPatternWeights")

    # ---- matrices -------------------------------------------------------------------------------
    def update_transition_matrices(self, eigen_index, prob_indices, edge_lengths):
        p, e = _i(prob_indices), _d(edge_lengths)
        self._chk(self.lib.beagleUpdateTransitionMatrices(self.id, eigen_index, p.ctypes.data_as(_ip), None, None,
                                                          e.ctypes.data_as(_dp), len(p)),
                  "beagleUpdateTransitionMatrices")

    def set_transition_matrix(self, idx, m):
        a = _d(m)
        self._chk(self.lib.beagleSetTransitionMatrix(self.id, idx, a.ctypes.data_as(_dp), 0.0), "beagleSetTransitionMatrix")

    def get_transition_matrix(self, idx) -> np.ndarray:
        out = np.empty((self.category_count, self.state_count, self.state_count))
        self._chk(self.lib.beagleGetTransitionMatrix(self.id, idx, out.ctypes.data_as(_dp)), "beagleGetTransitionMatrix")
        return out

    # ---- partials / scaling -----------------------------------------------------------------------
    def update_partials(self, operations, cumulative_scale_index=BEAGLE_OP_NONE):
        """operations: int array [n][7] in BeagleOperation field order, or a ctypes array."""
        if isinstance(operations, np.ndarray) or isinstance(operations, (list, tuple)):
            a = _i(operations).reshape(-1, 7)
            ptr = a.ctypes.data_as(C.POINTER(BeagleOperation))   # keeps `a` alive
            n = a.shape[0]
        else:
            ptr, n = operations, len(operations)
        self._chk(self.lib.beagleUpdatePartials(self.id, ptr, n, cumulative_scale_index), "beagleUpdatePartials")

    def wait_for_partials(self):
        self._chk(self.lib.beagleWaitForPartials(self.id, None, 0), "beagleWaitForPartials")

    def accumulate_scale_factors(self, indices, cum):
        a = _i(indices)
        self._chk(self.lib.beagleAccumulateScaleFactors(self.id, a.ctypes.data_as(_ip), len(a), cum), "beagleAccumulateScaleFactors")

    def remove_scale_factors(self, indices, cum):
        a = _i(indices)
        self._chk(self.lib.beagleRemoveScaleFactors(self.id, a.ctypes.data_as(_ip), len(a), cum), "beagleRemoveScaleFactors")

    def reset_scale_factors(self, cum):
        self._chk(self.lib.beagleResetScaleFactors(self.id, cum), "beagleResetScaleFactors")

    def copy_scale_factors(self, dst, src):
        self._chk(self.lib.beagleCopyScaleFactors(self.id, dst, src), "beagleCopyScaleFactors")

    def get_scale_factors(self, idx) -> np.ndarray:
        out = np.empty(self.pattern_count)
        self._chk(self.lib.beagleGetScaleFactors(self.id, idx, out.ctypes.data_as(_dp)), "beagleGetScaleFactors")
        return out

    def get_scale_exponents(self, idx) -> np.ndarray:
        """Engine extension: the binary exponents behind a scale buffer, [categories][patterns]."""
        out = np.empty((self.category_count, self.pattern_count), dtype=np.int32)
        self._chk(self.lib.mbamdGetScaleExponents(self.id, idx, out.ctypes.data_as(_ip)), "mbamdGetScaleExponents")
        return out

    # ---- likelihood -------------------------------------------------------------------------------
    def calculate_root_log_likelihoods(self, buffers, weights, freqs, cums):
        """Returns (return code, lnL); BEAGLE_ERROR_FLOATING_POINT is passed through, as MrBayes expects."""
        b, w, f, c = _i(buffers), _i(weights), _i(freqs), _i(cums)
        out = C.c_double(0.0)
        rc = self.lib.beagleCalculateRootLogLikelihoods(self.id, b.ctypes.data_as(_ip), w.ctypes.data_as(_ip),
                                                        f.ctypes.data_as(_ip), c.ctypes.data_as(_ip), len(b), C.byref(out))
        self._chk(rc, "beagleCalculateRootLogLikelihoods", allow=(BEAGLE_ERROR_FLOATING_POINT,))
        return rc, out.value

    def calculate_edge_log_likelihoods(self, parents, children, probs, weights, freqs, cums):
        p, ch, pr, w, f, c = _i(parents), _i(children), _i(probs), _i(weights), _i(freqs), _i(cums)
        out = C.c_double(0.0)
        rc = self.lib.beagleCalculateEdgeLogLikelihoods(self.id, p.ctypes.data_as(_ip), ch.ctypes.data_as(_ip),
                                                        pr.ctypes.data_as(_ip), None, None, w.ctypes.data_as(_ip),
                                                        f.ctypes.data_as(_ip), c.ctypes.data_as(_ip), len(p),
                                                        C.byref(out), None, None)
        self._chk(rc, "beagleCalculateEdgeLogLikelihoods", allow=(BEAGLE_ERROR_FLOATING_POINT,))
        return rc, out.value

    # ---- BEAGLE v3: multi-partition instances (reference src/mbbeagle.c:1500-3010) -------------------------------
    def child_count(self) -> int:
        return int(self.lib.mbamdGetChildCount(self.id))

    def set_pattern_partitions(self, partition_count, pattern_partitions):
        a = _i(pattern_partitions)
        self._chk(self.lib.beagleSetPatternPartitions(self.id, partition_count, a.ctypes.data_as(_ip)), "beagleSetPatternPartitions")

    def set_category_rates_with_index(self, index, rates):
        a = _d(rates)
        self._chk(self.lib.beagleSetCategoryRatesWithIndex(self.id, index, a.ctypes.data_as(_dp)), "beagleSetCategoryRatesWithIndex")

    def update_transition_matrices_with_multiple_models(self, eigen_indices, rate_indices, prob_indices, edge_lengths):
        g, r, p, e = _i(eigen_indices), _i(rate_indices), _i(prob_indices), _d(edge_lengths)
        self._chk(self.lib.beagleUpdateTransitionMatricesWithMultipleModels(
            self.id, g.ctypes.data_as(_ip), r.ctypes.data_as(_ip), p.ctypes.data_as(_ip), None, None, e.ctypes.data_as(_dp), len(p)),
            "beagleUpdateTransitionMatricesWithMultipleModels")

    def update_partials_by_partition(self, operations):
        """operations: int array [n][9] in BeagleOperationByPartition field order."""
        a = _i(operations).reshape(-1, 9)
        self._chk(self.lib.beagleUpdatePartialsByPartition(self.id, a.ctypes.data_as(C.c_void_p), a.shape[0]),
                  "beagleUpdatePartialsByPartition")

    def reset_scale_factors_by_partition(self, cum, partition):
        self._chk(self.lib.beagleResetScaleFactorsByPartition(self.id, cum, partition), "beagleResetScaleFactorsByPartition")

    def accumulate_scale_factors_by_partition(self, idx, cum, partition):
        a = _i(idx)
        self._chk(self.lib.beagleAccumulateScaleFactorsByPartition(self.id, a.ctypes.data_as(_ip), len(a), cum, partition),
                  "beagleAccumulateScaleFactorsByPartition")

    def remove_scale_factors_by_partition(self, idx, cum, partition):
        a = _i(idx)
        self._chk(self.lib.beagleRemoveScaleFactorsByPartition(self.id, a.ctypes.data_as(_ip), len(a), cum, partition),
                  "beagleRemoveScaleFactorsByPartition")

    def calculate_edge_log_likelihoods_by_partition(self, parents, children, probs, weights, freqs, cums, partitions, count):
        """index arrays laid out [count][len(partitions)]; returns (rc, per-partition sums, total)"""
        p, ch, pr, w, f, c, pt = _i(parents), _i(children), _i(probs), _i(weights), _i(freqs), _i(cums), _i(partitions)
        by = np.zeros(len(pt))
        out = C.c_double(0.0)
        rc = self.lib.beagleCalculateEdgeLogLikelihoodsByPartition(
            self.id, p.ctypes.data_as(_ip), ch.ctypes.data_as(_ip), pr.ctypes.data_as(_ip), None, None, w.ctypes.data_as(_ip),
            f.ctypes.data_as(_ip), c.ctypes.data_as(_ip), pt.ctypes.data_as(_ip), len(pt), count, by.ctypes.data_as(_dp),
            C.byref(out), None, None, None, None)
        self._chk(rc, "beagleCalculateEdgeLogLikelihoodsByPartition", allow=(BEAGLE_ERROR_FLOATING_POINT,))
        return rc, by, out.value

    def calculate_root_log_likelihoods_by_partition(self, buffers, weights, freqs, cums, partitions, count):
        b, w, f, c, pt = _i(buffers), _i(weights), _i(freqs), _i(cums), _i(partitions)
        by = np.zeros(len(pt))
        out = C.c_double(0.0)
        rc = self.lib.beagleCalculateRootLogLikelihoodsByPartition(
            self.id, b.ctypes.data_as(_ip), w.ctypes.data_as(_ip), f.ctypes.data_as(_ip), c.ctypes.data_as(_ip),
            pt.ctypes.data_as(_ip), len(pt), count, by.ctypes.data_as(_dp), C.byref(out))
        self._chk(rc, "beagleCalculateRootLogLikelihoodsByPartition", allow=(BEAGLE_ERROR_FLOATING_POINT,))
        return rc, by, out.value

    def get_site_log_likelihoods(self) -> np.ndarray:
        out = np.empty(self.pattern_count)
        self._chk(self.lib.beagleGetSiteLogLikelihoods(self.id, out.ctypes.data_as(_dp)), "beagleGetSiteLogLikelihoods")
        return out

    # ---- engine extensions --------------------------------------------------------------------------
    def synchronize(self):
        self._chk(self.lib.mbamdSynchronize(self.id), "mbamdSynchronize")

    def kernel_timing(self, enable: bool):
        self._chk(self.lib.mbamdKernelTiming(self.id, 1 if enable else 0), "mbamdKernelTiming")

    def get_kernel_timing(self, reset=True):
        ms, n = C.c_double(0.0), C.c_long(0)
        self._chk(self.lib.mbamdGetKernelTiming(self.id, C.byref(ms), C.byref(n), 1 if reset else 0), "mbamdGetKernelTiming")
        return ms.value, n.value

    def get_list_counts(self):
        """(lists, paths, forked paths, paths fused with their log-likelihood, tree walks, their operations) of the 4-state lists."""
        out = (C.c_long * 6)()
        self._chk(self.lib.mbamdGetListCounts(self.id, out), "mbamdGetListCounts")
        return tuple(int(v) for v in out)

    # ---- reports (include/libhmsbeagle/mbamd_reports.h) --------------------------------------------
    def update_final_partials(self, operations):
        """operations: int array [n][5] in MbamdFinalOperation field order
        (destinationPartials, ancestorFinal, downPartials, transitionMatrix, rootTip)."""
        a = _i(operations).reshape(-1, 5)
        self._chk(self.lib.mbamdUpdateFinalPartials(self.id, a.ctypes.data_as(C.c_void_p), a.shape[0]), "mbamdUpdateFinalPartials")

    def get_scaled_partials(self, idx, cumulative_scale_index=BEAGLE_OP_NONE):
        """-> (partials float32 [K][P][S] with the categories of a pattern at one scale, lnScale float32 [P])."""
        out = np.empty((self.category_count, self.pattern_count, self.state_count), dtype=np.float32)
        ln = np.empty(self.pattern_count, dtype=np.float32)
        self._chk(self.lib.mbamdGetScaledPartials(self.id, idx, cumulative_scale_index, out.ctypes.data_as(C.POINTER(C.c_float)),
                                                  ln.ctypes.data_as(C.POINTER(C.c_float))), "mbamdGetScaledPartials")
        return out, ln

    def get_step_timing(self, reset=True):
        """(ms, spans): device time of whole evaluations (all kernels of a step and the gaps between them)."""
        ms, n = C.c_double(0.0), C.c_long(0)
        self._chk(self.lib.mbamdGetStepTiming(self.id, C.byref(ms), C.byref(n), 1 if reset else 0), "mbamdGetStepTiming")
        return ms.value, n.value

    def set_deferred_result(self, enable: bool):
        self._chk(self.lib.mbamdSetDeferredResult(self.id, 1 if enable else 0), "mbamdSetDeferredResult")

    def reduce_log_likelihood(self, device_ptr, waiting_stream=0):
        """The pending sum added up on the device into the double at `device_ptr`; `waiting_stream` (raw hipStream_t) waits for it."""
        self._chk(self.lib.mbamdReduceLogLikelihood(self.id, C.c_void_p(device_ptr), C.c_void_p(waiting_stream)), "mbamdReduceLogLikelihood")

    def devices(self):
        """Resource numbers of the engine(s) behind this instance (one per shard of a sharded instance)."""
        out = (C.c_int * 64)()
        n = self.lib.mbamdGetInstanceDevices(self.id, C.cast(out, _ip), 64)
        return [int(out[i]) for i in range(min(n, 64))]

    def fetch_log_likelihood(self):
        out = C.c_double(0.0)
        rc = self.lib.mbamdFetchLogLikelihood(self.id, C.byref(out))
        self._chk(rc, "mbamdFetchLogLikelihood", allow=(BEAGLE_ERROR_FLOATING_POINT,))
        return rc, out.value
