/*
 * libhmsbeagle/beagle.h -- C ABI of the MI355X-native conditional-likelihood engine.
 *
 * This is the drop-in boundary: MrBayes' own adapter, reference
 * src/mbbeagle.c + src/likelihood.c + src/mcmc.c, includes "libhmsbeagle/beagle.h"
 * (reference src/bayes.h:62-64) and calls the functions declared here; building the
 * unmodified reference with -DBEAGLE_ENABLED against this header and linking
 * mrbayes_amd/libhmsbeagle.so makes `set usebeagle=yes` run on the HIP engine.
 *
 * The upstream header (beagle-dev/beagle-lib, "release 3.1.2 ... also 2.1.3", reference
 * INSTALL:199-201) is not vendored in the reference tree, so this file is authored from the
 * reference's call sites (cited per declaration) and the published BEAGLE API (Ayres et al.
 * 2012/2019); numeric constant values follow the published API so existing clients keep working.
 *
 * Every pointer argument is caller-owned host memory, read (or written) during the call; no
 * device or framework types appear in any signature.  All calls on one instance must come from
 * one host thread at a time (MrBayes is single-threaded, reference src/mcmc.c:16718).
 *
 * Conventions:  partials [category][pattern][state], transition matrices [category][from][to]
 * (row-major), eigenvectors row-major S x S, all double at the boundary; single precision inside.
 */
#ifndef MBAMD_LIBHMSBEAGLE_BEAGLE_H_
#define MBAMD_LIBHMSBEAGLE_BEAGLE_H_

#ifdef __cplusplus
extern "C" {
#endif

#define BEAGLE_DLLEXPORT __attribute__((visibility("default")))

/* return codes (reference use: src/mbbeagle.c:471,1283,1291) */
enum BeagleReturnCodes {
    BEAGLE_SUCCESS                      =  0,
    BEAGLE_ERROR_GENERAL                = -1,
    BEAGLE_ERROR_OUT_OF_MEMORY          = -2,
    BEAGLE_ERROR_UNIDENTIFIED_EXCEPTION = -3,
    BEAGLE_ERROR_UNINITIALIZED_INSTANCE = -4,
    BEAGLE_ERROR_OUT_OF_RANGE           = -5,
    BEAGLE_ERROR_NO_RESOURCE            = -6,
    BEAGLE_ERROR_NO_IMPLEMENTATION      = -7,
    BEAGLE_ERROR_FLOATING_POINT         = -8
};

/* capability / preference / requirement flags (reference use: src/mbbeagle.c:685-753,
 * src/command.c:6637-7090, src/bayes.c:608-640) */
enum BeagleFlags {
    BEAGLE_FLAG_PRECISION_SINGLE    = 1 << 0,   /* the tuned engines (fp32 conditional likelihoods, fp64 sums and logs) */
    BEAGLE_FLAG_PRECISION_DOUBLE    = 1 << 1,   /* required, or preferred without SINGLE (`set beagleprecision=double`): the fp64 engine */
    BEAGLE_FLAG_COMPUTATION_SYNCH   = 1 << 2,
    BEAGLE_FLAG_COMPUTATION_ASYNCH  = 1 << 3,
    BEAGLE_FLAG_EIGEN_REAL          = 1 << 4,
    BEAGLE_FLAG_EIGEN_COMPLEX       = 1 << 5,
    BEAGLE_FLAG_SCALING_MANUAL      = 1 << 6,
    BEAGLE_FLAG_SCALING_AUTO        = 1 << 7,
    BEAGLE_FLAG_SCALING_ALWAYS      = 1 << 8,
    BEAGLE_FLAG_SCALERS_RAW         = 1 << 9,
    BEAGLE_FLAG_SCALERS_LOG         = 1 << 10,
    BEAGLE_FLAG_VECTOR_SSE          = 1 << 11,
    BEAGLE_FLAG_VECTOR_NONE         = 1 << 12,
    BEAGLE_FLAG_THREADING_OPENMP    = 1 << 13,
    BEAGLE_FLAG_THREADING_NONE      = 1 << 14,
    BEAGLE_FLAG_PROCESSOR_CPU       = 1 << 15,
    BEAGLE_FLAG_PROCESSOR_GPU       = 1 << 16,
    BEAGLE_FLAG_PROCESSOR_FPGA      = 1 << 17,
    BEAGLE_FLAG_PROCESSOR_CELL      = 1 << 18,
    BEAGLE_FLAG_PROCESSOR_PHI       = 1 << 19,
    BEAGLE_FLAG_INVEVEC_STANDARD    = 1 << 20,
    BEAGLE_FLAG_INVEVEC_TRANSPOSED  = 1 << 21,
    BEAGLE_FLAG_FRAMEWORK_CUDA      = 1 << 22,
    BEAGLE_FLAG_FRAMEWORK_OPENCL    = 1 << 23,
    BEAGLE_FLAG_VECTOR_AVX          = 1 << 24,
    BEAGLE_FLAG_SCALING_DYNAMIC     = 1 << 25,
    BEAGLE_FLAG_PROCESSOR_OTHER     = 1 << 26,
    BEAGLE_FLAG_FRAMEWORK_CPU       = 1 << 27,
    BEAGLE_FLAG_PARALLELOPS_STREAMS = 1 << 28,
    BEAGLE_FLAG_PARALLELOPS_GRID    = 1 << 29,
    BEAGLE_FLAG_THREADING_CPP       = 1 << 30
};
/* this engine: native HIP on AMD CDNA; reported in BeagleInstanceDetails.flags next to
   PROCESSOR_GPU.  Deliberately NOT FRAMEWORK_CUDA/OPENCL (those switch MrBayes' v3 build to
   level-order operation lists, reference src/mbbeagle.c:363-381; the engine levels ops itself). */
#define BEAGLE_FLAG_FRAMEWORK_HIP (1L << 31)

/* benchmarking hints of the v3 resource benchmark (reference src/mbbeagle.c:220-228) */
enum BeagleBenchmarkFlags {
    BEAGLE_BENCHFLAG_SCALING_NONE    = 1 << 0,
    BEAGLE_BENCHFLAG_SCALING_ALWAYS  = 1 << 1,
    BEAGLE_BENCHFLAG_SCALING_DYNAMIC = 1 << 2
};

enum BeagleOpCodes {
    BEAGLE_OP_COUNT           = 7,   /* ints per BeagleOperation */
    BEAGLE_PARTITION_OP_COUNT = 9,   /* ints per BeagleOperationByPartition */
    BEAGLE_OP_NONE            = -1   /* "no buffer" sentinel (reference src/mbbeagle.c:834-860) */
};

/* filled by beagleCreateInstance (reference src/mbbeagle.c:178, 329-356) */
typedef struct {
    int   resourceNumber;
    char* resourceName;
    char* implName;
    char* implDescription;
    long  flags;
} BeagleInstanceDetails;

/* one entry of beagleGetResourceList (reference src/mbbeagle.c:609-660) */
typedef struct {
    char* name;
    char* description;
    long  supportFlags;
    long  requiredFlags;
} BeagleResource;

typedef struct {
    BeagleResource* list;
    int             length;
} BeagleResourceList;

/* v3 benchmark list (reference src/mbbeagle.c:231-307) */
typedef struct {
    int    number;
    char*  name;
    char*  description;
    long   supportFlags;
    long   requiredFlags;
    int    returnCode;
    char*  implName;
    long   benchedFlags;
    double benchmarkResult;
    double performanceRatio;
} BeagleBenchmarkedResource;

typedef struct {
    BeagleBenchmarkedResource* list;
    int                        length;
} BeagleBenchmarkedResourceList;

/* one partial-likelihood update (reference src/mbbeagle.c:817-846, 919-947, 1042-1075):
 *   dest[k,c,i] = (sum_j P1_k[i,j] child1[k,c,j]) * (sum_j P2_k[i,j] child2[k,c,j])
 * destinationScaleWrite >= 0: rescale dest per pattern, store the factors in that scale buffer and,
 *   if updatePartials' cumulativeScaleIndex >= 0, add their logs to that cumulative buffer;
 * destinationScaleRead  >= 0: divide dest by the factors already stored in that buffer. */
typedef struct {
    int destinationPartials;
    int destinationScaleWrite;
    int destinationScaleRead;
    int child1Partials;
    int child1TransitionMatrix;
    int child2Partials;
    int child2TransitionMatrix;
} BeagleOperation;

/* v3 multi-partition form (reference src/mbbeagle.c:2292) */
typedef struct {
    int destinationPartials;
    int destinationScaleWrite;
    int destinationScaleRead;
    int child1Partials;
    int child1TransitionMatrix;
    int child2Partials;
    int child2TransitionMatrix;
    int partition;
    int cumulativeScaleIndex;
} BeagleOperationByPartition;

/* ---------------------------------------------------------------------------------------------
 * library-level queries
 * ------------------------------------------------------------------------------------------- */
/* reference src/mbbeagle.c:345,350,659 */
BEAGLE_DLLEXPORT const char* beagleGetVersion(void);
BEAGLE_DLLEXPORT const char* beagleGetCitation(void);
/* reference src/mbbeagle.c:614,645: one resource per visible AMD GPU (supportFlags has
 * PROCESSOR_GPU, which `set beagledevice=gpu` filters on, src/mbbeagle.c:620) */
BEAGLE_DLLEXPORT BeagleResourceList* beagleGetResourceList(void);

/* ---------------------------------------------------------------------------------------------
 * instance life cycle
 * ------------------------------------------------------------------------------------------- */
/* reference src/mbbeagle.c:313-326.  Returns instance id >= 0 or a negative BeagleReturnCodes.
 * resourceList == NULL / resourceCount == 0 lets the library choose the device
 * (src/mbbeagle.c:322-323). */
BEAGLE_DLLEXPORT int beagleCreateInstance(int tipCount, int partialsBufferCount, int compactBufferCount,
                                          int stateCount, int patternCount, int eigenBufferCount,
                                          int matrixBufferCount, int categoryCount, int scaleBufferCount,
                                          int* resourceList, int resourceCount,
                                          long preferenceFlags, long requirementFlags,
                                          BeagleInstanceDetails* returnInfo);
/* reference src/mbbeagle.c:340, src/mcmc.c:4602 */
BEAGLE_DLLEXPORT int beagleFinalizeInstance(int instance);
BEAGLE_DLLEXPORT int beagleFinalize(void);

/* ---------------------------------------------------------------------------------------------
 * data upload
 * ------------------------------------------------------------------------------------------- */
/* reference src/mbbeagle.c:148: patternCount ints, value >= stateCount = missing/gap */
BEAGLE_DLLEXPORT int beagleSetTipStates(int instance, int tipIndex, const int* inStates);
/* reference src/mbbeagle.c:165: [pattern][state] doubles, replicated over categories */
BEAGLE_DLLEXPORT int beagleSetTipPartials(int instance, int tipIndex, const double* inPartials);
/* [category][pattern][state] doubles */
BEAGLE_DLLEXPORT int beagleSetPartials(int instance, int bufferIndex, const double* inPartials);
/* read a partials buffer back, [category][pattern][state]; scaleIndex must be BEAGLE_OP_NONE */
BEAGLE_DLLEXPORT int beagleGetPartials(int instance, int bufferIndex, int scaleIndex, double* outPartials);
/* reference src/likelihood.c:10652,10752: row-major U, U^-1 (S x S) and S real eigenvalues */
BEAGLE_DLLEXPORT int beagleSetEigenDecomposition(int instance, int eigenIndex, const double* inEigenVectors,
                                                 const double* inInverseEigenVectors, const double* inEigenValues);
/* reference src/mbbeagle.c:1179 */
BEAGLE_DLLEXPORT int beagleSetStateFrequencies(int instance, int stateFrequenciesIndex, const double* inStateFrequencies);
/* reference src/mbbeagle.c:1201,1210; src/mcmc.c:6289,6293 */
BEAGLE_DLLEXPORT int beagleSetCategoryWeights(int instance, int categoryWeightsIndex, const double* inCategoryWeights);
/* reference src/mbbeagle.c:1409 */
BEAGLE_DLLEXPORT int beagleSetCategoryRates(int instance, const double* inCategoryRates);
/* reference src/mcmc.c:6264, src/mbbeagle.c:1225 */
BEAGLE_DLLEXPORT int beagleSetPatternWeights(int instance, const double* inPatternWeights);

/* ---------------------------------------------------------------------------------------------
 * transition matrices
 * ------------------------------------------------------------------------------------------- */
/* reference src/mbbeagle.c:1477-1483: P_k = U diag(exp(lambda * rate_k * t)) U^-1 for each listed
 * branch, negatives clamped to 0 like the native path (src/likelihood.c:9540).  Derivative index
 * lists must be NULL (MrBayes always passes NULL). */
BEAGLE_DLLEXPORT int beagleUpdateTransitionMatrices(int instance, int eigenIndex, const int* probabilityIndices,
                                                    const int* firstDerivativeIndices,
                                                    const int* secondDerivativeIndices,
                                                    const double* edgeLengths, int count);
/* [category][from][to] doubles; paddedValue is ignored */
BEAGLE_DLLEXPORT int beagleSetTransitionMatrix(int instance, int matrixIndex, const double* inMatrix, double paddedValue);
BEAGLE_DLLEXPORT int beagleGetTransitionMatrix(int instance, int matrixIndex, double* outMatrix);

/* ---------------------------------------------------------------------------------------------
 * partial likelihoods and scale factors
 * ------------------------------------------------------------------------------------------- */
/* reference src/mbbeagle.c:868, 978, 1100: operations are in dependency order (post-order) */
BEAGLE_DLLEXPORT int beagleUpdatePartials(int instance, const BeagleOperation* operations, int operationCount,
                                          int cumulativeScaleIndex);
BEAGLE_DLLEXPORT int beagleWaitForPartials(int instance, const int* destinationPartials, int destinationPartialsCount);
/* reference src/likelihood.c:8088 / src/mbbeagle.c:1094 / src/mbbeagle.c:419,512,567 */
BEAGLE_DLLEXPORT int beagleAccumulateScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex);
BEAGLE_DLLEXPORT int beagleRemoveScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex);
BEAGLE_DLLEXPORT int beagleResetScaleFactors(int instance, int cumulativeScaleIndex);
BEAGLE_DLLEXPORT int beagleCopyScaleFactors(int instance, int destScalingIndex, int srcScalingIndex);
/* log scale factors of one buffer, patternCount doubles */
BEAGLE_DLLEXPORT int beagleGetScaleFactors(int instance, int srcScalingIndex, double* outScaleFactors);

/* ---------------------------------------------------------------------------------------------
 * log-likelihood integration
 * ------------------------------------------------------------------------------------------- */
/* reference src/mbbeagle.c:1251-1257 (rooted trees) */
BEAGLE_DLLEXPORT int beagleCalculateRootLogLikelihoods(int instance, const int* bufferIndices,
                                                       const int* categoryWeightsIndices,
                                                       const int* stateFrequenciesIndices,
                                                       const int* cumulativeScaleIndices, int count,
                                                       double* outSumLogLikelihood);
/* reference src/mbbeagle.c:1262-1274 (unrooted trees: integrate across the root branch).
 * Returns BEAGLE_ERROR_FLOATING_POINT when the sum is NaN or infinite -- the signal MrBayes'
 * dynamic-rescaling state machine reacts to (src/mbbeagle.c:471-535).  Derivative arguments must be NULL. */
BEAGLE_DLLEXPORT int beagleCalculateEdgeLogLikelihoods(int instance, const int* parentBufferIndices,
                                                       const int* childBufferIndices,
                                                       const int* probabilityIndices,
                                                       const int* firstDerivativeIndices,
                                                       const int* secondDerivativeIndices,
                                                       const int* categoryWeightsIndices,
                                                       const int* stateFrequenciesIndices,
                                                       const int* cumulativeScaleIndices, int count,
                                                       double* outSumLogLikelihood,
                                                       double* outSumFirstDerivative,
                                                       double* outSumSecondDerivative);
/* reference src/mbbeagle.c:1295,1309,1329: per-pattern log-likelihoods of the last Calculate call */
BEAGLE_DLLEXPORT int beagleGetSiteLogLikelihoods(int instance, double* outLogLikelihoods);

/* ---------------------------------------------------------------------------------------------
 * engine extensions (not part of the upstream API; used by bench.py / the multi-GPU driver)
 * ------------------------------------------------------------------------------------------- */
/* Block until all queued device work of the instance has finished. */
/* ---------------------------------------------------------------------------------------------
 * BEAGLE v3 surface (compiled into MrBayes when the library exports beagleSetCPUThreadCount, reference
 * configure.ac:220-223): resource benchmark, multi-partition instances.  A multi-partition instance holds the site
 * patterns of ALL data divisions (same state / category / eigen-part counts); operations, scale-factor bookkeeping and
 * log-likelihood calls name the partition they act on.  In this engine such an instance is a facade over one child engine
 * per partition: the children run on their own streams, only the sums meet on the host.
 * ------------------------------------------------------------------------------------------- */
/* reference src/mbbeagle.c:232-245 */
BEAGLE_DLLEXPORT BeagleBenchmarkedResourceList* beagleGetBenchmarkedResourceList(
    int tipCount, int compactBufferCount, int stateCount, int patternCount, int categoryCount, int* resourceList,
    int resourceCount, long preferenceFlags, long requirementFlags, int eigenModelCount, int partitionCount,
    int calculateDerivatives, long benchmarkFlags);
/* reference src/mbbeagle.c:386 (CPU implementations only; a no-op here) */
BEAGLE_DLLEXPORT int beagleSetCPUThreadCount(int instance, int threadCount);
/* reference src/mcmc.c:6464: patternCount ints, partitions are contiguous increasing pattern ranges; call before the
 * first matrix / partials update (tip data and pattern weights may already be set) */
BEAGLE_DLLEXPORT int beagleSetPatternPartitions(int instance, int partitionCount, const int* inPatternPartitions);
/* reference src/mbbeagle.c:2055 */
BEAGLE_DLLEXPORT int beagleSetCategoryRatesWithIndex(int instance, int categoryRatesIndex, const double* inCategoryRates);
/* reference src/mbbeagle.c:2140-2147 */
BEAGLE_DLLEXPORT int beagleUpdateTransitionMatricesWithMultipleModels(int instance, const int* eigenIndices,
                                                                      const int* categoryRateIndices,
                                                                      const int* probabilityIndices,
                                                                      const int* firstDerivativeIndices,
                                                                      const int* secondDerivativeIndices,
                                                                      const double* edgeLengths, int count);
/* reference src/mbbeagle.c:2292, 2440, 2616 */
BEAGLE_DLLEXPORT int beagleUpdatePartialsByPartition(int instance, const BeagleOperationByPartition* operations, int operationCount);
/* reference src/likelihood.c:8096-8103, 8134; src/mbbeagle.c:593, 1736, 1913, 2566 */
BEAGLE_DLLEXPORT int beagleAccumulateScaleFactorsByPartition(int instance, const int* scaleIndices, int count,
                                                             int cumulativeScaleIndex, int partitionIndex);
BEAGLE_DLLEXPORT int beagleRemoveScaleFactorsByPartition(int instance, const int* scaleIndices, int count,
                                                         int cumulativeScaleIndex, int partitionIndex);
BEAGLE_DLLEXPORT int beagleResetScaleFactorsByPartition(int instance, int cumulativeScaleIndex, int partitionIndex);
/* reference src/mbbeagle.c:2817-2850: index arrays are [count][partitionCount]; outSumLogLikelihoodByPartition has one
 * value per named partition */
BEAGLE_DLLEXPORT int beagleCalculateRootLogLikelihoodsByPartition(int instance, const int* bufferIndices,
                                                                  const int* categoryWeightsIndices,
                                                                  const int* stateFrequenciesIndices,
                                                                  const int* cumulativeScaleIndices,
                                                                  const int* partitionIndices, int partitionCount, int count,
                                                                  double* outSumLogLikelihoodByPartition,
                                                                  double* outSumLogLikelihood);
BEAGLE_DLLEXPORT int beagleCalculateEdgeLogLikelihoodsByPartition(
    int instance, const int* parentBufferIndices, const int* childBufferIndices, const int* probabilityIndices,
    const int* firstDerivativeIndices, const int* secondDerivativeIndices, const int* categoryWeightsIndices,
    const int* stateFrequenciesIndices, const int* cumulativeScaleIndices, const int* partitionIndices, int partitionCount,
    int count, double* outSumLogLikelihoodByPartition, double* outSumLogLikelihood, double* outSumFirstDerivativeByPartition,
    double* outSumFirstDerivative, double* outSumSecondDerivativeByPartition, double* outSumSecondDerivative);

/* ---------------------------------------------------------------------------------------------
 * engine extensions (mbamd*)
 * ------------------------------------------------------------------------------------------- */
/* child engines behind an instance: pattern partitions x shards (1: an ordinary instance).  Site patterns of one
 * partition are sharded over several GPUs when beagleCreateInstance names several resources, or MBAMD_SHARD=<g> is set
 * (MrBayes names at most one resource, reference src/mbbeagle.c:322-323). */
BEAGLE_DLLEXPORT int mbamdGetChildCount(int instance);
BEAGLE_DLLEXPORT int mbamdSynchronize(int instance);
/* The binary exponents behind a scale buffer: out[k * patternCount + c].  The 4-state path keeps one exponent per
 * (pattern, category) -- beagleGetScaleFactors reports the largest of a pattern's exponents times ln 2; the general-state
 * path keeps one per pattern (every category row is the same). */
BEAGLE_DLLEXPORT int mbamdGetScaleExponents(int instance, int srcScalingIndex, int* out);
/* Eigen-systems computed ON THE DEVICE (the host step it replaces: UpDateCijk -> GetEigens, reference
 * src/likelihood.c:10476-10804, src/utils.c:11201): `count` reversible rate matrices q (count x S x S doubles, row-major,
 * rows summing to zero; mode 1: symmetric exchangeabilities r_ij instead, Q_ij = r_ij pi_j is built and normalised on the
 * device) with stationary frequencies pi (all positive) -> [U | U^-1 | lambda] of eigen buffers firstEigenIndex ...
 * The unmodified MrBayes does not call it (the BEAGLE API takes finished eigen-systems: beagleSetEigenDecomposition);
 * hosts that own their model code do (mrbayes_amd/likelihood.py, device_eigen=True). */
BEAGLE_DLLEXPORT int mbamdSetRateMatrices(int instance, int firstEigenIndex, int count, const double* q, const double* pi, int mode);
/* The same for a client whose rate matrices change a little from call to call -- an MCMC move on the substitution parameters:
 * warmFirstEigenIndex (>= 0) names the eigen buffers that hold the systems of the state the proposal started from (MrBayes:
 * m->cijkScratchIndex after FlipCijkSpace, reference src/likelihood.c:5614-5622); if the device computed those too, the
 * iteration starts from their eigenvectors and needs two or three sweeps instead of nine.  All `count` systems -- the parts of a
 * codon or covarion model -- in ONE asynchronous launch; nothing waits.
 * mode bit 0: q are exchangeabilities (as mbamdSetRateMatrices); bit 1 (2): shield -- the NEXT beagleSetEigenDecomposition on
 * each of these eigen buffers is ignored, for clients whose own code path still sends its host result afterwards (the binding
 * of reference UpDateCijk, src/likelihood.c:10476-10804: integration/mrbayes/mbamd_eigen_glue.c). */
BEAGLE_DLLEXPORT int mbamdSetRateMatricesFrom(int instance, int firstEigenIndex, int count, const double* q, const double* pi, int mode,
                                              int warmFirstEigenIndex);
/* Last HIP/engine error text of the calling thread ("" if none). */
BEAGLE_DLLEXPORT const char* mbamdGetLastError(void);
/* Device-side timing of the partials kernels: accumulates HIP-event time (ms) and launch count of every
 * beagleUpdatePartials since the last reset.  enable=0 turns it off (default). */
BEAGLE_DLLEXPORT int mbamdKernelTiming(int instance, int enable);
BEAGLE_DLLEXPORT int mbamdGetKernelTiming(int instance, double* outMilliseconds, long* outLaunches, int reset);
/* How the 4-state beagleUpdatePartials lists of this instance were run since it was made: out[0] lists, out[1] of them root-ward
 * paths (k_path4), out[2] of those forked (arms that join: the lists of topology moves), out[3] paths run together with the
 * log-likelihood behind them as one launch, out[4] lists compiled for the tree-walk kernel, out[5] their operations.  (Counters for
 * tests and MBAMD_STATS; a facade or a double-precision instance reports zeros.) */
BEAGLE_DLLEXPORT int mbamdGetListCounts(int instance, long* out6);
/* While the timing is on: device time (ms) of whole evaluations -- from the first kernel launched after a
 * Calculate*LogLikelihoods call to the end of the next integration kernel, i.e. every kernel of a step (transition matrices,
 * partials, integration) and the gaps between them -- and how many such spans were closed. */
BEAGLE_DLLEXPORT int mbamdGetStepTiming(int instance, double* outMilliseconds, long* outSteps, int reset);
/* Select the partials kernel family: 0 = automatic, 1 = level-synchronous generic kernels,
 * 2 = tree-walk kernel (4-state), 3 = MFMA kernels (20/61-state).  For A/B measurements. */
BEAGLE_DLLEXPORT int mbamdSetKernelPath(int instance, int path);
/* Calculate*LogLikelihoods without the device->host copy: leaves the sum on the device and returns
 * immediately; mbamdFetchLogLikelihood blocks and returns it (same error convention). */
BEAGLE_DLLEXPORT int mbamdSetDeferredResult(int instance, int enable);
/* timing experiments only: per-step clock stamps of workgroup 0 of the tree-walk kernel (needs MBAMD_WALK_TRACE
 * in the environment when the instance is created); out: [steps][8][3] 64-bit stamps */
BEAGLE_DLLEXPORT int mbamdWalkTrace(int instance, long long* out, int maxSteps, int* outSteps, int* outWaves);
BEAGLE_DLLEXPORT int mbamdFetchLogLikelihood(int instance, double* outSumLogLikelihood);
/* The pending (deferred) sum, added up ON THE DEVICE into *deviceOut (device memory of the instance's GPU) in a fixed order, after
 * the integration that produced it; `waitingStream` (a hipStream_t, may be null) is made to wait for it.  No host synchronisation:
 * a multi-GPU client all-reduces deviceOut over RCCL on its own stream and reads one number per step (bench.py, pattern_sharded).
 * The value stays pending: mbamdFetchLogLikelihood still returns it. */
BEAGLE_DLLEXPORT int mbamdReduceLogLikelihood(int instance, double* deviceOut, void* waitingStream);
/* Which GPUs: the PCI bus id ("0000:c1:00.0") of a resource of beagleGetResourceList, and the resource numbers of the child engines
 * of an instance (a plain instance: one; a sharded one: a device per shard) -- so that a multi-GPU driver can PROVE that its ranks /
 * shards sit on different devices.  mbamdGetInstanceDevices returns the number of children (<= maxCount written). */
BEAGLE_DLLEXPORT int mbamdGetResourcePciBusId(int resource, char* out, int length);
BEAGLE_DLLEXPORT int mbamdGetInstanceDevices(int instance, int* outResources, int maxCount);

#ifdef __cplusplus
}
#endif
#endif /* MBAMD_LIBHMSBEAGLE_BEAGLE_H_ */
