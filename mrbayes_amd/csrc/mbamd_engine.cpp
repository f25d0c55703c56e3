// mbamd_engine.cpp -- host side of the MI355X conditional-likelihood engine and its C ABI
// (include/libhmsbeagle/beagle.h).  One Instance = one MrBayes data division: it owns every
// partials / transition-matrix / scale buffer of all local chains in HBM (the reference's
// condLikes/tiProbs/scalers arrays, src/mcmc.c:5703-6510) and turns each BEAGLE call coming from
// src/mbbeagle.c into HIP kernel launches on a private stream.
//
// Built by hipcc for gfx950 (see mrbayes_amd/build.py).  There is no CPU code path in the product:
// without a HIP device beagleCreateInstance fails with BEAGLE_ERROR_NO_RESOURCE.
#include <mbamd_dev_runtime.h>   // the HIP runtime + launch macros (csrc/device/; tests/hostemu/ has the CPU stand-in for the test build)

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <limits>
#include <memory>
#include <unordered_map>
#include <vector>

#include "libhmsbeagle/beagle.h"
#include "mbamd_kernels.h"
#include "mbamd_reports.h"
#include "libhmsbeagle/mbamd_reports.h"
#include "mbamd_walk4_host.h"
#include "mbamd_kernels_mfma.h"

#include <chrono>

namespace mbamd {

// MBAMD_STATS=1: per-entry-point call counts and host wall time, printed when an instance is finalized
struct ApiStats {
    const char* name;
    long calls = 0;
    double seconds = 0.0;
};
static ApiStats g_stats[] = {{"beagleUpdateTransitionMatrices"}, {"beagleUpdatePartials"}, {"beagleCalculate*LogLikelihoods"},
                             {"beagle*ScaleFactors"}, {"beagleSet*"}, {"beagleGetSiteLogLikelihoods"}, {"plan build"},
                             {"mbamdParsDownPass/FinalPass"}, {"mbamdParsScore"},
                             {"  (launching the deferred work)"}, {"  (waiting for the device)"},
                             {"  (parsimony: compiling the queued passes)"}, {"  (parsimony: waiting for the device)"}};
enum { ST_MATRICES = 0, ST_PARTIALS, ST_LNL, ST_SCALE, ST_SET, ST_SITE, ST_PLAN, ST_PARS_PASS, ST_PARS_SCORE, ST_FLUSH, ST_WAIT, ST_PARS_COMPILE, ST_PARS_WAIT };
static const bool g_statsOn = std::getenv("MBAMD_STATS") != nullptr;
// MBAMD_API_TRACE=1: one stderr line per C-ABI call (integration debugging: what does the client really send?)
static const bool g_apiTrace = std::getenv("MBAMD_API_TRACE") != nullptr;
#define API_TRACE(...) do { if (g_apiTrace) { std::fprintf(stderr, "[mbamd api] " __VA_ARGS__); std::fputc('\n', stderr); } } while (0)
static std::string trace_ints(const int* v, int n) {
    std::string r = "[";
    for (int i = 0; v && i < n; ++i) r += (i ? "," : "") + std::to_string(v[i]);
    return r + "]";
}
static std::string trace_doubles(const double* v, int n) {
    std::string r = "[";
    char buf[32];
    for (int i = 0; v && i < n; ++i) { std::snprintf(buf, sizeof buf, "%s%.6g", i ? "," : "", v[i]); r += buf; }
    return r + "]";
}
struct StatTimer {
    int id;
    std::chrono::steady_clock::time_point t0;
    explicit StatTimer(int i) : id(i) { if (g_statsOn) t0 = std::chrono::steady_clock::now(); }
    bool stopped = false;
    void stop()                                      // (a span that ends before its scope does)
    {
        if (!g_statsOn || stopped) return;
        stopped = true;
        g_stats[id].calls++;
        g_stats[id].seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    ~StatTimer() { stop(); }
};

static thread_local std::string g_last_error;

static int fail(int code, const char* what, const char* detail = "")
{
    g_last_error = std::string(what) + (detail[0] ? ": " : "") + detail;
    if (std::getenv("MBAMD_VERBOSE")) std::fprintf(stderr, "[mbamd] error %d: %s\n", code, g_last_error.c_str());
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? BEAGLE_ERROR_OUT_OF_MEMORY : BEAGLE_ERROR_GENERAL, \
                        #expr, hipGetErrorString(e_));                                             \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// state counts the 20/61-state tree walk (mbamd_walkg.h) is instantiated for: amino acids, doublets, and the sense codons of
// every genetic code MrBayes knows (60 vertebrate mitochondrial ... 63; reference src/model.c SetCode)
// state counts MrBayes sends: restriction sites 2, covarion nucleotides 8, doublets 16, amino acids 20, covarion amino acids 40, the
// sense codons of every genetic code 60..63 (4 has its own kernel; anything else runs on the level kernels)
// (round 5: 3, 5, 6, 7, 9, 10 as well -- the state counts of standard (morphology) characters, whose transition-matrix classes are
//  engine instances of a few hundred patterns: a launch per dependency level of the level kernels was ten launches where this is one)
static inline bool wg_compiled(int S) { return (S >= 2 && S <= 10 && S != 4) || S == 16 || S == 20 || S == 40 || (S >= 60 && S <= 63); }
// FN<SC, WMAX, CH, DEPTH>: one row tile -> whole jobs two ahead; two row tiles -> half jobs one ahead (see k_walkg)
#if !defined(MBAMD_WG_DEPTH61)
#define MBAMD_WG_DEPTH61 1       // chunks the operand fetch of the 60..63-state kernels runs ahead (experiments: 2)
#endif
#define MBAMD_WG_DISPATCH(S, FN, ...)                                   \
    switch (S) {                                                        \
        case 2: FN<2, 8, 1, 2>(__VA_ARGS__); break;                     \
        case 3: FN<3, 8, 1, 2>(__VA_ARGS__); break;                     \
        case 5: FN<5, 8, 1, 2>(__VA_ARGS__); break;                     \
        case 6: FN<6, 8, 1, 2>(__VA_ARGS__); break;                     \
        case 7: FN<7, 8, 1, 2>(__VA_ARGS__); break;                     \
        case 8: FN<8, 8, 1, 2>(__VA_ARGS__); break;                     \
        case 9: FN<9, 8, 1, 2>(__VA_ARGS__); break;                     \
        case 10: FN<10, 8, 1, 2>(__VA_ARGS__); break;                   \
        case 16: FN<16, 8, 1, 2>(__VA_ARGS__); break;                   \
        case 20: FN<20, 8, 1, 2>(__VA_ARGS__); break;                   \
        case 40: FN<40, 4, 1, 1>(__VA_ARGS__); break;                   \
        case 60: FN<60, 4, 2, MBAMD_WG_DEPTH61>(__VA_ARGS__); break;    \
        case 61: FN<61, 4, 2, MBAMD_WG_DEPTH61>(__VA_ARGS__); break;    \
        case 62: FN<62, 4, 2, MBAMD_WG_DEPTH61>(__VA_ARGS__); break;    \
        default: FN<63, 4, 2, MBAMD_WG_DEPTH61>(__VA_ARGS__); break;    \
    }

template <int SC_, int WMAX_, int CH_, int DEPTH_>
static void raise_walkg_lds(int maxLds)
{
    if (hipFuncSetAttribute((const void*) k_walkg<SC_, WMAX_, CH_, DEPTH_>, hipFuncAttributeMaxDynamicSharedMemorySize, maxLds) != hipSuccess ||
        hipFuncSetAttribute((const void*) k_walkg<SC_, WMAX_, CH_, DEPTH_, WalkGArgsInline>, hipFuncAttributeMaxDynamicSharedMemorySize, maxLds) != hipSuccess)
        (void) hipGetLastError();
}

enum KernelPath { PATH_AUTO = 0, PATH_GENERIC = 1, PATH_WALK = 2, PATH_MFMA = 3 };

// A compiled operation list: the device-resident table a partials kernel walks, cached under the exact
// BeagleOperation array it was built from.  MrBayes re-issues the same lists all the time (every move
// that dirties the whole tree alternates between the two buffer-flip states), so the host-side
// scheduling and the table upload happen once per distinct list, not once per generation.
struct Plan {
    std::vector<int> key;            // the BeagleOperation ints + cumulative index + layout epoch
    uint64_t hash = 0, lastUse = 0;
    uint64_t lastLaunch = 0;         // Instance::launchClock value of the latest launch that reads d_table
    PartialsOp* d_table = nullptr;
    size_t cap = 0;                  // bytes allocated for d_table
    struct Segment {                             // tree-walk path: one launch per hazard-free segment
        size_t first; int W, entries, nslots, tail = 2;
    };
    std::vector<Segment> segments;               // (Walk4Entry index of its program in d_table, geometry)
    std::vector<Walk4Entry> inlineProg;          // 4-state walk: a short single-segment program travels in the kernel arguments instead
    bool path = false;                           // 4-state walk: the list is a root-ward path -- inlineProg holds k_path4's entries
    bool forked = false;                         // ... of several arms that join (the list of a topology move)
    bool pathG = false;                          // 20/61-state walk: every list is a root-ward path of the same length -- inlineProg holds k_pathg's entries
    int lists = 1;                               // 20/61-state walk: > 1 = the segments are that many independent lists, ONE launch
    std::vector<int> start;                      // general path: first table entry of each dependency level
    bool anyScale = false;
    bool narrow = false;                         // general path: few operations per level -> one serial launch
    std::vector<std::pair<int, int>> chains;     // narrow general-state lists: (first table entry, operations) of up to four
                                                 // mutually independent sub-lists (they walk in parallel workgroups)
    std::vector<std::pair<int, int>> spineChains;   // the same for the serial tail of a level-launched list (levels >= serialFrom)
    int tipTip = 0;                              // general path: the first tipTip operations of level 0 have two compact tip children
    int serialFrom = 0;                          // general path: levels >= serialFrom are narrow (the spine towards the
                                                 // root): they run as one serial launch after the level launches
    std::vector<int> bufsRead, bufsWritten, scalesUsed;   // buffer / scale indices the list touches (deferral hazards)
};

struct Instance {
    // ---- facade: an instance whose site patterns are split over child instances -- one per pattern PARTITION
    // (BEAGLE v3 multi-partition mode, reference src/mbbeagle.c:1500-3010) and/or per SHARD (pattern blocks of one
    // partition on different GPUs).  Site patterns are independent through the whole recursion, so every
    // call fans out to the children (each with its own stream, possibly its own device) and only the log-likelihood
    // sums meet again on the host.  A facade owns no device memory itself.
    struct Child { Instance* in; int start, count, partition; };
    std::vector<Child> children;
    bool facade() const { return !children.empty(); }
    int createArgs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<int> shardDevices;   // devices the patterns of every partition are spread over (size 1: no sharding)
    int partitionCount = 1;
    bool released = false;           // device buffers handed back (became a facade)
    // tip data and pattern weights as the client gave them, kept until the first computation: a v3 client sets them
    // BEFORE it declares the partitions (reference src/mbbeagle.c:1655-1700 then src/mcmc.c:6461-6466)
    bool logOpen = true;
    std::vector<std::pair<int, std::vector<int>>> logTipStates;
    std::vector<std::pair<int, std::vector<double>>> logTipPartials;
    std::vector<double> logWeights;
    int makeChildren(const std::vector<std::pair<int, int>>& partitionRanges);
    void destroyChildren();
    int getSites(double* out);
    int getScaleExponents(int idx, int* out);
    int finalPass(const MbamdFinalOperation* ops, int count);
    int getScaledPartials(int idx, int cumIdx, float* out, float* outLn);
    void closeLog()                  // the first computation: the set-up data is where it belongs, drop the host copies
    {
        if (!logOpen) return;
        logOpen = false;
        std::vector<std::pair<int, std::vector<int>>>().swap(logTipStates);
        std::vector<std::pair<int, std::vector<double>>>().swap(logTipPartials);
        std::vector<double>().swap(logWeights);
    }

    int device = 0;
    hipStream_t stream{};
    int tipCount = 0, nBuffers = 0, S = 0, SP = 0, P = 0, Ppad = 0, nEigen = 0, nMatrices = 0, K = 0, nScale = 0;
    bool s4 = false;                 // 4-state float4 layout + tree-walk kernel
    bool wg = false;                 // 20/61-state tree-walk kernel on the matrix cores (mbamd_walkg.h) + its arenas
    bool noWalkG = false;            // the arenas of that path did not fit: level kernels with buffers allocated on first use
    bool arena() const { return s4 || wg; }   // buffers are slices of arenas, exponents are per (pattern, category)
    bool mfma = false;               // general-state path on the matrix cores (mbamd_kernels_mfma.h)
    bool mfmaWhole = false;          // MBAMD_MFMA_WHOLE: one wave per (operation, 32 patterns) instead of per factor tile
    int walkWaves = 1, lastWalkSteps = 0;   // (kernel trace bookkeeping of the serial MFMA kernels, tools/trace_*.py)
    // ---- 4-state tree-walk path (mbamd_walk4.h / mbamd_walk4_host.h) -------------------------------------------
    Walk4Builder w4;                 // launch geometry limits + the program compiler
    // Programs are a function of the list's dependency STRUCTURE only (who produces whose child, which children are
    // tips): the buffer / matrix / scale indices -- which change with every accept / reject flip -- just fill the
    // entries.  A move that touches the branches it touched before re-uses its template and only re-fills it.
    std::unordered_map<uint64_t, Walk4Template> w4templates;
    uint64_t scheduleHits = 0, scheduleMisses = 0;
    std::vector<Walk4Op> w4ops;      // scratch
    std::vector<int> w4key, w4writer, w4segList;                    // buildWalk scratch: no allocation per compiled list
    std::vector<char> w4written, w4segRead, w4segWritten, w4segScale;
    std::vector<Walk4Entry> w4table;
    int8_t* arenaExp = nullptr;      // node exponents int8 [block][scale buffer][K][64] (+ one scratch buffer)
    unsigned estride = 0;            // bytes between blocks
    std::vector<int32_t*> wideScale; // cumulative exponents int32 [K][Ppad], allocated on first use
    std::vector<char> scaleState;    // 0 = never written (zero), 1 = node exponents in the arena, 2 = cumulative (wide)
    int lastWalkW = 0, lastWalkSlots = 0, lastWalkEntries = 0, lastWalkPhases = 0;
    bool walkCumFresh = false;       // the cumulative buffer of the list being submitted holds nothing yet (store, do not add)
    // ---- 20/61-state tree walk: lists are deferred and merged (MrBayes submits one list per eigen-system part of a codon
    // model, reference src/mbbeagle.c:1095-1104; together they are ONE forest for the program compiler)
    std::vector<BeagleOperation> wgOps;          // operations of the deferred lists, concatenated
    std::vector<int> wgListStart, wgListCum;     // first operation / cumulative scale index (or BEAGLE_OP_NONE) of each list
    int32_t* wgCum[MBAMD_WG_MAXLISTS] = {nullptr, nullptr, nullptr, nullptr};
    int wgFresh = 0;
    uint8_t* arenaTipStates = nullptr;           // uint8 [tile][buffer][32]
    unsigned long wgTileBytes = 0;               // partials arena: bytes between 32-pattern tiles
    unsigned wgTipTileBytes = 0;
    size_t wgTabFloats = 0;                      // first float of the tree-walk tables inside a matrix buffer
    bool hasPending() const { return !pending.empty() || !wgListCum.empty() || heldPath != nullptr; }
    // ---- 4-state path: a root-ward path (k_path4 plan) is HELD until the next call: if that call is the log-likelihood over the
    // path's last result, both run as one launch (k_path4_lnl); anything else runs the path first, as before
    Plan* heldPath = nullptr;
    int32_t* heldPathCum = nullptr;
    bool heldPathFresh = false;
    int heldPathDst = -1;                        // the partials buffer the path's last operation writes
    bool noFusePath = false;                     // MBAMD_NO_FUSE_PATH: never hold a path
    bool noForkPath = false;                     // MBAMD_NO_FORK_PATH: paths that join are compiled for k_walk4_t (A/B)
    int runHeldPath();
    int integratePath4(const int* parent, const int* child, const int* prob, const int* wIdx, const int* fIdx, const int* cumIdx);
    int updatePartialsG(const BeagleOperation* ops, int n, int cumIdx);
    int flushWalkG();
    int runWalkG(const Plan& plan);
    bool buildPath4(Plan& plan, const BeagleOperation* ops, int n);
    bool buildPathG(Plan& plan, const BeagleOperation* ops, int n, const std::vector<int>& starts, int nl);
    bool noPathG = false;                        // MBAMD_NO_PATHG: root-ward paths of the general-state walk through k_walkg
    bool noPath4 = false;                        // MBAMD_NO_PATH4: root-ward paths on k_walk4_t too
    void postResultFlag();
    bool scaleOpsIndependentOfPending(const int* idx, int n, int cumIdx) const;
    uint64_t launchClock = 0, syncedClock = 0;   // launches issued / launches known complete (last stream synchronisation)
    uint64_t flagClock = 0;                      // launchClock when the polled result flag was queued (postResultFlag)
    uint32_t siteSeq = 0, seenSeq = 0;           // flag value behind the integration that wrote the latest site values / latest flag value fetched
    std::vector<double> h_freqs, h_weights;      // host mirrors of d_freqs / d_weights (uploadIfChanged)
    long long* d_trace = nullptr;    // MBAMD_WALK_TRACE: per-step clock stamps of workgroup 0 (timing experiments)

    int NT = 0, T = 0;               // MFMA packing: i-tiles of 32 rows, j-pairs
    long flags = 0;
    class Engine64* f64 = nullptr;        // BEAGLE_FLAG_PRECISION_DOUBLE: this object is only the handle, the engine is mbamd_f64.h
    size_t partialsFloats = 0, matrixFloats = 0, eigenDoubles = 0;
    int path = PATH_AUTO;

    std::vector<float*> partials;      // general path: allocated on first use; 4-state path: slices of the arena
    std::vector<uint8_t*> tipStates;   // non-null while the buffer holds compact tip states
    std::vector<int32_t*> scale;
    std::vector<char> valid;           // partials buffer has been written (import or operation destination)
    // 4-state path: arenas (see mbamd_kernels.h: partials buffer-major, tips and exponents block-major), one allocation each
    float* arenaPartials = nullptr;
    uint64_t* arenaTips = nullptr;     // state bitplanes uint64 [block][buffer][4]
    BlockGeom geom{64, 64, 64};        // general path: linear [P_pad] arrays == block stride 64
    float* matrices = nullptr;
    double *d_eigen = nullptr, *d_freqs = nullptr, *d_weights = nullptr, *d_rates = nullptr, *d_pweights = nullptr;
    double *d_site = nullptr;
    // Clients that read the per-pattern values after every evaluation (MrBayes does for +I models,
    // src/mbbeagle.c:1295-1358) get them written straight into pinned host memory by the integration kernel:
    // switched on by the first beagleGetSiteLogLikelihoods call, from then on that call is a host memcpy.
    double* h_site = nullptr;
    double* h_site_dev = nullptr;
    bool siteToHost = false, siteOnHost = false;   // mode / where the latest evaluation put its values
    bool noSiteHost = false;                       // MBAMD_NO_SITE_HOST: always copy from device memory (comparison switch)
    int nblocks = 0;                  // partial sums of the weighted site log-likelihoods (one per integration workgroup)
    std::vector<RatesArg> rateSets;   // category rates by index (beagleSetCategoryRatesWithIndex; index 0 = beagleSetCategoryRates), passed to kernels by value
    int pendingRateSet = 0;           // the rate set of the queued transition-matrix jobs
    bool haveSite = false;

    // growable device scratch
    double* d_ev = nullptr;           size_t evCap = 0;
    void* d_tmp = nullptr;            size_t tmpCap = 0;

    // pinned staging ring for small asynchronous uploads / downloads
    unsigned char* stage = nullptr;
    size_t stageCap = 0, stageOff = 0;
    double* h_sums = nullptr;         // pinned host memory the integration kernel writes its block sums to
    double* h_sums_dev = nullptr;     // the device-side address of h_sums
    // The result is waited for by polling a word in pinned host memory that the STREAM writes behind the integration kernel
    // (hipStreamWriteValue32): 7 us less per evaluation than hipStreamSynchronize on the same stream (MrBayes fixed-topology
    // generation 83 -> 76 us, profiles/r04_mcmc_fixed_topology.txt).  A long wait falls back to the runtime's own wait.
    uint32_t* h_flag = nullptr;       // sequence number of the last integration whose results have landed
    uint32_t* h_flag_dev = nullptr;
    uint32_t flagSeq = 0;             // ... of the last integration launched
    bool pollResult = false;          // (off: MBAMD_NO_POLL, or the stream refused the write)
    // Round 6: the block sums are their own completion signal.  Before an integration is launched the host fills h_sums with a
    // bit pattern no sum can have; fetchResult then waits for every block sum to differ from it -- no gap + stream write behind the
    // kernel (8.6 us of every evaluation, profiles/r06_walk61.txt), and no fence in the kernel (each sum is one 8-byte store to
    // host-coherent memory).  Armed only when nothing else can still write h_sums (no unfetched result) and not in deferred mode
    // (mbamdReduceLogLikelihood reads them on the device).  MBAMD_NO_SUM_POLL=1: the stream's flag only (A/B).
    bool pollSums = false, sumsArmed = false, flagWritten = false;
    static constexpr uint64_t kSumSentinel = 0x7FF4DEADBEEF0001ull;      // a signalling NaN with a payload no arithmetic produces
    void armSums();
    unsigned char* stage_dev = nullptr;   // the device-side address of the staging ring

    // timing of the partials kernels
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double timedMs = 0.0;
    long timedLaunches = 0, pendingLaunches = 0;
    // ... and of whole evaluations: from the first kernel after a log-likelihood call (transition matrices, usually) to the
    // integration kernel's end -- every kernel of a step and the gaps between them, nothing of the host's wait
    hipEvent_t spanEv0{};
    bool spanOpen = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    double spanMs = 0.0;
    long spanCount = 0;
    int spanBegin()
    {
        if (!timing || spanOpen) return BEAGLE_SUCCESS;
        HIP_TRY(hipEventCreate(&spanEv0));
        HIP_TRY(hipEventRecord(spanEv0, stream));
        spanOpen = true;
        return BEAGLE_SUCCESS;
    }
    int spanEnd()
    {
        if (!spanOpen) return BEAGLE_SUCCESS;
        hipEvent_t e1{};
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e1, stream));
        spans.emplace_back(spanEv0, e1);
        spanOpen = false;
        if (spans.size() > 4096) {                   // a client that never polls: fold the finished spans into the running total
            HIP_TRY(hipStreamSynchronize(stream));
            spanFold();
        }
        return BEAGLE_SUCCESS;
    }
    int spanFold()                   // (stream synchronised by the caller)
    {
        for (auto& ev : spans) {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess) { spanMs += t; ++spanCount; }
            (void) hipEventDestroy(ev.first);
            (void) hipEventDestroy(ev.second);
        }
        spans.clear();
        return BEAGLE_SUCCESS;
    }

    bool deferred = false, pendingResult = false;
    hipEvent_t reduceEvent{};        // mbamdReduceLogLikelihood: orders a client's stream behind the device-side sum
    // final pass (mbamd_reports.h): per partials buffer, the exponents [K][Ppad] its final partials carry (nullptr: not final
    // partials); owned by the top node's destination buffers
    std::vector<int32_t*> finalExpOf;
    std::unordered_map<int, int32_t*> finalExpOwn;

    std::vector<std::pair<Plan*, int>> pending;   // deferred general-path lists (plan, cumulative scale index or -1)
    std::vector<Plan*> plans;        // small LRU cache of compiled operation lists
    uint64_t planClock = 0;
    int layoutEpoch = 0;             // bumped whenever a buffer changes between compact-tip and partials form
    long planHits = 0, planMisses = 0, fusedPaths = 0, heldPaths = 0, forkedPaths = 0, listsTotal = 0, listsPath = 0, opsWalked = 0, listsWalked = 0;

    // ---- helpers ----------------------------------------------------------------------------
    int grow(void** p, size_t* cap, size_t bytes)
    {
        if (bytes <= *cap) return BEAGLE_SUCCESS;
        HIP_TRY(hipStreamSynchronize(stream));
        if (*p) HIP_TRY(hipFree(*p));
        *p = nullptr;
        size_t n = std::max(bytes, *cap * 2);
        HIP_TRY(hipMalloc(p, n));
        *cap = n;
        return BEAGLE_SUCCESS;
    }

    // copy host bytes to the device asynchronously through the pinned ring
    int upload(void* dst, const void* src, size_t bytes)
    {
        if (bytes == 0) return BEAGLE_SUCCESS;
        if (bytes > stageCap / 2) {            // big one-off transfers (tip data): plain blocking copy
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
            return BEAGLE_SUCCESS;
        }
        // (round 6: small transfers -- eigen-systems, frequencies, weights, compiled programs -- go through the pinned ring and a copy
        //  kernel of ours: hipMemcpyAsync costs the host ~10 us a call and its blit kernel left the walk behind it 30 % slower,
        //  profiles/r06_ring_copy.txt)
        if (!noRingCopy && bytes <= ((size_t) 256 << 10) && bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 3u) == 0) return ringCopy(dst, src, bytes);
        size_t need = (bytes + 63) & ~(size_t) 63;
        if (stageOff + need > stageCap) {
            HIP_TRY(hipStreamSynchronize(stream));
            stageOff = 0;
        }
        std::memcpy(stage + stageOff, src, bytes);
        HIP_TRY(hipMemcpyAsync(dst, stage + stageOff, bytes, hipMemcpyHostToDevice, stream));
        stageOff += need;
        return BEAGLE_SUCCESS;
    }
    int ringCopy(void* dst, const void* src, size_t bytes);

    // MrBayes re-sends state frequencies and category weights before every evaluation (reference
    // src/mbbeagle.c:1179-1225); only a changed vector costs a stream operation.  `shadow` mirrors the device array.
    int uploadIfChanged(std::vector<double>& shadow, size_t off, double* devBase, const double* src, int n)
    {
        if (shadow.size() < off + n) shadow.resize(off + n, std::numeric_limits<double>::quiet_NaN());
        if (std::memcmp(shadow.data() + off, src, sizeof(double) * n) == 0) return BEAGLE_SUCCESS;
        std::memcpy(shadow.data() + off, src, sizeof(double) * n);
        return upload(devBase + off, src, sizeof(double) * n);
    }

    // small kernel inputs (job lists, pointer lists): placed in the pinned ring and read by the kernel
    // directly over the host link -- no copy engine, no extra stream operation
    int stageDirect(const void* src, size_t bytes, const void** devPtr)
    {
        size_t need = (bytes + 63) & ~(size_t) 63;
        if (need > stageCap / 2) return fail(BEAGLE_ERROR_OUT_OF_MEMORY, "staging ring too small");
        if (stageOff + need > stageCap) {
            HIP_TRY(hipStreamSynchronize(stream));
            stageOff = 0;
        }
        std::memcpy(stage + stageOff, src, bytes);
        *devPtr = stage_dev + stageOff;
        stageOff += need;
        return BEAGLE_SUCCESS;
    }

    int ensurePartials(int idx)
    {
        if (partials[idx]) return BEAGLE_SUCCESS;
        if (arena()) return fail(BEAGLE_ERROR_GENERAL, "partials arena not initialised");
        float* p = nullptr;
        HIP_TRY(hipMalloc(&p, partialsFloats * sizeof(float)));
        HIP_TRY(hipMemsetAsync(p, 0, partialsFloats * sizeof(float), stream));
        partials[idx] = p;
        return BEAGLE_SUCCESS;
    }
    int ensureScale(int idx)
    {
        if (scale[idx]) return BEAGLE_SUCCESS;
        if (arena()) return fail(BEAGLE_ERROR_GENERAL, "exponent arena not initialised");
        int32_t* p = nullptr;
        HIP_TRY(hipMalloc(&p, (size_t) Ppad * sizeof(int32_t)));
        HIP_TRY(hipMemsetAsync(p, 0, (size_t) Ppad * sizeof(int32_t), stream));
        scale[idx] = p;
        return BEAGLE_SUCCESS;
    }
    float* matrixPtr(int idx) const { return matrices + (size_t) idx * matrixFloats; }

    int create(int tipCount_, int partialsBufferCount, int compactBufferCount, int stateCount, int patternCount,
               int eigenBufferCount, int matrixBufferCount, int categoryCount, int scaleBufferCount, int dev);
    void destroy();

    int configureWalk();
    void wgGeometry(int lists, int& W, int& slots) const;
    int setTipStates(int tip, const int* states);
    int setTipMasks(int tip, const std::vector<uint8_t>& masks);
    int importPartials(int idx, const double* in, bool hasCategories);
    int getPartials(int idx, double* out);
    int setEigen(int idx, const double* U, const double* Ui, const double* lam);
    int setRateMatrices(int first, int count, const double* q, const double* pi, int mode, int warmFirst = -1);
    std::vector<int> eigenWarm;      // per eigen buffer: -1 = no orthonormal basis stored (host-set), else warm starts since the last cold one
    std::vector<char> eigenShield;   // per eigen buffer: the next beagleSetEigenDecomposition is ignored (mbamdSetRateMatricesFrom, mode bit 1)
    int updateMatrices(int eigenIndex, const int* probIdx, const double* lengths, int count, int rateSet = 0);
    int setRates(int index, const double* r);
    int setMatrix(int idx, const double* in);
    int getMatrix(int idx, double* out);
    int updatePartials(const BeagleOperation* ops, int n, int cumIdx);
    int updatePartials4(const BeagleOperation* ops, int n, int cumIdx);
    int buildWalk(Plan& plan, const BeagleOperation* ops, int n, const int* listOf = nullptr, bool perList = false);
    int ensureWide(int idx);
    int accumulate4(const int* idx, int n, int cumIdx, int sign);
    int deferredReset = -1;          // a beagleResetScaleFactors not launched yet (level-kernel path), see beagleAccumulateScaleFactors
    int runDeferredReset();
    int integrate4(const int* parent, const int* child, const int* prob, const int* wIdx, const int* fIdx,
                   const int* cumIdx, int count);
    int buildGeneric(Plan& plan, std::vector<PartialsOp>& dev, const std::vector<int>& dstIdx, const std::vector<int>& c1Idx,
                     const std::vector<int>& c2Idx);
    int runWalk(const Plan& plan, int32_t* cum);
    int runGeneric(const Plan& plan, int32_t* cum);
    int planTable(Plan& plan, const std::vector<PartialsOp>& table);
    int timedRun(const Plan& plan, int32_t* cum);
    int flushPending(bool keepPath = false);
    int flushMatrices();
    std::vector<MatrixJob> pendingJobs;          // queued beagleUpdateTransitionMatrices work
    std::vector<char> pendingMatrixOut;          // matrix buffers the queued jobs write
    int submit(Plan* plan, int cumIdx, int32_t* cumPtr);
    bool noDefer = false;            // MBAMD_NO_DEFER: run every list at once
    bool noInlineJobs = false;       // MBAMD_NO_INLINE_JOBS: a branch move's matrix jobs through the pinned ring too (A/B)
    bool noRingCopy = false;         // MBAMD_NO_RING_COPY: a walk program reaches its device buffer by hipMemcpyAsync (A/B)
    bool eagerMatrices = false;      // four states (one call per evaluation): beagleUpdateTransitionMatrices launches the matrix kernel itself -- it runs while
                                     // MrBayes assembles the operation list (+2 % on both chains, profiles/r06_scale_read.txt); MBAMD_LAZY_MATRICES=1: queued as for the other models
    bool noInlinePrograms = false;   // MBAMD_NO_INLINE_PROGRAMS: every walk program through a device buffer
    bool envVerbose = false, envTrace = false;   // MBAMD_VERBOSE, MBAMD_WALK_TRACE (read once)
    bool noSpine = false;            // MBAMD_NO_SPINE: serial launches use the plain (not software-pipelined) kernel
    int spineWidth = 1;              // MBAMD_SPINE_WIDTH: trailing levels of at most this many operations join the serial launch
    int serialRatio = 4;             // MBAMD_MFMA_SERIAL: lists with <= ratio * levels operations run as ONE serial launch (0 = never)
    bool independentOfPending(const Plan& plan, int cumIdx);
    int accumulate(const int* idx, int n, int cumIdx, int sign, bool fresh = false);   // fresh: cumIdx was reset just before -- store, do not add
    int integrate(const int* parent, const int* child, const int* prob, const int* wIdx, const int* fIdx,
                  const int* cumIdx, int count, double* out);
    int fetchResult(double* out);
};

static bool launch_mfma_split(Instance& in, const OpTables& tabs, int count);
static bool launch_mfma_serial(Instance& in, const OpTables& tabs, int ntables);
static bool launch_tips(Instance& in, const OpTables& tabs, int count);

static std::mutex g_mutex;
static std::vector<Instance*> g_instances;

// A new engine for one device.  The 20/61-state tree walk allocates every buffer up front (arenas); if that does not fit,
// the same instance is set up once more on the level kernels, which allocate a buffer when it is first written.
static int new_engine(Instance*& out, long flags, int tipCount, int partialsBufferCount, int compactBufferCount, int stateCount,
                      int patternCount, int eigenBufferCount, int matrixBufferCount, int categoryCount, int scaleBufferCount, int dev)
{
    for (int attempt = 0; attempt < 2; ++attempt) {
        Instance* c = new Instance();
        c->flags = flags;
        c->noWalkG = attempt == 1;
        const int rc = c->create(tipCount, partialsBufferCount, compactBufferCount, stateCount, patternCount, eigenBufferCount,
                                 matrixBufferCount, categoryCount, scaleBufferCount, dev);
        if (rc == BEAGLE_SUCCESS) { out = c; return rc; }
        const bool retry = rc == BEAGLE_ERROR_OUT_OF_MEMORY && c->wg && attempt == 0;
        c->destroy();
        delete c;
        (void) hipGetLastError();
        if (!retry) return rc;
    }
    return BEAGLE_ERROR_OUT_OF_MEMORY;
}

static Instance* lookup(int id)
{
    std::lock_guard<std::mutex> lk(g_mutex);
    if (id < 0 || id >= (int) g_instances.size()) return nullptr;
    return g_instances[id];
}

// ---------------------------------------------------------------------------------------------
int Instance::create(int tipCount_, int partialsBufferCount, int compactBufferCount, int stateCount,
                     int patternCount, int eigenBufferCount, int matrixBufferCount, int categoryCount,
                     int scaleBufferCount, int dev)
{
    device = dev;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    tipCount = tipCount_;
    nBuffers = partialsBufferCount + compactBufferCount;
    S = stateCount;
    P = patternCount;
    Ppad = round_up(P, 64);
    K = categoryCount;
    nEigen = eigenBufferCount;
    nMatrices = matrixBufferCount;
    nScale = scaleBufferCount;
    const bool forceGeneric = std::getenv("MBAMD_FORCE_GENERIC") != nullptr;
    // the 4-state tree walk addresses buffers with 32-bit byte offsets inside a (block, category) column set (Walk4Entry)
    s4 = (S == 4 && !forceGeneric && (size_t) nBuffers * K * 1024 < ((size_t) 1 << 32) && (size_t) nMatrices * K * 64 < ((size_t) 1 << 32) &&
          (size_t) (nScale + MBAMD_W4_SCRATCH_ROWS) * K * 64 < ((size_t) 1 << 32));
    // 20 / 61 states: the tree-walk kernel on the matrix cores (MBAMD_NO_WALKG=1: the level kernels of mbamd_kernels_mfma.h)
    {
        const size_t tb = wg_block_bytes(S), mf = (size_t) K * 64 * 64 + (size_t) K * wg_table_floats(S);
        wg = !s4 && wg_compiled(S) && K <= 16 && !forceGeneric && !noWalkG && std::getenv("MBAMD_NO_WALKG") == nullptr &&
             (size_t) (nBuffers + 1) * K * tb < ((size_t) 1 << 32) && (size_t) nMatrices * mf * 4 < ((size_t) 1 << 32) &&
             (size_t) (nScale + MBAMD_WG_SCRATCH_ROWS) * K * 64 < ((size_t) 1 << 32) && (size_t) nBuffers * MBAMD_WG_TW < ((size_t) 1 << 32);
    }
    if (s4) SP = 4;
    else if (S <= 4) SP = 4;
    else if (S <= 8) SP = 8;
    else if (S <= 16) SP = 16;
    else if (S <= 20) SP = 20;
    else if (S <= 32) SP = 32;
    else SP = 64;
    NT = (S + 31) / 32;
    T = (S + 1) / 2;
    mfma = !s4 && !wg && S >= 5 && S <= 64 && ((NT == 1 && K <= 4) || (NT == 2 && K <= 2)) &&
           std::getenv("MBAMD_NO_MFMA") == nullptr;
    if (mfma) SP = 32 * NT;          // transposed matrices padded to the MFMA tile height
    mfmaWhole = std::getenv("MBAMD_MFMA_WHOLE") != nullptr;
    noDefer = std::getenv("MBAMD_NO_DEFER") != nullptr;
    noInlineJobs = std::getenv("MBAMD_NO_INLINE_JOBS") != nullptr;
    noRingCopy = std::getenv("MBAMD_NO_RING_COPY") != nullptr;
    eagerMatrices = s4 && std::getenv("MBAMD_LAZY_MATRICES") == nullptr;
    noInlinePrograms = std::getenv("MBAMD_NO_INLINE_PROGRAMS") != nullptr;
    noPath4 = std::getenv("MBAMD_NO_PATH4") != nullptr;
    noFusePath = std::getenv("MBAMD_NO_FUSE_PATH") != nullptr;
    noForkPath = std::getenv("MBAMD_NO_FORK_PATH") != nullptr;
    noPathG = std::getenv("MBAMD_NO_PATHG") != nullptr;
    if (const char* e = std::getenv("MBAMD_MFMA_SERIAL")) serialRatio = std::max(0, std::atoi(e));
    noSpine = std::getenv("MBAMD_NO_SPINE") != nullptr;
    if (const char* e = std::getenv("MBAMD_SPINE_WIDTH")) spineWidth = std::max(1, std::atoi(e));
    // serial / spine kernels exist for these shapes only (other category counts: level launches throughout)
    if (!((NT == 1 && (K == 1 || K == 2 || K == 4)) || (NT == 2 && (K == 1 || K == 2)))) serialRatio = 0;
    envVerbose = std::getenv("MBAMD_VERBOSE") != nullptr;
    envTrace = std::getenv("MBAMD_WALK_TRACE") != nullptr;
    if (std::getenv("MBAMD_REPORT_DEVICE")) {    // one line per instance: which physical GPU (multi-rank drivers collect them: bench.py mpi_mcmc)
        char bus[64] = "?";
        if (hipDeviceGetPCIBusId(bus, (int) sizeof bus, device) != hipSuccess) (void) hipGetLastError();
        const char* mr = std::getenv("MBAMD_MPI_RANK");
        std::fprintf(stderr, "[mbamd] instance on device %d pci %s mpi-rank %s\n", device, bus, mr ? mr : "-");
    }
    noSiteHost = std::getenv("MBAMD_NO_SITE_HOST") != nullptr;
    partialsFloats = s4 ? (size_t) K * Ppad * 4 : (size_t) K * S * Ppad;
    matrixFloats = (size_t) K * SP * SP + (mfma ? (size_t) K * NT * T * 64 : 0);
    if (wg) {
        wgTabFloats = (size_t) K * SP * SP;
        matrixFloats = wgTabFloats + (size_t) K * wg_table_floats(S);
    }
    if (arena()) {
        int rc = configureWalk();
        if (rc) return rc;
    }
    eigenDoubles = (size_t) 3 * S * S + S;       // [U | U^-1 | lambda | V]: V = orthonormal eigenvectors kept for warm starts (k_eigen_reversible)
    partials.assign(nBuffers, nullptr);
    tipStates.assign(nBuffers, nullptr);
    scale.assign(std::max(nScale, 1), nullptr);
    valid.assign(nBuffers, 0);
    if (s4) {
        // everything up front, like the reference's InitChainCondLikes (src/mcmc.c:5756-5834): one arena per kind.  Partials
        // are BUFFER-major, [buffer][block][K][64]: the waves of a launch run the same program at about the same pace, so at any
        // moment they all write into one node's few MB -- a moving window like a fill -- instead of into a 1 KiB piece each of
        // regions 12 MB apart (block-major, rounds 1-3: the same kernel ran C4 in 0.65 to 0.84 ms depending on the box; with the
        // stores in one window 0.67 on a slow one, profiles/r03_exp_walk4_linear.txt).  Tips and exponents stay block-major.
        const size_t nb = (size_t) Ppad / 64;
        if ((size_t) nBuffers * nb * K >= ((size_t) 1 << 32)) return fail(BEAGLE_ERROR_OUT_OF_MEMORY, "beagleCreateInstance: more than 4 TiB of partials");   // (program entries hold KiB offsets in 32 bits)
        geom.pstride = (unsigned long) K * 64;
        geom.tstride = (unsigned) nBuffers * 4;
        geom.sstride = 64;
        estride = (unsigned) (scale.size() + MBAMD_W4_SCRATCH_ROWS) * K * 64;       // + the scratch rows (sinks of operations that record no exponents, in rotation)
        const size_t pBytes = nb * (size_t) nBuffers * K * 64 * 16, tBytes = nb * geom.tstride * 8, eBytes = nb * (size_t) estride;
        HIP_TRY(hipMalloc(&arenaPartials, pBytes));
        HIP_TRY(hipMalloc(&arenaTips, tBytes));
        HIP_TRY(hipMalloc(&arenaExp, eBytes));
        HIP_TRY(hipMemsetAsync(arenaPartials, 0, pBytes, stream));
        HIP_TRY(hipMemsetAsync(arenaTips, 0xFF, tBytes, stream));            // (a tip never set = all states compatible)
        HIP_TRY(hipMemsetAsync(arenaExp, 0, eBytes, stream));
        wideScale.assign(scale.size(), nullptr);
        scaleState.assign(scale.size(), 0);
        if (std::getenv("MBAMD_VERBOSE"))
            std::fprintf(stderr, "[mbamd] arenas: partials %p +%zu, tips %p +%zu, exponents %p +%zu\n",
                         (void*) arenaPartials, pBytes, (void*) arenaTips, tBytes, (void*) arenaExp, eBytes);
        for (int i = 0; i < nBuffers; ++i) partials[i] = arenaPartials + (size_t) i * nb * K * 64 * 4;
    }
    if (wg) {
        // the same for the 20/61-state tree walk (mbamd_walkg.h): tile-major arenas, one extra partials buffer per tile as
        // the sink of NOP entries; exponents in the 4-state path's format (two tiles per 64-pattern block)
        const size_t nt = (size_t) Ppad / MBAMD_WG_TW, nb = (size_t) Ppad / 64, tb = wg_block_bytes(S);
        wgTileBytes = (unsigned long) (nBuffers + 1) * K * tb;
        wgTipTileBytes = (unsigned) nBuffers * MBAMD_WG_TW;
        // (+ the scratch rows: sinks of entries that do not record exponents.  The general-state kernels store an exponent byte with every
        //  entry -- a conditional store would make the compiler's counted waits stricter -- and every entry of a SCALE_READ evaluation
        //  storing to ONE row made such an evaluation 29 % slower at 20 states (same-address stores, profiles/r06_scale_read.txt): the
        //  entries of a program rotate over MBAMD_WG_SCRATCH_ROWS rows)
        estride = (unsigned) (scale.size() + MBAMD_WG_SCRATCH_ROWS) * K * 64;
        const size_t pBytes = nt * wgTileBytes, tBytes = nt * wgTipTileBytes, eBytes = nb * (size_t) estride;
        HIP_TRY(hipMalloc(&arenaPartials, pBytes));
        HIP_TRY(hipMalloc(&arenaTipStates, tBytes));
        HIP_TRY(hipMalloc(&arenaExp, eBytes));
        HIP_TRY(hipMemsetAsync(arenaPartials, 0, pBytes, stream));
        HIP_TRY(hipMemsetAsync(arenaTipStates, S, tBytes, stream));           // (a tip never set = missing data)
        HIP_TRY(hipMemsetAsync(arenaExp, 0, eBytes, stream));
        wideScale.assign(scale.size(), nullptr);
        scaleState.assign(scale.size(), 0);
        if (envVerbose)
            std::fprintf(stderr, "[mbamd] arenas: partials +%zu, tips +%zu, exponents +%zu bytes\n", pBytes, tBytes, eBytes);
        for (int i = 0; i < nBuffers; ++i) partials[i] = arenaPartials + (size_t) i * K * tb / 4;
    }

    HIP_TRY(hipMalloc(&matrices, std::max<size_t>(1, (size_t) nMatrices * matrixFloats) * sizeof(float)));
    HIP_TRY(hipMemsetAsync(matrices, 0, std::max<size_t>(1, (size_t) nMatrices * matrixFloats) * sizeof(float), stream));
    if (wg && nMatrices > 0) {                   // the constant "missing data" column of every gather table
        const int total = nMatrices * K * S;
        MBAMD_LAUNCH(k_wg_init_tables, (unsigned) ((total + 255) / 256), 256, 0, stream, matrices, matrixFloats, wgTabFloats, S, K, total);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMalloc(&d_eigen, std::max<size_t>(1, (size_t) nEigen * eigenDoubles) * sizeof(double)));
    HIP_TRY(hipMalloc(&d_freqs, std::max<size_t>(1, (size_t) nEigen * S) * sizeof(double)));
    HIP_TRY(hipMalloc(&d_weights, std::max<size_t>(1, (size_t) nEigen * K) * sizeof(double)));
    HIP_TRY(hipMalloc(&d_rates, (size_t) K * sizeof(double)));
    HIP_TRY(hipMalloc(&d_pweights, (size_t) Ppad * sizeof(double)));
    HIP_TRY(hipMalloc(&d_site, (size_t) Ppad * sizeof(double)));
    nblocks = Ppad / 64;
    if (!s4 && S >= 8) nblocks = Ppad / 32;      // k_integrate_lnl_wide: one block sum per 32-pattern tile
    if (wg) nblocks = Ppad / MBAMD_INTEGRATE_WG_PATTERNS;      // the tree-walk layout's integration kernel, whatever the state count
    HIP_TRY(hipHostMalloc(&h_sums, (size_t) nblocks * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipHostGetDevicePointer((void**) &h_sums_dev, h_sums, 0));
    if (std::getenv("MBAMD_NO_POLL") == nullptr) {
        if (hipHostMalloc((void**) &h_flag, 64, hipHostMallocDefault) == hipSuccess && hipHostGetDevicePointer((void**) &h_flag_dev, h_flag, 0) == hipSuccess) {
            *h_flag = 0;
            pollResult = true;
            pollSums = std::getenv("MBAMD_NO_SUM_POLL") == nullptr;
        } else {
            (void) hipGetLastError();
        }
    }
    stageCap = (size_t) 8 << 20;
    HIP_TRY(hipHostMalloc(&stage, stageCap, hipHostMallocDefault));
    HIP_TRY(hipHostGetDevicePointer((void**) &stage_dev, stage, 0));

    // defaults: unit rates, uniform category weights, unit pattern weights (BEAGLE clients normally set them)
    std::vector<double> ones(std::max(Ppad, K), 1.0);
    HIP_TRY(hipMemcpy(d_rates, ones.data(), (size_t) K * sizeof(double), hipMemcpyHostToDevice));
    rateSets.assign(1, RatesArg{});
    for (int k = 0; k < MBAMD_MAX_RATES; ++k) rateSets[0].r[k] = 1.0;
    std::vector<double> pw(Ppad, 0.0);
    std::fill(pw.begin(), pw.begin() + P, 1.0);
    HIP_TRY(hipMemcpy(d_pweights, pw.data(), (size_t) Ppad * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> w((size_t) std::max(1, nEigen) * K, 1.0 / K);
    HIP_TRY(hipMemcpy(d_weights, w.data(), (size_t) nEigen * K * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipStreamSynchronize(stream));
    return BEAGLE_SUCCESS;
}

void Instance::destroy()
{
    (void) hipSetDevice(device);
    (void) hipStreamSynchronize(stream);
    if (arena()) {
        void* arenas[] = {arenaPartials, arenaTips, arenaTipStates, arenaExp};
        for (void* a : arenas) if (a) (void) hipFree(a);
        for (int32_t* w : wideScale) if (w) (void) hipFree(w);
    } else {
        for (float* p : partials) if (p) (void) hipFree(p);
        for (uint8_t* p : tipStates) if (p) (void) hipFree(p);
        for (int32_t* p : scale) if (p) (void) hipFree(p);
    }
    pending.clear();
    wgOps.clear(); wgListStart.clear(); wgListCum.clear();
    for (Plan* pl : plans) { if (pl->d_table) (void) hipFree(pl->d_table); delete pl; }
    plans.clear();
    void* bufs[] = {matrices, d_eigen, d_freqs, d_weights, d_rates, d_pweights, d_site,
                    d_ev, d_tmp, d_trace};
    for (void* b : bufs) if (b) (void) hipFree(b);
    if (h_sums) (void) hipHostFree(h_sums);
    if (h_flag) (void) hipHostFree(h_flag);
    if (h_site) (void) hipHostFree(h_site);
    if (stage) (void) hipHostFree(stage);
    for (auto& ev : events) { (void) hipEventDestroy(ev.first); (void) hipEventDestroy(ev.second); }
    for (auto& ev : spans) { (void) hipEventDestroy(ev.first); (void) hipEventDestroy(ev.second); }
    if (spanOpen) (void) hipEventDestroy(spanEv0);
    if (reduceEvent) (void) hipEventDestroy(reduceEvent);
    for (auto& kv : finalExpOwn) if (kv.second) (void) hipFree(kv.second);
    (void) hipStreamDestroy(stream);
}

// ---------------------------------------------------------------------------------------------
// Tree-walk geometry.  The grid is (pattern blocks) x (categories) workgroups of W waves; each wave owns `slots` LDS
// slots of 1 KiB.  W and the slot count are chosen so that the whole grid is resident at once when the chip allows it:
// few blocks -> more tree parallelism per block, many blocks -> single-wave workgroups with deep slot stacks.
int Instance::configureWalk()
{
    int numCU = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) numCU = prop.multiProcessorCount;
    const int maxLds = 160 * 1024;
    if (s4 && (hipFuncSetAttribute((const void*) k_walk4_t<Walk4Args>, hipFuncAttributeMaxDynamicSharedMemorySize, maxLds) != hipSuccess ||
               hipFuncSetAttribute((const void*) k_walk4_t<Walk4ArgsInline>, hipFuncAttributeMaxDynamicSharedMemorySize, maxLds) != hipSuccess))
        (void) hipGetLastError();
    if (wg) {
        // one wave = (32-pattern tile, category); registers bound the residency: 20 states 4 waves per SIMD, 61 states 2
        const unsigned slotBytes = wg_block_bytes(S);
        MBAMD_WG_DISPATCH(S, raise_walkg_lds, maxLds);
        wgGeometry(1, w4.maxW, w4.maxSlots);
        w4.maxSlots1 = w4.maxSlots;
        if (!std::getenv("MBAMD_WALK_WAVES") && !std::getenv("MBAMD_MAX_LDS_SLOTS")) {   // a single-wave program may use the LDS of the whole workgroup
            const long wgsG = (long) (Ppad / MBAMD_WG_TW) * K;
            const int perCUG = (int) std::max(1L, (wgsG + numCU - 1) / numCU);
            w4.maxSlots1 = std::max(w4.maxSlots, std::min(24, (int) (((160 * 1024) / std::min(perCUG, 32) - 64 - MBAMD_WG_STAGE) / (int) slotBytes)));
        }
        w4.memSlots = false;
        w4.leadNops = MBAMD_WG_LEAD; w4.unroll = 3; w4.tailNops = MBAMD_WG_TAIL;
        w4.prefetchDistance = 0;
        if (const char* e = std::getenv("MBAMD_WALK_SMALL_PHASE")) w4.smallPhase = std::max(1, std::atoi(e));
        if (envVerbose) std::fprintf(stderr, "[mbamd] tree walk (%d states): %ld workgroups, up to %d waves x %d slots of %u bytes\n",
                                     S, (long) (Ppad / MBAMD_WG_TW) * K, w4.maxW, w4.maxSlots, slotBytes);
        return BEAGLE_SUCCESS;
    }
    const long wgs = (long) (Ppad / 64) * K;
    const int perCU = (int) std::max(1L, (wgs + numCU - 1) / numCU);          // workgroups a CU must host for full residency
    const int ldsPerWG = (160 * 1024) / std::min(perCU, 32) - 64;
    auto slotsFor = [&](int W) { return (ldsPerWG / W - MBAMD_W4_STAGE) / 1024; };
    // measured (profiles/): about 12-15 waves per CU (3-4 per SIMD) is the sweet spot -- fewer leave the scalar-load
    // latency uncovered, more cost LDS (slots) and tree-partition efficiency (phases, padding) without buying anything.
    // (Round 6, profiles/r06_walk4_waves.txt: DNA 500 x 20 000 = 4.9 workgroups per CU ran two waves each until then; with three
    //  -- 15 waves per CU, 190 entries per wave instead of 264 -- the evaluation takes 0.158-0.164 ms instead of 0.181; four: 0.192.)
    int W = (int) std::max(1L, std::min((long) MBAMD_W4_MAXW, (14L * numCU + wgs / 2) / wgs));
    while (W > 1 && slotsFor(W) < 7) --W;
    if (const char* e = std::getenv("MBAMD_WALK_WAVES")) W = std::max(1, std::min(MBAMD_W4_MAXW, std::atoi(e)));
    int slots = std::max(3, std::min(48, slotsFor(W)));
    if (const char* e = std::getenv("MBAMD_MAX_LDS_SLOTS")) slots = std::max(3, std::min(150 / W, std::atoi(e)));
    w4.maxW = W;
    w4.maxSlots = slots;
    w4.maxSlots1 = std::max(slots, std::min(40, slotsFor(1)));
    if (std::getenv("MBAMD_MAX_LDS_SLOTS")) w4.maxSlots1 = slots;
    if (const char* e = std::getenv("MBAMD_WALK_PREFETCH")) w4.prefetchDistance = std::max(0, std::atoi(e));
    w4.forward = std::getenv("MBAMD_WALK_NO_FORWARD") == nullptr;
    w4.safeWaits = std::getenv("MBAMD_WALK_SAFE") != nullptr;
    if (const char* e = std::getenv("MBAMD_WALK_SMALL_PHASE")) w4.smallPhase = std::max(1, std::atoi(e));
    if (envVerbose) std::fprintf(stderr, "[mbamd] tree walk: %ld workgroups (%d per CU), up to %d waves x %d slots\n", wgs, perCU, W, slots);
    return BEAGLE_SUCCESS;
}

// 20/61-state walk: waves per workgroup and LDS slots per wave for (tiles x categories x lists) workgroups.  Waves per
// workgroup are a power of two (two-wave workgroups are launched as four, see k_walkg; three or five leave SIMDs idle).
void Instance::wgGeometry(int lists, int& W, int& slots) const
{
    int numCU = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) numCU = prop.multiProcessorCount;
    // registers bound the residency: 4 (20 states) / 2 (61 states) waves per SIMD
    const int maxW = S > 32 ? 4 : 8, wavesPerCU = S > 32 ? 6 : 12;
    const int slotBytes = (int) wg_block_bytes(S);
    const long wgs = (long) (Ppad / MBAMD_WG_TW) * K * lists;
    const int perCU = (int) std::max(1L, (wgs + numCU - 1) / numCU);
    const int ldsPerWG = (160 * 1024) / std::min(perCU, 32) - 64;
    auto slotsFor = [&](int w) { return (ldsPerWG / w - MBAMD_WG_STAGE) / slotBytes; };
    long want = std::max(1L, std::min((long) maxW, ((long) wavesPerCU * numCU + wgs / 2) / wgs));
    W = 1;
    while (W * 2 <= want) W *= 2;
    while (W > 1 && slotsFor(W) < 4) W /= 2;
    if (const char* e = std::getenv("MBAMD_WALK_WAVES")) W = std::max(1, std::min(maxW, std::atoi(e)));
    slots = std::max(3, std::min(24, slotsFor(W)));
    if (const char* e = std::getenv("MBAMD_MAX_LDS_SLOTS")) slots = std::max(1, std::min((160 * 1024 / W - MBAMD_WG_STAGE) / slotBytes, std::atoi(e)));
}

// 4-state path: one tip's state masks (bit i = state i compatible) -> four 64-bit bitplanes per pattern block
int Instance::setTipMasks(int tip, const std::vector<uint8_t>& h)
{
    const size_t nb = (size_t) Ppad / 64;
    std::vector<uint64_t> planes(nb * 4, 0);
    for (int c = 0; c < Ppad; ++c)
        for (int i = 0; i < 4; ++i)
            if (h[c] >> i & 1u) planes[(size_t) (c >> 6) * 4 + i] |= (uint64_t) 1 << (c & 63);
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy2D(arenaTips + (size_t) tip * 4, (size_t) geom.tstride * 8, planes.data(), 32, 32, nb, hipMemcpyHostToDevice));
    if (!tipStates[tip]) layoutEpoch++;
    tipStates[tip] = reinterpret_cast<uint8_t*>(arenaTips + (size_t) tip * 4);
    return BEAGLE_SUCCESS;
}

int Instance::setTipStates(int tip, const int* states)
{
    if (tip < 0 || tip >= nBuffers) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetTipStates: tip index");
    std::vector<uint8_t> h(Ppad, (uint8_t) S);
    for (int c = 0; c < P; ++c) h[c] = (uint8_t) ((states[c] < 0 || states[c] >= S) ? S : states[c]);
    if (s4) {
        for (int c = 0; c < Ppad; ++c) h[c] = (uint8_t) (h[c] >= 4 ? 0xF : 1u << h[c]);   // state masks (mbamd_walk4.h)
        return setTipMasks(tip, h);
    }
    if (wg) {                                    // 32 state codes per (tile, tip) in the tip arena
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy2D(arenaTipStates + (size_t) tip * MBAMD_WG_TW, (size_t) wgTipTileBytes, h.data(), MBAMD_WG_TW, MBAMD_WG_TW, (size_t) Ppad / MBAMD_WG_TW, hipMemcpyHostToDevice));
        if (!tipStates[tip]) layoutEpoch++;
        tipStates[tip] = arenaTipStates + (size_t) tip * MBAMD_WG_TW;
        return BEAGLE_SUCCESS;
    }
    if (!tipStates[tip]) { HIP_TRY(hipMalloc(&tipStates[tip], (size_t) Ppad)); layoutEpoch++; }
    return upload(tipStates[tip], h.data(), (size_t) Ppad);
}

int Instance::importPartials(int idx, const double* in, bool hasCategories)
{
    if (idx < 0 || idx >= nBuffers) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "partials buffer index");
    if (idx < (int) finalExpOf.size()) finalExpOf[idx] = nullptr;
    if (s4 && !hasCategories) {
        // beagleSetTipPartials with 0/1 entries (IUPAC ambiguity codes, reference src/mbbeagle.c:150-166): a state mask
        // per pattern says the same thing in one byte, and the tree walk reads it like any compact tip
        std::vector<uint8_t> h(Ppad, 0xF);
        bool binary = true;
        for (int c = 0; c < P && binary; ++c) {
            unsigned m = 0;
            for (int i = 0; i < 4; ++i) {
                const double v = in[(size_t) c * 4 + i];
                if (v == 1.0) m |= 1u << i;
                else if (v != 0.0) binary = false;
            }
            h[c] = (uint8_t) m;
        }
        if (binary) return setTipMasks(idx, h);
    }
    int rc = ensurePartials(idx);
    if (rc) return rc;
    const size_t nIn = (size_t) (hasCategories ? K : 1) * P * S;
    rc = grow(&d_tmp, &tmpCap, nIn * sizeof(double));
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(d_tmp, in, nIn * sizeof(double), hipMemcpyHostToDevice));
    const size_t total = (size_t) K * P * S;
    const unsigned blocks = (unsigned) ((total + 255) / 256);
    if (s4) MBAMD_LAUNCH(k_import_partials<1>, blocks, 256, 0, stream, (const double*) d_tmp, hasCategories ? 1 : 0, S, K, P, Ppad, (size_t) geom.pstride, partials[idx]);
    else if (wg) MBAMD_LAUNCH(k_import_partials<2>, blocks, 256, 0, stream, (const double*) d_tmp, hasCategories ? 1 : 0, S, K, P, Ppad, (size_t) (wgTileBytes / 4), partials[idx]);
    else    MBAMD_LAUNCH(k_import_partials<0>, blocks, 256, 0, stream, (const double*) d_tmp, hasCategories ? 1 : 0, S, K, P, Ppad, (size_t) geom.pstride, partials[idx]);
    HIP_TRY(hipGetLastError());
    valid[idx] = 1;
    if (tipStates[idx]) {                        // a tip switches from compact to partials form
        HIP_TRY(hipStreamSynchronize(stream));
        if (!arena()) (void) hipFree(tipStates[idx]);
        tipStates[idx] = nullptr;
        layoutEpoch++;
    }
    return BEAGLE_SUCCESS;
}

int Instance::getPartials(int idx, double* out)
{
    if (idx < 0 || idx >= nBuffers || !valid[idx]) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleGetPartials: buffer");
    const size_t total = (size_t) K * P * S;
    int rc = grow(&d_tmp, &tmpCap, total * sizeof(double));
    if (rc) return rc;
    const unsigned blocks = (unsigned) ((total + 255) / 256);
    if (s4) MBAMD_LAUNCH(k_export_partials<1>, blocks, 256, 0, stream, (const float*) partials[idx], S, K, P, Ppad, (size_t) geom.pstride, (double*) d_tmp);
    else if (wg) MBAMD_LAUNCH(k_export_partials<2>, blocks, 256, 0, stream, (const float*) partials[idx], S, K, P, Ppad, (size_t) (wgTileBytes / 4), (double*) d_tmp);
    else    MBAMD_LAUNCH(k_export_partials<0>, blocks, 256, 0, stream, (const float*) partials[idx], S, K, P, Ppad, (size_t) geom.pstride, (double*) d_tmp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(out, d_tmp, total * sizeof(double), hipMemcpyDeviceToHost));
    return BEAGLE_SUCCESS;
}

// Eigen-systems from rate matrices (or exchangeabilities), computed on the device: k_eigen_reversible (mbamd_kernels.h).
// Nothing here waits for the device: the rate matrices travel through the pinned ring and are read by the kernel from there.
int Instance::setRateMatrices(int first, int count, const double* q, const double* pi, int mode, int warmFirst)
{
    if (count <= 0) return BEAGLE_SUCCESS;
    if (first < 0 || first + count > nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdSetRateMatrices: eigen index");
    if (warmFirst >= 0 && warmFirst + count > nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdSetRateMatricesFrom: warm-start eigen index");
    // (a block reads its source's V before it writes its own: the same range is fine, a shifted overlap would read a buffer a
    //  neighbouring block of the same launch is writing)
    if (warmFirst >= 0 && warmFirst != first && warmFirst < first + count && first < warmFirst + count)
        return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdSetRateMatricesFrom: the warm-start range overlaps the destination range with a shift");
    if (S > 64) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdSetRateMatrices: more than 64 states");
    for (int i = 0; i < S; ++i)
        if (!(pi[i] > 0.0)) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdSetRateMatrices: a state frequency is not positive (no symmetric form)");
    if (eigenWarm.size() != (size_t) nEigen) { eigenWarm.assign(nEigen, -1); eigenShield.assign(nEigen, 0); }
    const size_t qd = (size_t) count * S * S, bytes = (qd + S) * sizeof(double);
    const double* dq = nullptr;
    std::vector<double> h(qd + S);
    std::memcpy(h.data(), q, qd * sizeof(double));
    std::memcpy(h.data() + qd, pi, (size_t) S * sizeof(double));
    int rc;
    if (bytes + 64 <= stageCap / 2) {
        rc = stageDirect(h.data(), bytes, (const void**) &dq);
        if (rc) return rc;
    } else {                                                       // (many large matrices at once: a device copy)
        rc = grow(&d_tmp, &tmpCap, bytes);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(stream));                     // (d_tmp may still be read by an earlier import)
        HIP_TRY(hipMemcpy(d_tmp, h.data(), bytes, hipMemcpyHostToDevice));
        dq = reinterpret_cast<const double*>(d_tmp);
    }
    std::vector<EigenJob> jobs(count);
    for (int i = 0; i < count; ++i) {
        jobs[i].q = dq + (size_t) i * S * S;
        jobs[i].pi = dq + qd;
        jobs[i].out = d_eigen + (size_t) (first + i) * eigenDoubles;
        jobs[i].warm = nullptr;
        // a warm start re-uses the orthonormal basis of the source; every 64th call starts cold again (rounding drift of the basis)
        if (warmFirst >= 0 && eigenWarm[warmFirst + i] >= 0 && eigenWarm[warmFirst + i] < 64)
            jobs[i].warm = d_eigen + (size_t) (warmFirst + i) * eigenDoubles + (size_t) 2 * S * S + S;
        jobs[i].mode = mode & 1;
        jobs[i].pad_ = 0;
    }
    std::vector<int> warmAfter(count);
    for (int i = 0; i < count; ++i) warmAfter[i] = jobs[i].warm ? eigenWarm[warmFirst + i] + 1 : 0;
    const EigenJob* djobs = nullptr;
    rc = stageDirect(jobs.data(), sizeof(EigenJob) * count, (const void**) &djobs);
    if (rc) return rc;
    static std::vector<char> ldsRaised(64, 0);                      // per device: the attribute belongs to the device's code object
    if (device >= 0 && device < (int) ldsRaised.size() && !ldsRaised[device]) {
        if (hipFuncSetAttribute((const void*) k_eigen_reversible<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*) k_eigen_reversible<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void) hipGetLastError();
        ldsRaised[device] = 1;
    }
    // (beyond 32 states a step's 2 x 2 blocks are spread over 1 024 threads: four waves per SIMD hide the LDS round trips of a Jacobi step)
    if (S > 32 && std::getenv("MBAMD_EIGEN_256") == nullptr)
        MBAMD_LAUNCH_BARRIER(k_eigen_reversible<32>, (unsigned) count, 1024, eigen_lds_doubles(S) * sizeof(double), stream, djobs, S, 30);
    else
        MBAMD_LAUNCH_BARRIER(k_eigen_reversible<8>, (unsigned) count, 256, eigen_lds_doubles(S) * sizeof(double), stream, djobs, S, 30);
    HIP_TRY(hipGetLastError());
    // bookkeeping only once the launch is in the stream: a failure above leaves the buffers "cold" and unshielded
    for (int i = 0; i < count; ++i) {
        eigenWarm[first + i] = warmAfter[i];
        eigenShield[first + i] = (mode & 2) ? 1 : 0;                // (a rewrite without the shield bit clears a stale shield)
    }
    return BEAGLE_SUCCESS;
}

int Instance::setEigen(int idx, const double* U, const double* Ui, const double* lam)
{
    if (idx < 0 || idx >= nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetEigenDecomposition: eigen index");
    if (eigenShield.size() == (size_t) nEigen && eigenShield[idx]) {     // the device computed this one (mbamdSetRateMatricesFrom, mode bit 1)
        eigenShield[idx] = 0;
        return BEAGLE_SUCCESS;
    }
    if (eigenWarm.size() == (size_t) nEigen) eigenWarm[idx] = -1;
    std::vector<double> h((size_t) 2 * S * S + S);
    std::memcpy(h.data(), U, sizeof(double) * S * S);
    std::memcpy(h.data() + (size_t) S * S, Ui, sizeof(double) * S * S);
    std::memcpy(h.data() + (size_t) 2 * S * S, lam, sizeof(double) * S);
    return upload(d_eigen + (size_t) idx * eigenDoubles, h.data(), h.size() * sizeof(double));
}

// beagleUpdateTransitionMatrices only queues its jobs: MrBayes calls it once per eigen-system part (reference
// src/mbbeagle.c:1475-1486), and all parts of an evaluation go out as ONE launch when the next other call arrives.
int Instance::setRates(int index, const double* r)
{
    if (index < 0 || index > 65535) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "category rates: index");
    if (K > MBAMD_MAX_RATES) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "more than 16 rate categories");
    if ((size_t) index >= rateSets.size()) rateSets.resize((size_t) index + 1, rateSets[0]);
    for (int k = 0; k < K; ++k) rateSets[index].r[k] = r[k];
    return BEAGLE_SUCCESS;
}

int Instance::updateMatrices(int eigenIndex, const int* probIdx, const double* lengths, int count, int rateSet)
{
    if (rateSet < 0 || (size_t) rateSet >= rateSets.size()) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdateTransitionMatrices: category rates index");
    if (!pendingJobs.empty() && rateSet != pendingRateSet) {       // one rate set per launch
        int frc = flushMatrices();
        if (frc) return frc;
    }
    pendingRateSet = rateSet;
    if (eigenIndex < 0 || eigenIndex >= nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdateTransitionMatrices: eigen index");
    if (count <= 0) return BEAGLE_SUCCESS;
    if (K > MBAMD_MAX_RATES) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "more than 16 rate categories");
    for (int i = 0; i < count; ++i)
        if (probIdx[i] < 0 || probIdx[i] >= nMatrices) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdateTransitionMatrices: matrix index");
    if (pendingMatrixOut.size() != (size_t) nMatrices) pendingMatrixOut.assign(nMatrices, 0);
    bool clash = false;
    for (int i = 0; i < count && !clash; ++i) clash = pendingMatrixOut[probIdx[i]] != 0;
    if (clash || (pendingJobs.size() + count) * sizeof(MatrixJob) > stageCap / 4) {
        int rc = flushMatrices();
        if (rc) return rc;
    }
    const double* eig = d_eigen + (size_t) eigenIndex * eigenDoubles;
    for (int i = 0; i < count; ++i) {
        MatrixJob j;
        j.out = matrixPtr(probIdx[i]);
        j.length = lengths[i];
        j.eig = eig;
        j.pad_ = 0.0;
        pendingJobs.push_back(j);
        pendingMatrixOut[probIdx[i]] = 1;
    }
    if (eagerMatrices) return flushMatrices();
    return BEAGLE_SUCCESS;
}

int Instance::flushMatrices()
{
    if (pendingJobs.empty()) return BEAGLE_SUCCESS;
    const RatesArg rates = rateSets[pendingRateSet];
    const int count = (int) pendingJobs.size();
    { int src = spanBegin(); if (src) return src; }
    if (s4 && count <= MBAMD_S4_INLINE_JOBS && count * K <= 64 && !noInlineJobs) {
        // a branch move's one or two matrices: the jobs in the kernel arguments (mbamd_kernels.h)
        MatrixJobs4 ja;
        std::memset(&ja, 0, sizeof ja);
        std::memcpy(ja.j, pendingJobs.data(), sizeof(MatrixJob) * count);
        pendingJobs.clear();
        std::fill(pendingMatrixOut.begin(), pendingMatrixOut.end(), 0);
        MBAMD_LAUNCH(k_transition_matrices_s4_inline, 1u, 64, 0, stream, ja, rates, K, count * K);
        HIP_TRY(hipGetLastError());
        return BEAGLE_SUCCESS;
    }
    const MatrixJob* djobs = nullptr;
    int rc = stageDirect(pendingJobs.data(), sizeof(MatrixJob) * count, (const void**) &djobs);
    pendingJobs.clear();
    std::fill(pendingMatrixOut.begin(), pendingMatrixOut.end(), 0);
    if (rc) return rc;
    if (s4) {
        const int total = count * K;
        MBAMD_LAUNCH(k_transition_matrices_s4, (unsigned) ((total + 255) / 256), 256, 0, stream, djobs, rates, K, total);
        HIP_TRY(hipGetLastError());
        return BEAGLE_SUCCESS;
    }
    if (S > 8 && S <= 64) {                       // fp64 matrix cores, one wave per 16 rows
        const unsigned grid = (unsigned) (count * K);
        const int packedT = mfma ? T : 0;
        const size_t wgTab = wg ? wgTabFloats : 0;
        switch ((S + 15) / 16) {
            case 1: MBAMD_LAUNCH_BARRIER(k_transition_matrices_mfma<1>, grid, 64, 0, stream, djobs, rates, S, SP, K, packedT, wgTab); break;
            case 2: MBAMD_LAUNCH_BARRIER(k_transition_matrices_mfma<2>, grid, 128, 0, stream, djobs, rates, S, SP, K, packedT, wgTab); break;
            case 3: MBAMD_LAUNCH_BARRIER(k_transition_matrices_mfma<3>, grid, 192, 0, stream, djobs, rates, S, SP, K, packedT, wgTab); break;
            default: MBAMD_LAUNCH_BARRIER(k_transition_matrices_mfma<4>, grid, 256, 0, stream, djobs, rates, S, SP, K, packedT, wgTab); break;
        }
        HIP_TRY(hipGetLastError());
        return BEAGLE_SUCCESS;
    }
    const int threads = std::min(256, round_up(S * S, 64));
    const double* evs = nullptr;                 // up to 64 states the matrix kernel forms the exponentials itself
    if (S > 64) {
        const size_t nev = (size_t) count * K * S;
        rc = grow((void**) &d_ev, &evCap, nev * sizeof(double));
        if (rc) return rc;
        MBAMD_LAUNCH(k_eigen_exponentials, (unsigned) ((nev + 255) / 256), 256, 0, stream, djobs, rates, S, K, (int) nev, d_ev);
        evs = d_ev;
    }
    MBAMD_LAUNCH_BARRIER(k_transition_matrices_ev, (unsigned) (count * K), threads, 0, stream, djobs, evs, rates, S, SP, K, 1,
                         mfma ? T : 0, wg ? wgTabFloats : (size_t) 0);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int Instance::setMatrix(int idx, const double* in)
{
    if (idx < 0 || idx >= nMatrices) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetTransitionMatrix: matrix index");
    std::vector<float> h(matrixFloats, 0.0f);
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j) {
                const float v = (float) in[((size_t) k * S + i) * S + j];
                h[(size_t) k * SP * SP + (size_t) j * SP + i] = v;
                if (mfma)
                    h[(size_t) K * SP * SP + ((size_t) (k * NT + i / 32) * T + j / 2) * 64 + (i % 32) + 32 * (j % 2)] = v;
                if (wg) wg_table_put(h.data() + wgTabFloats + (size_t) k * wg_table_floats(S), S, i, j, v);
            }
    if (wg)
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < S; ++i) wg_table_put_missing(h.data() + wgTabFloats + (size_t) k * wg_table_floats(S), S, i);
    return upload(matrixPtr(idx), h.data(), matrixFloats * sizeof(float));
}

int Instance::getMatrix(int idx, double* out)
{
    if (idx < 0 || idx >= nMatrices) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleGetTransitionMatrix: matrix index");
    std::vector<float> h(matrixFloats);
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(h.data(), matrixPtr(idx), matrixFloats * sizeof(float), hipMemcpyDeviceToHost));
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j)
                out[((size_t) k * S + i) * S + j] = h[(size_t) k * SP * SP + (size_t) j * SP + i];
    return BEAGLE_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// beagleUpdatePartials: resolve buffer indices to device pointers, then hand the list to the
// 4-state tree-walk kernel or to the level-synchronous general kernels.
// ---------------------------------------------------------------------------------------------
int Instance::updatePartials(const BeagleOperation* ops, int n, int cumIdx)
{
    if (n <= 0) return BEAGLE_SUCCESS;
    if (cumIdx != BEAGLE_OP_NONE && (cumIdx < 0 || cumIdx >= nScale))
        return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: cumulative scale index");
    if (!finalExpOf.empty())                     // a buffer an operation overwrites no longer holds final partials
        for (int o = 0; o < n; ++o)
            if (ops[o].destinationPartials >= 0 && ops[o].destinationPartials < nBuffers) finalExpOf[ops[o].destinationPartials] = nullptr;
    if (s4) return updatePartials4(ops, n, cumIdx);
    if (wg) return updatePartialsG(ops, n, cumIdx);
    int32_t* cumPtr = nullptr;
    if (cumIdx != BEAGLE_OP_NONE) {
        int rc = ensureScale(cumIdx);
        if (rc) return rc;
        cumPtr = scale[cumIdx];
    }
    // ---- plan cache ------------------------------------------------------------------------
    static_assert(sizeof(BeagleOperation) == 7 * sizeof(int), "BeagleOperation is 7 ints");
    const int* raw = reinterpret_cast<const int*>(ops);
    const size_t nints = (size_t) n * 7;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < nints; ++i) h = (h ^ (uint64_t) (uint32_t) raw[i]) * 1099511628211ull;
    h = (h ^ (uint64_t) (uint32_t) layoutEpoch) * 1099511628211ull;
    for (Plan* pl : plans)
        if (pl->hash == h && pl->key.size() == nints + 1 && pl->key[nints] == layoutEpoch &&
            std::memcmp(pl->key.data(), raw, nints * sizeof(int)) == 0) {
            pl->lastUse = ++planClock;
            planHits++;
            return submit(pl, cumIdx, cumPtr);
        }
    planMisses++;
    if (!mfma || mfmaWhole || noDefer || pending.empty()) {
        // launch the queued transition-matrix jobs now: the kernel runs while the host compiles the list
        int mrc = flushMatrices();
        if (mrc) return mrc;
    }
    std::vector<PartialsOp> dev(n);
    std::vector<int> dstIdx(n), c1Idx(n), c2Idx(n);
    std::vector<char> written(nBuffers, 0);
    for (int o = 0; o < n; ++o) {
        const BeagleOperation& b = ops[o];
        PartialsOp& d = dev[o];
        std::memset(&d, 0, sizeof d);
        if (b.destinationPartials < 0 || b.destinationPartials >= nBuffers || b.child1Partials < 0 ||
            b.child1Partials >= nBuffers || b.child2Partials < 0 || b.child2Partials >= nBuffers)
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: partials index");
        if (b.child1TransitionMatrix < 0 || b.child1TransitionMatrix >= nMatrices || b.child2TransitionMatrix < 0 ||
            b.child2TransitionMatrix >= nMatrices)
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: matrix index");
        int rc = ensurePartials(b.destinationPartials);
        if (rc) return rc;
        d.dst = partials[b.destinationPartials];
        const int ci[2] = {b.child1Partials, b.child2Partials};
        const void* cp[2];
        uint8_t ck[2];
        for (int s = 0; s < 2; ++s) {
            if (tipStates[ci[s]] && !written[ci[s]]) {
                cp[s] = tipStates[ci[s]];
                ck[s] = CHILD_STATES;
            } else {
                if (!valid[ci[s]] && !written[ci[s]]) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: child buffer was never written");
                cp[s] = partials[ci[s]];
                ck[s] = CHILD_PARTIALS;
            }
        }
        d.c1 = cp[0]; d.c1_kind = ck[0];
        d.c2 = cp[1]; d.c2_kind = ck[1];
        d.m1 = matrixPtr(b.child1TransitionMatrix);
        d.m2 = matrixPtr(b.child2TransitionMatrix);
        d.c1_slot = d.c2_slot = d.dst_slot = MBAMD_NO_SLOT;
        d.scale_mode = SCALE_NONE;
        if (b.destinationScaleWrite != BEAGLE_OP_NONE) {
            if (b.destinationScaleWrite < 0 || b.destinationScaleWrite >= nScale)
                return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: scale write index");
            rc = ensureScale(b.destinationScaleWrite);
            if (rc) return rc;
            d.scale = scale[b.destinationScaleWrite];
            d.scale_mode = SCALE_WRITE;
        } else if (b.destinationScaleRead != BEAGLE_OP_NONE) {
            if (b.destinationScaleRead < 0 || b.destinationScaleRead >= nScale)
                return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: scale read index");
            rc = ensureScale(b.destinationScaleRead);
            if (rc) return rc;
            d.scale = scale[b.destinationScaleRead];
            d.scale_mode = SCALE_READ;
        }
        dstIdx[o] = b.destinationPartials;
        c1Idx[o] = b.child1Partials;
        c2Idx[o] = b.child2Partials;
        written[b.destinationPartials] = 1;
    }
    for (int o = 0; o < n; ++o) valid[dstIdx[o]] = 1;
    // ---- compile the list into a plan (evicting the least recently used one) ----------------------
    Plan* plan;
    const size_t maxPlans = 24;
    if (plans.size() < maxPlans) {
        plan = new Plan();
        plans.push_back(plan);
    } else {
        plan = plans[0];
        for (Plan* pl : plans) if (pl->lastUse < plan->lastUse) plan = pl;
        for (auto& pp : pending)
            if (pp.first == plan) { int frc = flushPending(); if (frc) return frc; break; }
    }
    plan->key.assign(raw, raw + nints);
    plan->key.push_back(layoutEpoch);
    plan->hash = h;
    plan->lastUse = ++planClock;
    int rc;
    {
        StatTimer st_(ST_PLAN);
        rc = buildGeneric(*plan, dev, dstIdx, c1Idx, c2Idx);
    }
    if (rc) { plan->hash = 0; plan->key.clear(); return rc; }
    plan->bufsRead.assign(c1Idx.begin(), c1Idx.end());
    plan->bufsRead.insert(plan->bufsRead.end(), c2Idx.begin(), c2Idx.end());
    plan->bufsWritten.assign(dstIdx.begin(), dstIdx.end());
    plan->scalesUsed.clear();
    for (int o = 0; o < n; ++o) {
        if (ops[o].destinationScaleWrite != BEAGLE_OP_NONE) plan->scalesUsed.push_back(ops[o].destinationScaleWrite);
        if (ops[o].destinationScaleRead != BEAGLE_OP_NONE) plan->scalesUsed.push_back(ops[o].destinationScaleRead);
    }
    return submit(plan, cumIdx, cumPtr);
}

// Run a compiled list now, or -- general-state MFMA path -- defer it: consecutive mutually independent lists
// (one per eigen-system part, reference src/mbbeagle.c:1062-1104) are executed together, one launch per
// dependency level over all of them, when the next call that is not a beagleUpdatePartials arrives.
int Instance::submit(Plan* plan, int cumIdx, int32_t* cumPtr)
{
    {
        int mrc = flushMatrices();
        if (mrc) return mrc;
    }
    if (!s4 && mfma && !mfmaWhole && !noDefer) {
        if ((int) pending.size() >= MBAMD_MAX_TABLES || !independentOfPending(*plan, cumIdx)) {
            int rc = flushPending();
            if (rc) return rc;
        }
        pending.emplace_back(plan, cumIdx);
        return BEAGLE_SUCCESS;
    }
    (void) cumIdx;
    return timedRun(*plan, cumPtr);
}

bool Instance::independentOfPending(const Plan& plan, int cumIdx)
{
    if (pending.empty()) return true;
    std::vector<char> wr(nBuffers, 0), rd(nBuffers, 0), sc(scale.size(), 0);
    for (auto& pp : pending) {
        if (pp.first == &plan) return false;
        for (int b : pp.first->bufsWritten) wr[b] = 1;
        for (int b : pp.first->bufsRead) rd[b] = 1;
        for (int i : pp.first->scalesUsed) sc[i] = 1;
        if (pp.second >= 0) sc[pp.second] = 1;
    }
    for (int b : plan.bufsWritten) if (wr[b] || rd[b]) return false;
    for (int b : plan.bufsRead) if (wr[b]) return false;
    for (int i : plan.scalesUsed) if (sc[i]) return false;
    if (cumIdx >= 0 && sc[cumIdx]) return false;
    return true;
}

int Instance::runHeldPath()
{
    Plan* plan = heldPath;
    heldPath = nullptr;
    if (!plan) return BEAGLE_SUCCESS;
    walkCumFresh = heldPathFresh;
    return timedRun(*plan, heldPathCum);
}

int Instance::flushPending(bool keepPath)
{
    StatTimer st_(ST_FLUSH);
    int mrc = flushMatrices();                   // (queued matrix jobs precede the lists that read them)
    if (mrc) return mrc;
    if (heldPath && !keepPath) { int prc = runHeldPath(); if (prc) return prc; }
    if (wg) return flushWalkG();
    if (pending.empty()) return BEAGLE_SUCCESS;
    std::vector<std::pair<Plan*, int>> work;
    work.swap(pending);
    ++launchClock;
    for (auto& w : work) w.first->lastLaunch = launchClock;
    auto cumOf = [&](int idx) { return idx >= 0 ? scale[idx] : (int32_t*) nullptr; };
    // merged launches need a per-factor-tile kernel for this shape (launch_mfma_split); other shapes -- e.g. three rate
    // categories -- run their lists one after the other, in submission order
    const bool mergeable = (NT == 1 && (K == 1 || K == 2 || K == 4)) || (NT == 2 && (K == 1 || K == 2));
    if (work.size() == 1 || !mergeable) {
        for (auto& w : work) {
            const int rc = timedRun(*w.first, cumOf(w.second));
            if (rc) return rc;
        }
        return BEAGLE_SUCCESS;
    }
    hipEvent_t ev0{}, ev1{};
    { int src = spanBegin(); if (src) return src; }
    if (timing) {
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipEventRecord(ev0, stream));
    }
    size_t maxLevels = 0;
    bool allNarrow = true;
    for (auto& w : work) {
        maxLevels = std::max(maxLevels, w.first->start.size() - 1);
        allNarrow = allNarrow && w.first->narrow;
    }
    if (allNarrow) {                             // every list is a set of root-ward paths: one launch walks them all
        size_t nchains = 0;
        for (auto& w : work) nchains += w.first->chains.size();
        OpTables tabs;
        std::memset(&tabs, 0, sizeof tabs);
        int t = 0;
        for (auto& w : work) {
            if (nchains <= MBAMD_MAX_TABLES) {
                for (auto& ch : w.first->chains) {
                    tabs.ops[t] = w.first->d_table + ch.first;
                    tabs.cum[t] = cumOf(w.second);
                    tabs.start[t] = ch.second;
                    ++t;
                }
            } else {                             // too many sub-lists: each list in its own order
                tabs.ops[t] = w.first->d_table;
                tabs.cum[t] = cumOf(w.second);
                tabs.start[t] = w.first->start.back();
                ++t;
            }
        }
        if (launch_mfma_serial(*this, tabs, t)) {
            pendingLaunches += 1;
            maxLevels = 0;
        }
    }
    // levels every list runs as level launches; from spineFrom on each list is a spine of single operations
    size_t spineFrom = 0;
    for (auto& w : work) spineFrom = std::max(spineFrom, (size_t) w.first->serialFrom);
    if (maxLevels > 0) {
        int spineOps = 0;
        for (auto& w : work) spineOps += std::max(0, w.first->start.back() - w.first->start[std::min(spineFrom, w.first->start.size() - 1)]);
        if (spineOps < 2) spineFrom = maxLevels;
    }
    for (size_t l = 0; l < std::min(maxLevels, spineFrom); ++l) {
        OpTables tabs;
        std::memset(&tabs, 0, sizeof tabs);
        int t = 0, total = 0;
        bool tipsDone = false;
        if (l == 0) {                            // operations on two compact tips: their own kernel
            for (auto& w : work) {
                if (w.first->tipTip == 0) continue;
                tabs.ops[t] = w.first->d_table;
                tabs.cum[t] = cumOf(w.second);
                tabs.start[t] = total;
                total += w.first->tipTip;
                ++t;
            }
            for (int u = t; u <= MBAMD_MAX_TABLES; ++u) tabs.start[u] = 1 << 30;
            if (total > 0 && launch_tips(*this, tabs, total)) {
                pendingLaunches += 1;
                tipsDone = true;
            }
            std::memset(&tabs, 0, sizeof tabs);
            t = 0;
            total = 0;
        }
        for (auto& w : work) {
            const std::vector<int>& st = w.first->start;
            if (l + 1 >= st.size() || st[l + 1] == st[l]) continue;
            const int skip = (l == 0 && tipsDone) ? w.first->tipTip : 0;
            if (st[l + 1] - st[l] - skip == 0) continue;
            tabs.ops[t] = w.first->d_table + st[l] + skip;
            tabs.cum[t] = cumOf(w.second);
            tabs.start[t] = total;
            total += st[l + 1] - st[l] - skip;
            ++t;
        }
        for (int u = t; u <= MBAMD_MAX_TABLES; ++u) tabs.start[u] = 1 << 30;
        if (total == 0) continue;
        if (!launch_mfma_split(*this, tabs, total)) return fail(BEAGLE_ERROR_GENERAL, "no MFMA kernel for a deferred list");
        pendingLaunches += 1;
    }
    if (spineFrom < maxLevels) {
        OpTables tabs;
        std::memset(&tabs, 0, sizeof tabs);
        int t = 0;
        for (auto& w : work) {
            const std::vector<int>& st = w.first->start;
            if (spineFrom + 1 >= st.size()) continue;
            tabs.ops[t] = w.first->d_table + st[spineFrom];
            tabs.cum[t] = cumOf(w.second);
            tabs.start[t] = st.back() - st[spineFrom];
            ++t;
        }
        if (t > 0) {
            if (!launch_mfma_serial(*this, tabs, t)) return fail(BEAGLE_ERROR_GENERAL, "no serial MFMA kernel for a deferred list");
            pendingLaunches += 1;
        }
    }
    HIP_TRY(hipGetLastError());
    if (timing) {
        HIP_TRY(hipEventRecord(ev1, stream));
        events.emplace_back(ev0, ev1);
    }
    return BEAGLE_SUCCESS;
}

// upload a freshly built table into the plan's own device buffer
int Instance::planTable(Plan& plan, const std::vector<PartialsOp>& table)
{
    const size_t bytes = table.size() * sizeof(PartialsOp);
    const bool inFlight = plan.lastLaunch > syncedClock;     // the old table may still be read by a running kernel
    if (bytes > plan.cap) {
        if (inFlight) { HIP_TRY(hipStreamSynchronize(stream)); syncedClock = launchClock; }
        if (plan.d_table) HIP_TRY(hipFree(plan.d_table));
        plan.d_table = nullptr;
        plan.cap = 0;
        HIP_TRY(hipMalloc(&plan.d_table, bytes + bytes / 2));
        plan.cap = bytes + bytes / 2;
    } else if (inFlight) {
        HIP_TRY(hipStreamSynchronize(stream));
        syncedClock = launchClock;
    }
    return upload(plan.d_table, table.data(), bytes);
}

int Instance::timedRun(const Plan& plan, int32_t* cum)
{
    const_cast<Plan&>(plan).lastLaunch = ++launchClock;
    hipEvent_t ev0{}, ev1{};
    { int src = spanBegin(); if (src) return src; }
    if (timing) {
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipEventRecord(ev0, stream));
    }
    const int rc = s4 ? runWalk(plan, cum) : (wg ? runWalkG(plan) : runGeneric(plan, cum));
    if (timing) {
        HIP_TRY(hipEventRecord(ev1, stream));
        events.emplace_back(ev0, ev1);
        if (events.size() > 4096) {               // a client that never asks: fold the finished ones into the running total
            HIP_TRY(hipStreamSynchronize(stream));
            spanFold();
            for (auto& ev : events) {
                float t = 0.0f;
                if (hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess) timedMs += t;
                (void) hipEventDestroy(ev.first);
                (void) hipEventDestroy(ev.second);
            }
            events.clear();
        }
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------
// 4-state path: beagleUpdatePartials -> per-wave programs of the tree-walk kernel (mbamd_walk4.h).
// ---------------------------------------------------------------------------------------------
int Instance::ensureWide(int idx)
{
    if (!wideScale[idx]) {
        HIP_TRY(hipMalloc(&wideScale[idx], (size_t) K * Ppad * sizeof(int32_t)));
        if (scaleState[idx] != 1) HIP_TRY(hipMemsetAsync(wideScale[idx], 0, (size_t) K * Ppad * sizeof(int32_t), stream));
    }
    if (scaleState[idx] == 1) {                  // node exponents so far: a cumulative buffer continues from them
        MBAMD_LAUNCH(k_exp_widen, (unsigned) (((size_t) K * Ppad + 255) / 256), 256, 0, stream, (const int8_t*) arenaExp, estride, idx, K, Ppad,
                     wideScale[idx]);
        HIP_TRY(hipGetLastError());
    } else if (scaleState[idx] == 0) {
        HIP_TRY(hipMemsetAsync(wideScale[idx], 0, (size_t) K * Ppad * sizeof(int32_t), stream));
    }
    scaleState[idx] = 2;
    return BEAGLE_SUCCESS;
}

int Instance::updatePartials4(const BeagleOperation* ops, int n, int cumIdx)
{
    if (heldPath) { int prc = runHeldPath(); if (prc) return prc; }      // (a list behind a held path: the path runs first)
    int32_t* cumPtr = nullptr;
    walkCumFresh = false;
    if (cumIdx != BEAGLE_OP_NONE) {
        if (scaleState[cumIdx] == 0) {
            // a freshly reset cumulative buffer (the rescale-everything pass): the kernel STORES its sums, no zero-fill launch
            if (!wideScale[cumIdx]) HIP_TRY(hipMalloc(&wideScale[cumIdx], (size_t) K * Ppad * sizeof(int32_t)));
            scaleState[cumIdx] = 2;
            walkCumFresh = true;
        } else {
            int rc = ensureWide(cumIdx);
            if (rc) return rc;
        }
        cumPtr = wideScale[cumIdx];
    }
    // ---- plan cache ------------------------------------------------------------------------
    const int* raw = reinterpret_cast<const int*>(ops);
    const size_t nints = (size_t) n * 7;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < nints; ++i) h = (h ^ (uint64_t) (uint32_t) raw[i]) * 1099511628211ull;
    h = (h ^ (uint64_t) (uint32_t) layoutEpoch) * 1099511628211ull;
    Plan* plan = nullptr;
    for (Plan* pl : plans)
        if (pl->hash == h && pl->key.size() == nints + 1 && pl->key[nints] == layoutEpoch &&
            std::memcmp(pl->key.data(), raw, nints * sizeof(int)) == 0) {
            pl->lastUse = ++planClock;
            planHits++;
            plan = pl;
            break;
        }
    if (!plan) {
        planMisses++;
        int mrc = flushMatrices();               // the matrix kernel runs while the host compiles the list
        if (mrc) return mrc;
        const size_t maxPlans = 24;
        if (plans.size() < maxPlans) {
            plan = new Plan();
            plans.push_back(plan);
        } else {
            plan = plans[0];
            for (Plan* pl : plans) if (pl->lastUse < plan->lastUse) plan = pl;
        }
        plan->key.assign(raw, raw + nints);
        plan->key.push_back(layoutEpoch);
        plan->hash = h;
        plan->lastUse = ++planClock;
        int rc;
        {
            StatTimer st_(ST_PLAN);
            plan->path = plan->forked = false;
            rc = buildPath4(*plan, ops, n) ? BEAGLE_SUCCESS : buildWalk(*plan, ops, n);
        }
        if (rc) { plan->hash = 0; plan->key.clear(); return rc; }
    }
    // bookkeeping the list implies, whether compiled now or before: destinations valid, exponent buffers in node form
    for (int o = 0; o < n; ++o) {
        valid[ops[o].destinationPartials] = 1;
        if (ops[o].destinationScaleWrite != BEAGLE_OP_NONE) scaleState[ops[o].destinationScaleWrite] = 1;
    }
    int mrc = flushMatrices();
    if (mrc) return mrc;
    listsTotal++;
    if (plan->path) { listsPath++; if (plan->forked) forkedPaths++; }
    else { listsWalked++; opsWalked += n; }
    if (plan->path && !noFusePath && K <= 8 && plan->inlineProg.size() <= MBAMD_W4_INLINE) {
        // hold it: the next call decides (runHeldPath / integratePath4)
        heldPath = plan;
        heldPaths++;
        heldPathCum = cumPtr;
        heldPathFresh = walkCumFresh;
        heldPathDst = ops[n - 1].destinationPartials;
        return BEAGLE_SUCCESS;
    }
    return timedRun(*plan, cumPtr);
}

// Compile one operation list: validate, cut into hazard-free segments, build (or re-use) the structural template of
// each segment and fill it with this list's buffer / matrix / scale indices.
int Instance::buildWalk(Plan& plan, const BeagleOperation* ops, int n, const int* listOf, bool perList)
{
    const int scratchScale = (int) scale.size();              // sink / source of entries that do not rescale
    std::vector<int>& segList = w4segList;                     // (20/61-state walk) merged-list index of each operation of the segment
    segList.clear();
    std::vector<char>& written = w4written;
    written.assign((size_t) nBuffers, 0);
    w4table.clear();
    plan.segments.clear();
    std::vector<Walk4Op>& seg = w4ops;
    seg.clear();
    // segment state: buffers / exponent buffers the current segment has read or written
    std::vector<char>&segRead = w4segRead, &segWritten = w4segWritten, &segScale = w4segScale;
    segRead.assign((size_t) nBuffers, 0); segWritten.assign((size_t) nBuffers, 0); segScale.assign(scale.size() + 1, 0);
    if (w4writer.size() < (size_t) nBuffers) w4writer.assign((size_t) nBuffers, -1);
    int reloads = 0, externals = 0, phases = 0;
    auto flushSegment = [&]() -> int {
        if (seg.empty()) return BEAGLE_SUCCESS;
        // structural key
        std::vector<int>& key = w4key;
        key.clear();
        key.reserve(seg.size() * 3 + 4);
        key.push_back((int) seg.size()); key.push_back(w4.maxW); key.push_back(w4.maxSlots + 256 * w4.maxSlots1);
        key.push_back(w4.prefetchDistance * 2 + (w4.safeWaits ? 1 : 0) + (w4.forward ? 512 : 0));
        {
            std::vector<int>& writer = w4writer;          // buffer -> operation of this segment that writes it (-1 outside this block)
            for (size_t o = 0; o < seg.size(); ++o) {
                key.push_back(seg[o].tip1 ? -1 : writer[seg[o].c1]);
                key.push_back(seg[o].tip2 ? -1 : writer[seg[o].c2]);
                key.push_back((int) seg[o].tip1 | ((int) seg[o].tip2 << 1) | ((!seg[o].tip1 && !seg[o].tip2 && seg[o].c1 == seg[o].c2) ? 4 : 0) |
                              ((seg[o].scaleWrite < 0 && seg[o].scaleRead >= 0) ? 8 : 0));   // (SCALE_READ entries wait for an exponent DMA)
                writer[seg[o].dst] = (int) o;
            }
            for (size_t o = 0; o < seg.size(); ++o) writer[seg[o].dst] = -1;
        }
        uint64_t kh = 1469598103934665603ull;
        for (int v : key) kh = (kh ^ (uint64_t) (uint32_t) v) * 1099511628211ull;
        auto it = w4templates.find(kh);
        if (it == w4templates.end() || it->second.key != key) {
            scheduleMisses++;
            if (w4templates.size() >= 8192) w4templates.clear();
            Walk4Template& t = w4templates[kh];
            bool ok = w4.build(seg, t);
            if (!ok) {                           // out of slots with look-ahead prefetches: retry without, then on one wave
                Walk4Builder plain = w4;
                plain.prefetchDistance = 0;
                ok = plain.build(seg, t);
                if (!ok) { plain.maxW = 1; ok = plain.build(seg, t); }
            }
            if (!ok) { w4templates.erase(kh); return fail(BEAGLE_ERROR_GENERAL, "tree-walk scheduler: cannot place this list"); }
            t.key = key;
            it = w4templates.find(kh);
        } else {
            scheduleHits++;
        }
        const Walk4Template& t = it->second;
        reloads += t.reloads; externals += t.externals; phases = std::max(phases, t.phases);
        Plan::Segment sg;
        sg.first = w4table.size();
        sg.W = t.W; sg.entries = t.entries; sg.nslots = t.nslots; sg.tail = t.tail;
        plan.segments.push_back(sg);
        w4table.resize(sg.first + t.prog.size());
        // bytes per buffer inside a block / tile, bytes per LDS slot
        const uint32_t slotb = wg ? wg_block_bytes(S) : 1024u;
        // a partials buffer inside a tile (20/61-state walk: bytes) / in the buffer-major 4-state arena (KiB: P_pad/64 x K of them)
        const uint32_t pbuf = wg ? (uint32_t) K * slotb : (uint32_t) ((size_t) (Ppad / 64) * K);
        const uint32_t ebuf = (uint32_t) K * 64u, mbuf = wg ? (uint32_t) (matrixFloats * 4) : (uint32_t) K * 64u;
        int prevKept = -1;                           // (20/61-state walk) slot the previous operation of the same program kept its result in
        for (size_t i = 0; i < t.prog.size(); ++i) {
            const Walk4Template::Entry& te = t.prog[i];
            Walk4Entry& e = w4table[sg.first + i];
            std::memset(&e, 0, sizeof e);
            if (i % (size_t) t.entries == 0) prevKept = -1;
            uint32_t flags = te.flags, mode = SCALE_NONE, keep = 0;
            e.ewrite = (uint32_t) scratchScale * ebuf;
            e.eread = (uint32_t) scratchScale * ebuf;
            e.ewrite = (uint32_t) (scratchScale + (int) (i % (size_t) (wg ? MBAMD_WG_SCRATCH_ROWS : MBAMD_W4_SCRATCH_ROWS))) * ebuf;   // (neighbouring entries: different scratch rows)
            if (wg) {
                // k_walkg (mbamd_walkg.h): no prefetch entries; a child that is neither a tip nor in a slot is read from
                // HBM by the operand pipeline; NOP entries store zeros to the extra buffer of the tile
                e.dst = (uint32_t) nBuffers * pbuf;
                if (te.op >= 0) {
                    const Walk4Op& op = seg[te.op];
                    e.dst = (uint32_t) op.dst * pbuf;
                    if (op.tip1) { e.c1 = (uint32_t) op.c1 * (uint32_t) MBAMD_WG_TW; flags |= MBAMD_W4_TIP1; }
                    else if (te.c1slot == 0xFF) { e.c1 = (uint32_t) op.c1 * pbuf; flags |= MBAMD_WG_MEM1; }
                    else e.c1 = (uint32_t) te.c1slot * slotb;
                    if (op.tip2) { e.c2 = (uint32_t) op.c2 * (uint32_t) MBAMD_WG_TW; flags |= MBAMD_W4_TIP2; }
                    else if (te.c2slot == 0xFF) { e.c2 = (uint32_t) op.c2 * pbuf; flags |= MBAMD_WG_MEM2; }
                    else e.c2 = (uint32_t) te.c2slot * slotb;
                    e.m1 = (uint32_t) op.m1 * mbuf;
                    e.m2 = (uint32_t) op.m2 * mbuf;
                    if (te.dslot != 0xFF) { keep = te.dslot; flags |= MBAMD_W4_KEEP; }
                    if (!op.tip1 && te.c1slot != 0xFF && (int) te.c1slot == prevKept) flags |= MBAMD_WG_PREV1;
                    if (!op.tip2 && te.c2slot != 0xFF && (int) te.c2slot == prevKept) flags |= MBAMD_WG_PREV2;
                    prevKept = te.dslot != 0xFF ? (int) te.dslot : -1;
                    mode = op.scaleWrite >= 0 ? SCALE_WRITE : (op.scaleRead >= 0 ? SCALE_READ : SCALE_NONE);
                    if (op.scaleWrite >= 0) e.ewrite = (uint32_t) op.scaleWrite * ebuf;
                    if (op.scaleRead >= 0) e.eread = (uint32_t) op.scaleRead * ebuf;
                    e.ctl = flags | (mode << 8) | ((uint32_t) segList[te.op] << 10) | (keep << 16);
                } else {
                    e.ctl = (flags & (MBAMD_W4_NOP | MBAMD_W4_BARRIER)) | MBAMD_W4_NOP;
                }
                continue;
            }
            if (te.pfOp[0] >= 0) {                          // PF entry
                const Walk4Op& p0 = seg[te.pfOp[0]];
                e.dst = (uint32_t) (te.pfChild[0] == 0 ? p0.c1 : p0.c2) * pbuf;
                e.c1 = (uint32_t) te.pfSlot[0] * 1024u;
                flags |= MBAMD_W4_PF0 | MBAMD_W4_NOP;
                if (te.pfOp[1] >= 0) {
                    const Walk4Op& p1 = seg[te.pfOp[1]];
                    e.c2 = (uint32_t) (te.pfChild[1] == 0 ? p1.c1 : p1.c2) * pbuf;
                    e.m1 = (uint32_t) te.pfSlot[1] * 1024u;
                    flags |= MBAMD_W4_PF1;
                }
            } else if (te.op >= 0) {
                const Walk4Op& op = seg[te.op];
                e.dst = (uint32_t) op.dst * pbuf;
                if (op.tip1) { e.c1 = (uint32_t) op.c1 * 32u; flags |= MBAMD_W4_TIP1; }
                else if (te.c1slot == 0xFE) flags |= MBAMD_W4_FWD1;
                else e.c1 = (uint32_t) te.c1slot * 1024u;
                if (op.tip2) { e.c2 = (uint32_t) op.c2 * 32u; flags |= MBAMD_W4_TIP2; }
                else if (te.c2slot == 0xFE) flags |= MBAMD_W4_FWD2;
                else e.c2 = (uint32_t) te.c2slot * 1024u;
                e.m1 = (uint32_t) op.m1 * mbuf;
                e.m2 = (uint32_t) op.m2 * mbuf;
                if (te.dslot != 0xFF) { keep = te.dslot; flags |= MBAMD_W4_KEEP; }
                mode = op.scaleWrite >= 0 ? SCALE_WRITE : (op.scaleRead >= 0 ? SCALE_READ : SCALE_NONE);
                if (op.scaleWrite >= 0) e.ewrite = (uint32_t) op.scaleWrite * ebuf;
                if (op.scaleRead >= 0) e.eread = (uint32_t) op.scaleRead * ebuf;
            }
            if (te.vmwait != 0xFF) flags |= MBAMD_W4_VMWAIT;
            // the entry in front of a SCALE_READ entry of the same wave fetches that entry's stored exponents (mbamd_walk4.h)
            if ((i + 1) % (size_t) t.entries != 0) {
                const Walk4Template::Entry& tn = t.prog[i + 1];
                if (tn.op >= 0 && seg[tn.op].scaleWrite < 0 && seg[tn.op].scaleRead >= 0) flags |= MBAMD_W4_NEXT_READS;
            }
            e.ctl = flags | (mode << 8) | ((uint32_t) (te.vmwait == 0xFF ? 0 : te.vmwait) << 10) | (keep << 16);
        }
        lastWalkW = t.W; lastWalkSlots = t.nslots; lastWalkEntries = t.entries; lastWalkPhases = t.phases;
        seg.clear();
        segList.clear();
        std::fill(segRead.begin(), segRead.end(), 0);
        std::fill(segWritten.begin(), segWritten.end(), 0);
        std::fill(segScale.begin(), segScale.end(), 0);
        return BEAGLE_SUCCESS;
    };
    for (int o = 0; o < n; ++o) {
        const BeagleOperation& b = ops[o];
        if (b.destinationPartials < 0 || b.destinationPartials >= nBuffers || b.child1Partials < 0 ||
            b.child1Partials >= nBuffers || b.child2Partials < 0 || b.child2Partials >= nBuffers)
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: partials index");
        if (b.child1TransitionMatrix < 0 || b.child1TransitionMatrix >= nMatrices || b.child2TransitionMatrix < 0 ||
            b.child2TransitionMatrix >= nMatrices)
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: matrix index");
        Walk4Op w;
        w.dst = b.destinationPartials;
        w.c1 = b.child1Partials; w.c2 = b.child2Partials;
        w.m1 = b.child1TransitionMatrix; w.m2 = b.child2TransitionMatrix;
        const int ci[2] = {w.c1, w.c2};
        uint8_t tip[2];
        for (int c = 0; c < 2; ++c) {
            tip[c] = (tipStates[ci[c]] && !written[ci[c]]) ? 1 : 0;
            if (!tip[c] && !valid[ci[c]] && !written[ci[c]])
                return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: child buffer was never written");
        }
        w.tip1 = tip[0]; w.tip2 = tip[1];
        w.scaleWrite = w.scaleRead = -1;
        if (b.destinationScaleWrite != BEAGLE_OP_NONE) {
            if (b.destinationScaleWrite < 0 || b.destinationScaleWrite >= nScale)
                return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: scale write index");
            w.scaleWrite = b.destinationScaleWrite;
        } else if (b.destinationScaleRead != BEAGLE_OP_NONE) {
            if (b.destinationScaleRead < 0 || b.destinationScaleRead >= nScale)
                return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: scale read index");
            if (scaleState[b.destinationScaleRead] == 2 && !segScale[b.destinationScaleRead])
                return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleUpdatePartials: destinationScaleRead names a cumulative buffer");
            w.scaleRead = b.destinationScaleRead;
        }
        // hazards that the in-launch dependency analysis does not cover end the segment (MrBayes never produces them):
        // a buffer written twice or written after it was read, an exponent buffer touched twice unless only read
        bool hazard = segWritten[w.dst] || segRead[w.dst];
        if (perList && o > 0 && listOf[o] != listOf[o - 1]) hazard = true;       // (independent lists: one program set each)
        if (w.scaleWrite >= 0 && segScale[w.scaleWrite]) hazard = true;
        if (w.scaleRead >= 0 && segScale[w.scaleRead] == 2) hazard = true;
        if (hazard) {
            int rc = flushSegment();
            if (rc) return rc;
        }
        segWritten[w.dst] = 1;
        if (!tip[0]) segRead[w.c1] = 1;
        if (!tip[1]) segRead[w.c2] = 1;
        if (w.scaleWrite >= 0) segScale[w.scaleWrite] = 2;
        if (w.scaleRead >= 0 && !segScale[w.scaleRead]) segScale[w.scaleRead] = 1;
        written[w.dst] = 1;
        seg.push_back(w);
        segList.push_back(listOf ? listOf[o] : 0);
    }
    int rc = flushSegment();
    if (rc) return rc;
    if (envVerbose)
        std::fprintf(stderr, "[mbamd] walk plan: %d ops, %zu segment(s), W=%d, %d entries/wave, %d slots/wave, %d phases, %d reloads, %d external children\n",
                     n, plan.segments.size(), lastWalkW, lastWalkEntries, lastWalkSlots, phases, reloads, externals);
    // a short program goes out with the launch itself (k_walk4_t<Walk4ArgsInline>, k_walkg<..., WalkGArgsInline>)
    plan.inlineProg.clear();
    if (!noInlinePrograms && plan.segments.size() == 1 && w4table.size() <= (size_t) MBAMD_W4_INLINE) {
        plan.inlineProg = w4table;
        return BEAGLE_SUCCESS;
    }
    // upload the programs into the plan's device buffer
    const size_t bytes = w4table.size() * sizeof(Walk4Entry);
    const bool inFlight = plan.lastLaunch > syncedClock;
    if (bytes > plan.cap || inFlight) {
        if (inFlight) { HIP_TRY(hipStreamSynchronize(stream)); syncedClock = launchClock; }
        if (bytes > plan.cap) {
            if (plan.d_table) HIP_TRY(hipFree(plan.d_table));
            plan.d_table = nullptr;
            plan.cap = 0;
            HIP_TRY(hipMalloc(&plan.d_table, bytes + bytes / 2));
            plan.cap = bytes + bytes / 2;
        }
    }
    return upload(plan.d_table, w4table.data(), bytes);
}

// host data -> the pinned ring -> a device buffer, by a launch of ours on the instance's stream
int Instance::ringCopy(void* dst, const void* src, size_t bytes)
{
    const void* ring = nullptr;
    int rc = stageDirect(src, bytes, &ring);
    if (rc) return rc;
    if (bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        const unsigned n16 = (unsigned) (bytes / 16);
        MBAMD_LAUNCH(k_copy_from_ring, (n16 + 255u) / 256u, 256, 0, stream, static_cast<const copy16_t*>(ring), static_cast<copy16_t*>(dst), n16);
    } else {
        const unsigned n4 = (unsigned) (bytes / 4);
        MBAMD_LAUNCH(k_copy_from_ring4, (n4 + 255u) / 256u, 256, 0, stream, static_cast<const unsigned*>(ring), static_cast<unsigned*>(dst), n4);
    }
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

// A root-ward path (the list of a move that dirtied one branch) as k_path4's entries: operation i has the result of operation i - 1
// as one child; its other child -- and both children of operation 0 -- are compact tips or buffers the list does not write; no buffer
// or exponent buffer is written twice or read after it is written.  Round 6: also FORKED paths (the list of a topology move: root-ward
// paths that join, in post-order) -- an operation that does not read its predecessor's result begins a new ARM (the predecessor's
// result is saved), an operation whose other child is the saved result JOINS the arms; one saved result at a time.  Anything else
// (false) is compiled by buildWalk.
bool Instance::buildPath4(Plan& plan, const BeagleOperation* ops, int n)
{
    if (noPath4 || n < 1 || n > MBAMD_W4_INLINE) return false;
    const int scratchScale = (int) scale.size();
    const uint32_t pbuf = (uint32_t) ((size_t) (Ppad / 64) * K), ebuf = (uint32_t) K * 64u, mbuf = (uint32_t) K * 64u;
    std::vector<Walk4Entry>& prog = plan.inlineProg;
    prog.assign((size_t) n, Walk4Entry());
    auto inList = [&](int buf, int upto) { for (int q = 0; q < upto; ++q) if (ops[q].destinationPartials == buf) return true; return false; };
    int saved = -1;                              // buffer of the saved result (an arm that waits for its join), or -1
    int armStart = 0, arms = 0;
    for (int i = 0; i < n; ++i) {
        const BeagleOperation& b = ops[i];
        if (b.destinationPartials < 0 || b.destinationPartials >= nBuffers || b.child1Partials < 0 || b.child1Partials >= nBuffers ||
            b.child2Partials < 0 || b.child2Partials >= nBuffers || b.child1TransitionMatrix < 0 || b.child1TransitionMatrix >= nMatrices ||
            b.child2TransitionMatrix < 0 || b.child2TransitionMatrix >= nMatrices) return false;      // (buildWalk reports it)
        if (tipStates[b.destinationPartials] || inList(b.destinationPartials, i)) return false;
        int chain, sib, mchain, msib;
        const int prev = i == 0 ? -1 : ops[i - 1].destinationPartials;
        const bool one = i > 0 && b.child1Partials == prev, two = i > 0 && b.child2Partials == prev;
        if (one && two) return false;
        const bool start = !one && !two;
        bool join = false;
        if (start) {
            if (i > 0) {
                if (noForkPath || saved >= 0) return false;                // (two results waiting: not this kernel's shape)
                saved = prev;
                prog[(size_t) armStart].ctl |= (uint32_t) (i - armStart) << 16;
                armStart = i;
            }
            ++arms;
            chain = b.child1Partials; sib = b.child2Partials; mchain = b.child1TransitionMatrix; msib = b.child2TransitionMatrix;
        } else {
            chain = one ? b.child1Partials : b.child2Partials; sib = one ? b.child2Partials : b.child1Partials;
            mchain = one ? b.child1TransitionMatrix : b.child2TransitionMatrix; msib = one ? b.child2TransitionMatrix : b.child1TransitionMatrix;
            if (saved >= 0 && sib == saved) { join = true; saved = -1; }
        }
        // what comes from outside must not be written anywhere in the list (before: a second dependency; after: a hazard)
        for (int ext : {join ? -1 : sib, start ? chain : -1})
            if (ext >= 0) {
                if (inList(ext, n)) return false;
                if (!tipStates[ext] && !valid[ext]) return false;
            }
        Walk4Entry& e = prog[(size_t) i];
        std::memset(&e, 0, sizeof e);
        uint32_t flags = 0, mode = SCALE_NONE;
        e.dst = (uint32_t) b.destinationPartials * pbuf;
        if (start) {
            flags |= MBAMD_P4_START;
            if (tipStates[chain]) { e.c1 = (uint32_t) chain * 32u; flags |= MBAMD_W4_TIP1; }
            else e.c1 = (uint32_t) chain * pbuf;
        }
        if (join) flags |= MBAMD_P4_JOIN;
        else if (tipStates[sib]) { e.c2 = (uint32_t) sib * 32u; flags |= MBAMD_W4_TIP2; }
        else e.c2 = (uint32_t) sib * pbuf;
        e.m1 = (uint32_t) mchain * mbuf;
        e.m2 = (uint32_t) msib * mbuf;
        e.ewrite = e.eread = (uint32_t) scratchScale * ebuf;
        if (b.destinationScaleWrite != BEAGLE_OP_NONE) {
            if (b.destinationScaleWrite < 0 || b.destinationScaleWrite >= nScale) return false;
            for (int q = 0; q < n; ++q)
                if (q != i && (ops[q].destinationScaleWrite == b.destinationScaleWrite || ops[q].destinationScaleRead == b.destinationScaleWrite)) return false;
            mode = SCALE_WRITE;
            e.ewrite = (uint32_t) b.destinationScaleWrite * ebuf;
        } else if (b.destinationScaleRead != BEAGLE_OP_NONE) {
            if (b.destinationScaleRead < 0 || b.destinationScaleRead >= nScale || scaleState[b.destinationScaleRead] == 2) return false;
            mode = SCALE_READ;
            e.eread = (uint32_t) b.destinationScaleRead * ebuf;
        }
        e.ctl = flags | (mode << 8);
    }
    if (saved >= 0) return false;                // (an arm nobody joins: two trees in one list)
    prog[(size_t) armStart].ctl |= (uint32_t) (n - armStart) << 16;
    plan.forked = arms > 1;
    plan.path = true;
    plan.segments.clear();
    Plan::Segment sg;
    sg.first = 0; sg.W = 1; sg.entries = n; sg.nslots = 0; sg.tail = 0;
    plan.segments.push_back(sg);
    lastWalkW = 1; lastWalkSlots = 0; lastWalkEntries = n; lastWalkPhases = 1;
    return true;
}

// The same for the 20/61-state walk (k_pathg, mbamd_pathg_kernel.h): `nl` mutually independent lists (the eigen-system parts of a
// codon model; one for a protein model), each a root-ward path, all of the same length.  Entries [list][operation] in the tile
// arena's units (byte offsets of a buffer inside a tile, tip states at 32 bytes per buffer, matrix buffers in bytes).
static inline bool pathg_compiled(int S) { return S == 20 || (S >= 60 && S <= 63); }
bool Instance::buildPathG(Plan& plan, const BeagleOperation* ops, int n, const std::vector<int>& starts, int nl)
{
    if (noPathG || !pathg_compiled(S) || nl < 1 || nl > MBAMD_WG_MAXLISTS || n < nl || n % nl != 0) return false;
    const int L = n / nl;
    if (L < 2 || (size_t) n > (size_t) MBAMD_W4_INLINE) return false;      // (a single operation gains nothing; the program travels in the kernel arguments)
    for (int q = 0; q < nl; ++q) if (starts[(size_t) q] != q * L) return false;
    const int scratchScale = (int) scale.size();
    const uint32_t slotb = wg_block_bytes(S), pbuf = (uint32_t) K * slotb, ebuf = (uint32_t) K * 64u, mbuf = (uint32_t) (matrixFloats * 4);
    auto writtenIn = [&](int buf, int lo, int hi) { for (int o = lo; o < hi; ++o) if (ops[o].destinationPartials == buf) return true; return false; };
    std::vector<Walk4Entry>& prog = plan.inlineProg;
    prog.assign((size_t) n, Walk4Entry());
    for (int q = 0; q < nl; ++q) {
        const int lo = q * L;
        for (int i = 0; i < L; ++i) {
            const BeagleOperation& b = ops[lo + i];
            if (b.destinationPartials < 0 || b.destinationPartials >= nBuffers || b.child1Partials < 0 || b.child1Partials >= nBuffers ||
                b.child2Partials < 0 || b.child2Partials >= nBuffers || b.child1TransitionMatrix < 0 || b.child1TransitionMatrix >= nMatrices ||
                b.child2TransitionMatrix < 0 || b.child2TransitionMatrix >= nMatrices) return false;      // (buildWalk reports it)
            if (tipStates[b.destinationPartials] || writtenIn(b.destinationPartials, 0, lo + i)) return false;
            int chain, sib, mchain, msib;
            if (i == 0) { chain = b.child1Partials; sib = b.child2Partials; mchain = b.child1TransitionMatrix; msib = b.child2TransitionMatrix; }
            else {
                const int prev = ops[lo + i - 1].destinationPartials;
                const bool one = b.child1Partials == prev, two = b.child2Partials == prev;
                if (one == two) return false;
                chain = one ? b.child1Partials : b.child2Partials; sib = one ? b.child2Partials : b.child1Partials;
                mchain = one ? b.child1TransitionMatrix : b.child2TransitionMatrix; msib = one ? b.child2TransitionMatrix : b.child1TransitionMatrix;
            }
            for (int ext : {sib, i == 0 ? chain : -1})
                if (ext >= 0) {
                    if (writtenIn(ext, 0, n)) return false;                // (nothing any of the lists writes)
                    if (!tipStates[ext] && !valid[ext]) return false;
                }
            Walk4Entry& e = prog[(size_t) (lo + i)];
            std::memset(&e, 0, sizeof e);
            uint32_t flags = 0, mode = SCALE_NONE;
            e.dst = (uint32_t) b.destinationPartials * pbuf;
            if (i == 0) {
                if (tipStates[chain]) { e.c1 = (uint32_t) chain * (uint32_t) MBAMD_WG_TW; flags |= MBAMD_W4_TIP1; }
                else e.c1 = (uint32_t) chain * pbuf;
            }
            if (tipStates[sib]) { e.c2 = (uint32_t) sib * (uint32_t) MBAMD_WG_TW; flags |= MBAMD_W4_TIP2; }
            else e.c2 = (uint32_t) sib * pbuf;
            e.m1 = (uint32_t) mchain * mbuf;
            e.m2 = (uint32_t) msib * mbuf;
            e.ewrite = e.eread = (uint32_t) scratchScale * ebuf;
            e.ewrite = (uint32_t) (scratchScale + i % MBAMD_WG_SCRATCH_ROWS) * ebuf;      // (a different scratch row for neighbouring entries, see the arena)
            if (b.destinationScaleWrite != BEAGLE_OP_NONE) {
                if (b.destinationScaleWrite < 0 || b.destinationScaleWrite >= nScale) return false;
                for (int o = 0; o < n; ++o)
                    if (o != lo + i && (ops[o].destinationScaleWrite == b.destinationScaleWrite || ops[o].destinationScaleRead == b.destinationScaleWrite)) return false;
                mode = SCALE_WRITE;
                e.ewrite = (uint32_t) b.destinationScaleWrite * ebuf;
            } else if (b.destinationScaleRead != BEAGLE_OP_NONE) {
                if (b.destinationScaleRead < 0 || b.destinationScaleRead >= nScale || scaleState[b.destinationScaleRead] == 2) return false;
                mode = SCALE_READ;
                e.eread = (uint32_t) b.destinationScaleRead * ebuf;
            }
            e.ctl = flags | (mode << 8) | ((uint32_t) q << 10);
        }
    }
    if (envVerbose) std::fprintf(stderr, "[mbamd] walk plan: %d list(s) of %d operations each: root-ward paths (k_pathg)\n", nl, L);
    plan.pathG = true;
    plan.lists = nl;
    plan.segments.clear();
    Plan::Segment sg;
    sg.first = 0; sg.W = 1; sg.entries = L; sg.nslots = 0; sg.tail = 0;
    plan.segments.push_back(sg);
    lastWalkW = 1; lastWalkSlots = 0; lastWalkEntries = L; lastWalkPhases = 1;
    return true;
}

int Instance::runWalk(const Plan& plan, int32_t* cum)
{
    if (plan.path) {
        Walk4ArgsInline ai;
        Walk4Args& a = ai.a;
        a.prog = nullptr;
        a.entries = (int) plan.inlineProg.size();
        a.nslots = 0;
        a.partials = reinterpret_cast<f4*>(arenaPartials);
        a.pstride = geom.pstride;
        a.tips = arenaTips;
        a.tstride = geom.tstride;
        a.exps = arenaExp;
        a.estride = estride;
        a.matrices = matrices;
        a.cum = cum;
        a.cumFresh = walkCumFresh ? 1 : 0;
        a.K = K;
        a.Ppad = Ppad;
        a.nblocks = Ppad / 64;
        a.tail = 0;
        std::memcpy(ai.inl, plan.inlineProg.data(), plan.inlineProg.size() * sizeof(Walk4Entry));
        auto kernel = k_path4<Walk4ArgsInline>;
        MBAMD_LAUNCH_BARRIER(kernel, walk4_grid(Ppad / 64, K), 64, path4_lds_bytes((int) plan.inlineProg.size()), stream, ai);    // (lanes exchange through LDS: the emulation runs them as fibers)
        HIP_TRY(hipGetLastError());
        pendingLaunches += 1;
        return BEAGLE_SUCCESS;
    }
    for (const Plan::Segment& sg : plan.segments) {
        Walk4Args a;
        a.prog = reinterpret_cast<const Walk4Entry*>(plan.d_table) + sg.first;
        a.entries = sg.entries;
        a.nslots = sg.nslots;
        a.partials = reinterpret_cast<f4*>(arenaPartials);
        a.pstride = geom.pstride;
        a.tips = arenaTips;
        a.tstride = geom.tstride;
        a.exps = arenaExp;
        a.estride = estride;
        a.matrices = matrices;
        a.cum = cum;
        a.cumFresh = (walkCumFresh && &sg == &plan.segments.front()) ? 1 : 0;
        a.K = K;
        a.Ppad = Ppad;
        a.nblocks = Ppad / 64;
        a.tail = sg.tail;
        if (!plan.inlineProg.empty()) {
            Walk4ArgsInline ai;
            ai.a = a;
            ai.a.prog = nullptr;
            std::memcpy(ai.inl, plan.inlineProg.data(), plan.inlineProg.size() * sizeof(Walk4Entry));
            auto kernel = k_walk4_t<Walk4ArgsInline>;
            MBAMD_LAUNCH_BARRIER(kernel, walk4_grid(Ppad / 64, K), 64 * sg.W, walk4_lds_bytes(sg.W, sg.nslots), stream, ai);
        } else {
            auto kernel = k_walk4_t<Walk4Args>;
            MBAMD_LAUNCH_BARRIER(kernel, walk4_grid(Ppad / 64, K), 64 * sg.W, walk4_lds_bytes(sg.W, sg.nslots), stream, a);
        }
        HIP_TRY(hipGetLastError());
        pendingLaunches += 1;
    }
    return BEAGLE_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// 20/61-state tree walk (mbamd_walkg.h).  beagleUpdatePartials only queues: MrBayes submits one list per eigen-system
// part of a codon model (reference src/mbbeagle.c:1095-1104), and a forest of three trees fills the chip where one tree
// cannot.  The queue runs -- as ONE program per wave, hazards cut into segments like any list -- when the next call
// arrives that depends on it.
// ---------------------------------------------------------------------------------------------
int Instance::updatePartialsG(const BeagleOperation* ops, int n, int cumIdx)
{
    bool clash = (int) wgListCum.size() >= MBAMD_WG_MAXLISTS;
    if (cumIdx != BEAGLE_OP_NONE)
        for (int c : wgListCum) clash |= c == cumIdx;            // one list per cumulative buffer and launch
    if (clash) {
        int rc = flushPending();
        if (rc) return rc;
    }
    wgListStart.push_back((int) wgOps.size());
    wgListCum.push_back(cumIdx);
    wgOps.insert(wgOps.end(), ops, ops + n);
    if (noDefer) return flushPending();
    return BEAGLE_SUCCESS;
}

// does a beagle{Accumulate,Remove}ScaleFactors call commute with the queued lists?  (MrBayes removes the old node factors
// of part j+1 between the lists of parts j and j+1, reference src/mbbeagle.c:1086-1104)
bool Instance::scaleOpsIndependentOfPending(const int* idx, int n, int cumIdx) const
{
    for (int c : wgListCum) if (c == cumIdx) return false;
    for (const BeagleOperation& b : wgOps) {
        if (b.destinationScaleWrite == cumIdx || b.destinationScaleRead == cumIdx) return false;
        for (int i = 0; i < n; ++i)
            if (b.destinationScaleWrite == idx[i]) return false;
    }
    for (int i = 0; i < n; ++i)
        for (int c : wgListCum) if (c == idx[i]) return false;
    return true;
}

int Instance::flushWalkG()
{
    if (wgListCum.empty()) return BEAGLE_SUCCESS;
    std::vector<BeagleOperation> ops;
    std::vector<int> starts, cums;
    ops.swap(wgOps); starts.swap(wgListStart); cums.swap(wgListCum);
    const int n = (int) ops.size();
    int nl = (int) cums.size();
    std::vector<int> listOf(n, 0);
    for (int q = 0; q < nl; ++q)
        for (int o = starts[q]; o < (q + 1 < nl ? starts[q + 1] : n); ++o) listOf[o] = q;
    // ---- plan cache key: the operations as submitted, the list boundaries, the layout epoch ------------------------------
    std::vector<int> key(reinterpret_cast<const int*>(ops.data()), reinterpret_cast<const int*>(ops.data()) + (size_t) n * 7);
    for (int q = 0; q < nl; ++q) key.push_back(starts[q]);
    key.push_back(layoutEpoch);
    // One list that does not rescale may hold several independent trees: without a cumulative buffer MrBayes submits the
    // operations of all eigen-system parts as ONE list (reference src/mbbeagle.c:1029-1104, No_Rescale), and a move dirties
    // the same root-ward path in each part.  Its connected components are treated like lists of their own.
    if (nl == 1 && cums[0] == BEAGLE_OP_NONE && n >= 2) {
        std::vector<int> comp(n);
        for (int o = 0; o < n; ++o) comp[o] = o;
        auto find = [&](int x) { while (comp[x] != x) x = comp[x] = comp[comp[x]]; return x; };
        std::unordered_map<int, int> writer, scaleUser;
        bool ok = true;
        for (int o = 0; o < n && ok; ++o) {
            const BeagleOperation& b = ops[o];
            for (int c : {b.child1Partials, b.child2Partials}) {
                auto it = writer.find(c);
                if (it != writer.end()) comp[find(o)] = find(it->second);
            }
            if (writer.count(b.destinationPartials)) ok = false;            // (written twice: leave it to the hazard segments)
            writer[b.destinationPartials] = o;
            for (int sc : {b.destinationScaleWrite, b.destinationScaleRead})
                if (sc != BEAGLE_OP_NONE) {
                    auto it = scaleUser.find(sc);
                    if (it != scaleUser.end()) comp[find(o)] = find(it->second); else scaleUser[sc] = o;
                }
        }
        for (int o = 0; o < n && ok; ++o)                                    // a buffer read before a later operation writes it
            for (int c : {ops[o].child1Partials, ops[o].child2Partials}) {
                auto it = writer.find(c);
                if (it != writer.end() && it->second > o) ok = false;
            }
        std::vector<int> roots;
        for (int o = 0; o < n && ok; ++o) if (find(o) == o) roots.push_back(o);
        if (ok && roots.size() >= 2 && roots.size() <= (size_t) MBAMD_WG_MAXLISTS) {
            std::vector<BeagleOperation> sorted;
            sorted.reserve(n);
            starts.clear();
            for (size_t q = 0; q < roots.size(); ++q) {
                starts.push_back((int) sorted.size());
                for (int o = 0; o < n; ++o) if (find(o) == roots[q]) sorted.push_back(ops[o]);
            }
            ops.swap(sorted);
            nl = (int) roots.size();
            cums.assign(nl, BEAGLE_OP_NONE);
            for (int q = 0; q < nl; ++q)
                for (int o = starts[q]; o < (q + 1 < nl ? starts[q + 1] : n); ++o) listOf[o] = q;
        }
    }
    wgFresh = 0;
    for (int q = 0; q < MBAMD_WG_MAXLISTS; ++q) wgCum[q] = nullptr;
    for (int q = 0; q < nl; ++q) {
        const int ci = cums[q];
        if (ci == BEAGLE_OP_NONE) continue;
        if (scaleState[ci] == 0) {
            // a freshly reset cumulative buffer (the rescale-everything pass): the kernel STORES its sums, no zero-fill launch
            if (!wideScale[ci]) HIP_TRY(hipMalloc(&wideScale[ci], (size_t) K * Ppad * sizeof(int32_t)));
            scaleState[ci] = 2;
            wgFresh |= 1 << q;
        } else {
            int rc = ensureWide(ci);
            if (rc) return rc;
        }
        wgCum[q] = wideScale[ci];
    }
    uint64_t h = 1469598103934665603ull;
    for (int v : key) h = (h ^ (uint64_t) (uint32_t) v) * 1099511628211ull;
    Plan* plan = nullptr;
    for (Plan* pl : plans)
        if (pl->hash == h && pl->key == key) {
            pl->lastUse = ++planClock;
            planHits++;
            plan = pl;
            break;
        }
    if (!plan) {
        planMisses++;
        const size_t maxPlans = 24;
        if (plans.size() < maxPlans) {
            plan = new Plan();
            plans.push_back(plan);
        } else {
            plan = plans[0];
            for (Plan* pl : plans) if (pl->lastUse < plan->lastUse) plan = pl;
        }
        plan->key = key;
        plan->hash = h;
        plan->lastUse = ++planClock;
        // Mutually independent lists (the eigen-system parts of a codon model) run as separate workgroups of ONE launch --
        // three times the workgroups for a grid that does not fill the chip otherwise -- if they compile to the same geometry;
        // anything else is one merged forest.
        bool independent = nl > 1;
        if (independent) {
            std::vector<int> wr(nBuffers, -1), rd(nBuffers, -1);
            std::vector<int> sc(scale.size(), -1);
            for (int o = 0; o < n && independent; ++o) {
                const BeagleOperation& b = ops[o];
                const int q = listOf[o];
                auto clash = [&](std::vector<int>& v, int i) { if (i < 0 || i >= (int) v.size()) return false; if (v[i] >= 0 && v[i] != q) return true; v[i] = q; return false; };
                if (clash(wr, b.destinationPartials) || (b.destinationPartials >= 0 && b.destinationPartials < nBuffers && rd[b.destinationPartials] >= 0 && rd[b.destinationPartials] != q)) independent = false;
                for (int c : {b.child1Partials, b.child2Partials})
                    if (c >= 0 && c < nBuffers && !(tipStates[c] && wr[c] < 0)) {
                        if (wr[c] >= 0 && wr[c] != q) independent = false;
                        if (rd[c] < 0) rd[c] = q; else if (rd[c] != q) rd[c] = 1 << 20;      // (read by several lists: fine unless one writes it)
                    }
                if (b.destinationScaleWrite != BEAGLE_OP_NONE && clash(sc, b.destinationScaleWrite)) independent = false;
                if (b.destinationScaleRead != BEAGLE_OP_NONE && b.destinationScaleRead >= 0 && b.destinationScaleRead < (int) sc.size() &&
                    sc[b.destinationScaleRead] >= 0 && sc[b.destinationScaleRead] != q) independent = false;
            }
            for (int o = 0; o < n && independent; ++o)                  // a buffer one list writes must not be read by another
                for (int c : {ops[o].child1Partials, ops[o].child2Partials})
                    if (c >= 0 && c < nBuffers && wr[c] >= 0 && wr[c] != listOf[o]) independent = false;
            if (envVerbose) std::fprintf(stderr, "[mbamd] %d queued lists, %d operations: %s\n", nl, n, independent ? "independent" : "one forest");
        }
        int rc;
        {
            StatTimer st_(ST_PLAN);
            plan->lists = 1;
            plan->pathG = false;
            rc = BEAGLE_SUCCESS;
            bool done = (nl == 1 || independent) && buildPathG(*plan, ops.data(), n, starts, nl);
            if (!done) plan->inlineProg.clear();
            if (!done && independent) {
                const int keepW = w4.maxW, keepS = w4.maxSlots, keepS1 = w4.maxSlots1;
                wgGeometry(nl, w4.maxW, w4.maxSlots);
                w4.maxSlots1 = w4.maxSlots;
                rc = buildWalk(*plan, ops.data(), n, listOf.data(), true);
                w4.maxW = keepW; w4.maxSlots = keepS; w4.maxSlots1 = keepS1;
                bool same = rc == BEAGLE_SUCCESS && (int) plan->segments.size() == nl;
                for (size_t i = 1; same && i < plan->segments.size(); ++i)
                    same = plan->segments[i].W == plan->segments[0].W && plan->segments[i].entries == plan->segments[0].entries &&
                           plan->segments[i].first == plan->segments[0].first + i * (size_t) plan->segments[0].W * plan->segments[0].entries;
                if (same) {
                    int ns = 0;
                    for (const Plan::Segment& sg : plan->segments) ns = std::max(ns, sg.nslots);
                    for (Plan::Segment& sg : plan->segments) sg.nslots = ns;
                    plan->lists = nl;
                    done = true;
                    // (several independent lists = several segments, one launch: short enough, they travel in its arguments too)
                    if (!noInlinePrograms && w4table.size() <= (size_t) MBAMD_W4_INLINE) plan->inlineProg = w4table;
                }
            }
            if (!done) rc = buildWalk(*plan, ops.data(), n, listOf.data(), false);
        }
        if (rc) { plan->hash = 0; plan->key.clear(); return rc; }
    }
    for (int o = 0; o < n; ++o) {
        valid[ops[o].destinationPartials] = 1;
        if (ops[o].destinationScaleWrite != BEAGLE_OP_NONE) scaleState[ops[o].destinationScaleWrite] = 1;
    }
    return timedRun(*plan, nullptr);
}

template <int SC_, int WMAX_, int CH_, int DEPTH_>
static void launch_walkg_t(Instance& in, const WalkGArgs& a, int W, int nslots, const std::vector<Walk4Entry>* inlineProg)
{
    if (inlineProg && !inlineProg->empty()) {
        WalkGArgsInline ai;
        ai.a = a;
        ai.a.prog = nullptr;
        std::memcpy(ai.inl, inlineProg->data(), inlineProg->size() * sizeof(Walk4Entry));
        auto kern = k_walkg<SC_, WMAX_, CH_, DEPTH_, WalkGArgsInline>;
        MBAMD_LAUNCH_BARRIER(kern, walkg_grid(in.Ppad / MBAMD_WG_TW, in.K * a.lists), 64 * W * (a.spread ? 2 : 1), wg_lds_bytes(W, nslots, in.S), in.stream, ai);
        return;
    }
    auto kern = k_walkg<SC_, WMAX_, CH_, DEPTH_>;
#if defined(MBAMD_WG_ABL_TAIL_FENCE)
    if (std::getenv("MBAMD_WG_TAIL_FENCE")) {     // (experiment: see the end of k_walkg)
        static long long* counters = nullptr;
        if (!counters) { (void) hipMalloc(&counters, (size_t) (in.Ppad / MBAMD_WG_TW) * 8); (void) hipMemset(counters, 0, (size_t) (in.Ppad / MBAMD_WG_TW) * 8); }
        WalkGArgs b = a;
        b.reserved = counters;
        MBAMD_LAUNCH_BARRIER(kern, walkg_grid(in.Ppad / MBAMD_WG_TW, in.K * a.lists), 64 * W * (a.spread ? 2 : 1), wg_lds_bytes(W, nslots, in.S), in.stream, b);
        return;
    }
#endif
    MBAMD_LAUNCH_BARRIER(kern, walkg_grid(in.Ppad / MBAMD_WG_TW, in.K * a.lists), 64 * W * (a.spread ? 2 : 1), wg_lds_bytes(W, nslots, in.S), in.stream, a);
}

template <int SC_>
static void launch_pathg_t(Instance& in, const WalkGArgs& a, const std::vector<Walk4Entry>& prog)
{
    WalkGArgsInline ai;
    ai.a = a;
    ai.a.prog = nullptr;
    std::memcpy(ai.inl, prog.data(), prog.size() * sizeof(Walk4Entry));
    auto kern = k_pathg<SC_, WalkGArgsInline>;
    MBAMD_LAUNCH_BARRIER(kern, walkg_grid(in.Ppad / MBAMD_WG_TW, in.K * a.lists), 128, pathg_lds_bytes(in.S), in.stream, ai);
}

int Instance::runWalkG(const Plan& plan)
{
    if (plan.pathG) {
        const Plan::Segment& sg = plan.segments.front();
        WalkGArgs a;
        std::memset(&a, 0, sizeof a);
        a.entries = sg.entries;
        a.partials = arenaPartials;
        a.tileBytes = wgTileBytes;
        a.tips = arenaTipStates;
        a.tipTileBytes = wgTipTileBytes;
        a.exps = arenaExp;
        a.estride = estride;
        a.matrices = matrices;
        a.tabOff = (unsigned) (wgTabFloats * 4);
        a.tabBytes = (unsigned) (wg_table_floats(S) * 4);
        for (int q = 0; q < MBAMD_WG_MAXLISTS; ++q) a.cum[q] = wgCum[q];
        a.cumFresh = wgFresh;
        a.K = K; a.Ppad = Ppad; a.ntiles = Ppad / MBAMD_WG_TW; a.S = S; a.SP = SP;
        a.lists = plan.lists;
        switch (S) {
            case 20: launch_pathg_t<20>(*this, a, plan.inlineProg); break;
            case 60: launch_pathg_t<60>(*this, a, plan.inlineProg); break;
            case 61: launch_pathg_t<61>(*this, a, plan.inlineProg); break;
            case 62: launch_pathg_t<62>(*this, a, plan.inlineProg); break;
            default: launch_pathg_t<63>(*this, a, plan.inlineProg); break;
        }
        HIP_TRY(hipGetLastError());
        pendingLaunches += 1;
        return BEAGLE_SUCCESS;
    }
    for (const Plan::Segment& sg : plan.segments) {
        if (plan.lists > 1 && &sg != &plan.segments.front()) break;     // (independent lists: one launch covers all segments)
        WalkGArgs a;
        std::memset(&a, 0, sizeof a);
        a.prog = reinterpret_cast<const Walk4Entry*>(plan.d_table) + sg.first;
        a.entries = sg.entries;
        a.nslots = sg.nslots;
        a.partials = arenaPartials;
        a.tileBytes = wgTileBytes;
        a.tips = arenaTipStates;
        a.tipTileBytes = wgTipTileBytes;
        a.exps = arenaExp;
        a.estride = estride;
        a.matrices = matrices;
        a.tabOff = (unsigned) (wgTabFloats * 4);
        a.tabBytes = (unsigned) (wg_table_floats(S) * 4);
        for (int q = 0; q < MBAMD_WG_MAXLISTS; ++q) a.cum[q] = wgCum[q];
        a.cumFresh = (&sg == &plan.segments.front()) ? wgFresh : 0;
        a.K = K; a.Ppad = Ppad; a.ntiles = Ppad / MBAMD_WG_TW; a.S = S; a.SP = SP;
        a.lists = plan.lists;
        a.spread = sg.W == 2 ? 1 : 0;     // two-wave workgroups are launched as four (see k_walkg)
        MBAMD_WG_DISPATCH(S, launch_walkg_t, *this, a, sg.W, sg.nslots, &plan.inlineProg);
        HIP_TRY(hipGetLastError());
        pendingLaunches += 1;
    }
    return BEAGLE_SUCCESS;
}

template <int NT_, int SC_, int KC_>
static void launch_mfma_t(Instance& in, const PartialsOp* ops, int count, int32_t* cum)
{
    const int gx = (in.Ppad + 127) / 128;
    const unsigned grid = (unsigned) (8 * ((gx + 7) / 8) * count);
    auto kern = k_partials_mfma<NT_, SC_, KC_>;
    MBAMD_LAUNCH_BARRIER(kern, grid, 256, 0, in.stream, ops, in.S, in.SP, in.Ppad, gx, cum);
}
template <int NT_, int SC_, int KC_>
static void launch_mfma_split_t(Instance& in, const OpTables& tabs, int count)
{
    constexpr int NP = 2 * KC_ * NT_;
    const int gx = in.Ppad / 32;
    auto kern = k_partials_mfma_split<NT_, SC_, KC_>;
    MBAMD_LAUNCH_BARRIER(kern, (unsigned) (gx * count), 64 * NP, (size_t) NP * (8 * 64 + 32 + 16 * 32) * sizeof(float), in.stream, tabs, in.S,
                 in.SP, in.Ppad, gx);
}
// one launch over up to four operation tables (false: no split kernel for this shape)
static bool launch_mfma_split(Instance& in, const OpTables& tabs, int count)
{
    const int S = in.S, K = in.K;
    if (in.NT == 1 && S == 20 && K == 4) { launch_mfma_split_t<1, 20, 4>(in, tabs, count); return true; }
    if (in.NT == 1 && S == 20 && K == 1) { launch_mfma_split_t<1, 20, 1>(in, tabs, count); return true; }
    if (in.NT == 2 && S == 61 && K == 1) { launch_mfma_split_t<2, 61, 1>(in, tabs, count); return true; }
    if (in.NT == 1 && K == 1) { launch_mfma_split_t<1, 0, 1>(in, tabs, count); return true; }
    if (in.NT == 1 && K == 2) { launch_mfma_split_t<1, 0, 2>(in, tabs, count); return true; }
    if (in.NT == 1 && K == 4) { launch_mfma_split_t<1, 0, 4>(in, tabs, count); return true; }
    if (in.NT == 2 && K == 1) { launch_mfma_split_t<2, 0, 1>(in, tabs, count); return true; }
    if (in.NT == 2 && K == 2) { launch_mfma_split_t<2, 0, 2>(in, tabs, count); return true; }
    return false;
}
template <int SC_, int KC_>
static void launch_tips_t(Instance& in, const OpTables& tabs, int count)
{
    const int gx4 = (in.Ppad + 127) / 128;
    auto kern = k_partials_tips<SC_, KC_>;
    MBAMD_LAUNCH_BARRIER(kern, (unsigned) (gx4 * count), 256, (size_t) 4 * in.S * 32 * sizeof(float), in.stream, tabs, in.S, in.SP, in.Ppad, gx4);
}
// operations on two compact tips, up to four tables (false: no kernel for this shape)
static bool launch_tips(Instance& in, const OpTables& tabs, int count)
{
    const int S = in.S, K = in.K;
    if (S > 64) return false;
    if (S == 20 && K == 4) { launch_tips_t<20, 4>(in, tabs, count); return true; }
    if (S == 20 && K == 1) { launch_tips_t<20, 1>(in, tabs, count); return true; }
    if (S == 61 && K == 1) { launch_tips_t<61, 1>(in, tabs, count); return true; }
    if (K == 1) { launch_tips_t<0, 1>(in, tabs, count); return true; }
    if (K == 2 && S <= 32) { launch_tips_t<0, 2>(in, tabs, count); return true; }
    return false;
}
template <int NT_, int SC_, int KC_>
static void launch_mfma_serial_t(Instance& in, const OpTables& tabs, int ntables)
{
    constexpr int NP = 2 * KC_ * NT_;
    const int gx = in.Ppad / 32;
    auto kern = k_partials_mfma_serial<NT_, SC_, KC_>;
    if (!in.d_trace && in.envTrace) {
        if (hipMalloc(&in.d_trace, (size_t) 4096 * 8 * 3 * sizeof(long long)) != hipSuccess) in.d_trace = nullptr;
        else (void) hipMemset(in.d_trace, 0, (size_t) 4096 * 8 * 3 * sizeof(long long));
    }
    MBAMD_LAUNCH_BARRIER(kern, (unsigned) (gx * ntables), 64 * NP, (size_t) NP * (8 * 64 + 32 + 16 * 32) * sizeof(float), in.stream, tabs, in.S,
                 in.SP, in.Ppad, gx, in.d_trace);
    if (in.d_trace) { in.lastWalkSteps = tabs.start[0]; in.walkWaves = NP - 1; }
}
template <int NT_, int SC_, int KC_>
static void launch_mfma_spine_t(Instance& in, const OpTables& tabs, int ntables)
{
    constexpr int NP = 2 * KC_ * NT_;
    const int gx = in.Ppad / 32;
    if (!in.d_trace && in.envTrace) {
        if (hipMalloc(&in.d_trace, (size_t) 4096 * 8 * 3 * sizeof(long long)) != hipSuccess) in.d_trace = nullptr;
        else (void) hipMemset(in.d_trace, 0, (size_t) 4096 * 8 * 3 * sizeof(long long));
    }
    auto kern = k_partials_mfma_spine<NT_, SC_, KC_>;
    MBAMD_LAUNCH_BARRIER(kern, (unsigned) (gx * ntables), 64 * (NP + 1), ((size_t) NP * (8 * 64 + 32) + (size_t) 2 * KC_ * SC_ * 32) * sizeof(float),
                 in.stream, tabs, in.SP, gx, in.d_trace);
    if (in.d_trace) { in.lastWalkSteps = tabs.start[0]; in.walkWaves = NP - 1; }
}
// one launch that walks up to four whole (narrow) operation lists; tabs.start[t] = operations of list t
static bool launch_mfma_serial(Instance& in, const OpTables& tabs, int ntables)
{
    const int S = in.S, K = in.K;
    if (!in.noSpine) {                           // software-pipelined variant (MBAMD_NO_SPINE=1: plain serial kernel)
        if (in.NT == 1 && S == 20 && K == 4) { launch_mfma_spine_t<1, 20, 4>(in, tabs, ntables); return true; }
        if (in.NT == 1 && S == 20 && K == 1) { launch_mfma_spine_t<1, 20, 1>(in, tabs, ntables); return true; }
        if (in.NT == 2 && S == 61 && K == 1) { launch_mfma_spine_t<2, 61, 1>(in, tabs, ntables); return true; }
    }
    if (in.NT == 1 && S == 20 && K == 4) { launch_mfma_serial_t<1, 20, 4>(in, tabs, ntables); return true; }
    if (in.NT == 1 && S == 20 && K == 1) { launch_mfma_serial_t<1, 20, 1>(in, tabs, ntables); return true; }
    if (in.NT == 2 && S == 61 && K == 1) { launch_mfma_serial_t<2, 61, 1>(in, tabs, ntables); return true; }
    if (in.NT == 1 && K == 1) { launch_mfma_serial_t<1, 0, 1>(in, tabs, ntables); return true; }
    if (in.NT == 1 && K == 2) { launch_mfma_serial_t<1, 0, 2>(in, tabs, ntables); return true; }
    if (in.NT == 1 && K == 4) { launch_mfma_serial_t<1, 0, 4>(in, tabs, ntables); return true; }
    if (in.NT == 2 && K == 1) { launch_mfma_serial_t<2, 0, 1>(in, tabs, ntables); return true; }
    if (in.NT == 2 && K == 2) { launch_mfma_serial_t<2, 0, 2>(in, tabs, ntables); return true; }
    return false;
}
static bool launch_mfma(Instance& in, const PartialsOp* ops, int count, int32_t* cum)
{
    const int S = in.S, K = in.K;
    if (!in.mfmaWhole) {                 // default: one wave per factor tile (MBAMD_MFMA_WHOLE=1 selects the wave-per-tile-column kernel)
        OpTables tabs;
        std::memset(&tabs, 0, sizeof tabs);
        tabs.ops[0] = ops;
        tabs.cum[0] = cum;
        for (int t = 1; t <= MBAMD_MAX_TABLES; ++t) tabs.start[t] = 1 << 30;
        if (launch_mfma_split(in, tabs, count)) return true;
    }
    if (in.NT == 1) {
        if (S == 20 && K == 4) launch_mfma_t<1, 20, 4>(in, ops, count, cum);
        else if (S == 20 && K == 1) launch_mfma_t<1, 20, 1>(in, ops, count, cum);
        else if (K == 1) launch_mfma_t<1, 0, 1>(in, ops, count, cum);
        else if (K == 2) launch_mfma_t<1, 0, 2>(in, ops, count, cum);
        else if (K == 3) launch_mfma_t<1, 0, 3>(in, ops, count, cum);
        else if (K == 4) launch_mfma_t<1, 0, 4>(in, ops, count, cum);
        else return false;
    } else {
        if (S == 61 && K == 1) launch_mfma_t<2, 61, 1>(in, ops, count, cum);
        else if (K == 1) launch_mfma_t<2, 0, 1>(in, ops, count, cum);
        else if (K == 2) launch_mfma_t<2, 0, 2>(in, ops, count, cum);
        else return false;
    }
    return true;
}

template <int SP_, int FK_>
static void launch_gen(Instance& in, const PartialsOp* ops, int count, int32_t* cum)
{
    auto kern = k_partials_gen<SP_, FK_>;
    MBAMD_LAUNCH(kern, dim3(in.Ppad / 64, count), 64, 0, in.stream, ops, in.S, in.K, in.Ppad, cum);
}

// General path: order the operations by dependency level (RAW, WAR and WAW on buffer indices) and
// launch one grid per level.
int Instance::buildGeneric(Plan& plan, std::vector<PartialsOp>& dev, const std::vector<int>& dstIdx,
                           const std::vector<int>& c1Idx, const std::vector<int>& c2Idx)
{
    const int n = (int) dev.size();
    std::vector<int> lastWrite(nBuffers, -1), lastRead(nBuffers, -1), level(n, 0);
    // scale buffers are dependencies too: an operation that divides by the factors of a buffer (SCALE_READ) must run after
    // the operation of this list that writes them, a second writer after the first writer and all its readers
    std::unordered_map<const void*, std::pair<int, int>> scaleLevels;     // scale buffer -> (last write level, last read level)
    int nLevels = 0;
    for (int o = 0; o < n; ++o) {
        int l = 0;
        l = std::max(l, lastWrite[c1Idx[o]] + 1);
        l = std::max(l, lastWrite[c2Idx[o]] + 1);
        l = std::max(l, lastWrite[dstIdx[o]] + 1);
        l = std::max(l, lastRead[dstIdx[o]] + 1);
        if (dev[o].scale_mode != SCALE_NONE) {
            auto it = scaleLevels.find(dev[o].scale);
            if (it != scaleLevels.end()) {
                l = std::max(l, it->second.first + 1);
                if (dev[o].scale_mode == SCALE_WRITE) l = std::max(l, it->second.second + 1);
            }
        }
        level[o] = l;
        if (dev[o].scale_mode != SCALE_NONE) {
            auto& sl = scaleLevels.emplace(dev[o].scale, std::make_pair(-1, -1)).first->second;
            if (dev[o].scale_mode == SCALE_WRITE) sl.first = l; else sl.second = std::max(sl.second, l);
        }
        lastWrite[dstIdx[o]] = l;
        lastRead[c1Idx[o]] = std::max(lastRead[c1Idx[o]], l);
        lastRead[c2Idx[o]] = std::max(lastRead[c2Idx[o]], l);
        nLevels = std::max(nLevels, l + 1);
    }
    std::vector<int> start(nLevels + 1, 0);
    for (int o = 0; o < n; ++o) start[level[o] + 1]++;
    for (int l = 0; l < nLevels; ++l) start[l + 1] += start[l];
    std::vector<PartialsOp> sorted(n);
    {
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (int o = 0; o < n; ++o) sorted[fill[level[o]]++] = dev[o];
    }
    // level 0: operations on two compact tips first (they get their own kernel)
    auto tipPair = [](const PartialsOp& d) { return d.c1_kind == CHILD_STATES && d.c2_kind == CHILD_STATES; };
    plan.tipTip = nLevels > 0 ? (int) (std::stable_partition(sorted.begin(), sorted.begin() + start[1], tipPair) - sorted.begin()) : 0;
    plan.anyScale = false;
    for (const PartialsOp& d : sorted) plan.anyScale |= d.scale_mode != SCALE_NONE;
    plan.start = start;
    plan.narrow = serialRatio > 0 && n <= serialRatio * nLevels;
    // Independent sub-lists.  A list often is several root-ward paths interleaved (MrBayes puts the operations of all
    // eigen-system parts of a codon model into one list, reference src/mbbeagle.c:1029-1100).  chainsOf() splits a
    // subset of the list (original indices, list order) into connected components of the "touches a buffer a member
    // writes" relation and packs them into at most MBAMD_MAX_TABLES bins; the serial kernel walks the bins side by side.
    auto chainsOf = [&](const std::vector<int>& sub) {
        const int m = (int) sub.size();
        std::vector<int> comp(m);
        for (int x = 0; x < m; ++x) comp[x] = x;
        auto find = [&](int x) { while (comp[x] != x) x = comp[x] = comp[comp[x]]; return x; };
        auto unite = [&](int a, int b) { a = find(a); b = find(b); if (a != b) comp[std::max(a, b)] = std::min(a, b); };
        std::vector<int> owner(nBuffers, -1);                    // a member that writes the buffer
        for (int x = 0; x < m; ++x) {
            const int o = sub[x];
            if (owner[dstIdx[o]] >= 0) unite(x, owner[dstIdx[o]]);
            owner[dstIdx[o]] = x;
        }
        for (int x = 0; x < m; ++x) {
            const int o = sub[x];
            if (owner[c1Idx[o]] >= 0) unite(x, owner[c1Idx[o]]);
            if (owner[c2Idx[o]] >= 0) unite(x, owner[c2Idx[o]]);
        }
        for (int x = 0; x < m; ++x)                              // node scale buffers written by one, used by another
            for (int y = x + 1; y < m; ++y) {
                const PartialsOp &dx = dev[sub[x]], &dy = dev[sub[y]];
                if (dx.scale == dy.scale && dx.scale_mode != SCALE_NONE && dy.scale_mode != SCALE_NONE &&
                    (dx.scale_mode == SCALE_WRITE || dy.scale_mode == SCALE_WRITE))
                    unite(x, y);
            }
        std::vector<int> roots, size(m, 0);
        for (int x = 0; x < m; ++x) { size[find(x)]++; if (find(x) == x) roots.push_back(x); }
        std::sort(roots.begin(), roots.end(), [&](int a, int b) { return size[a] > size[b]; });
        const int nb = std::min<int>(MBAMD_MAX_TABLES, (int) roots.size());
        std::vector<int> binLen(nb, 0), binOf(m, 0);
        for (int r : roots) {                                    // largest first, each into the currently shortest bin
            const int bsel = (int) (std::min_element(binLen.begin(), binLen.end()) - binLen.begin());
            binOf[r] = bsel;
            binLen[bsel] += size[r];
        }
        std::vector<std::vector<int>> bins(nb);
        for (int x = 0; x < m; ++x) bins[binOf[find(x)]].push_back(sub[x]);
        return bins;
    };
    auto appendBins = [&](const std::vector<std::vector<int>>& bins, std::vector<std::pair<int, int>>& out) {
        for (const auto& bin : bins) {
            if (bin.empty()) continue;
            out.emplace_back((int) sorted.size(), (int) bin.size());
            for (int o : bin) sorted.push_back(dev[o]);          // (list order inside a bin = dependency order)
        }
    };
    plan.chains.clear();
    plan.spineChains.clear();
    plan.serialFrom = nLevels;
    if (plan.narrow) {
        std::vector<int> all(n);
        for (int o = 0; o < n; ++o) all[o] = o;
        const auto bins = chainsOf(all);
        if (bins.size() > 1) appendBins(bins, plan.chains);
        else plan.chains.emplace_back(0, n);
    } else if (serialRatio > 0) {
        // the tail of a level-launched list: trailing levels of a few operations each.  If they fall apart into parallel
        // chains (three codon parts -> three chains) one serial launch walks them side by side; otherwise only the
        // strictly single-operation levels (the spine towards the root) go serial.
        int from = nLevels;
        while (from > 0 && start[from] - start[from - 1] <= MBAMD_MAX_TABLES) from--;
        if (nLevels - from >= 2) {
            std::vector<int> sub;
            for (int o = 0; o < n; ++o) if (level[o] >= from) sub.push_back(o);
            const auto bins = chainsOf(sub);
            size_t longest = 0;
            for (const auto& bin : bins) longest = std::max(longest, bin.size());
            if (bins.size() >= 2 && 4 * longest <= 5 * (size_t) (nLevels - from)) {
                plan.serialFrom = from;
                appendBins(bins, plan.spineChains);
            }
        }
        if (plan.spineChains.empty()) {
            from = nLevels;
            while (from > 0 && start[from] - start[from - 1] <= spineWidth) from--;
            if (nLevels - from >= 2) {
                plan.serialFrom = from;
                plan.spineChains.emplace_back(start[from], n - start[from]);
            }
        }
    }
    return planTable(plan, sorted);
}

int Instance::runGeneric(const Plan& plan, int32_t* cum)
{
    const std::vector<int>& start = plan.start;
    const int nLevels = (int) start.size() - 1;
    const bool anyScale = plan.anyScale;
    if (plan.narrow && mfma && !mfmaWhole) {
        OpTables tabs;
        std::memset(&tabs, 0, sizeof tabs);
        int nt = 0;
        for (auto& ch : plan.chains) {
            tabs.ops[nt] = plan.d_table + ch.first;
            tabs.cum[nt] = cum;
            tabs.start[nt] = ch.second;
            ++nt;
        }
        if (launch_mfma_serial(*this, tabs, nt)) {
            pendingLaunches += 1;
            HIP_TRY(hipGetLastError());
            return BEAGLE_SUCCESS;
        }
    }
    int levelEnd = nLevels;
    if (mfma && !mfmaWhole) levelEnd = plan.serialFrom;
    for (int l = 0; l < levelEnd; ++l) {
        int off = start[l];
        int remaining = start[l + 1] - start[l];
        if (l == 0 && mfma && !mfmaWhole && plan.tipTip > 0 && plan.tipTip <= 8192) {
            OpTables tabs;
            std::memset(&tabs, 0, sizeof tabs);
            tabs.ops[0] = plan.d_table;
            tabs.cum[0] = cum;
            for (int t = 1; t <= MBAMD_MAX_TABLES; ++t) tabs.start[t] = 1 << 30;
            if (launch_tips(*this, tabs, plan.tipTip)) {
                pendingLaunches += 1;
                off += plan.tipTip;
                remaining -= plan.tipTip;
            }
        }
        while (remaining > 0) {
            const int count = std::min(remaining, 32768);
            const PartialsOp* ops = plan.d_table + off;
            bool fused = true;
            if (mfma && launch_mfma(*this, ops, std::min(count, 8192), cum)) {
                const int done = std::min(count, 8192);
                pendingLaunches += 1;
                off += done;
                remaining -= done;
                continue;
            }
            if (SP == 20 && K == 4) launch_gen<20, 4>(*this, ops, count, cum);
            else if (SP == 20 && K == 1) launch_gen<20, 1>(*this, ops, count, cum);
            else if (SP == 64 && K == 1) launch_gen<64, 1>(*this, ops, count, cum);
            else if (SP == 4 && K == 4) launch_gen<4, 4>(*this, ops, count, cum);
            else if (SP == 4 && K == 1) launch_gen<4, 1>(*this, ops, count, cum);
            else {
                fused = false;
                switch (SP) {
                    case 4: launch_gen<4, 0>(*this, ops, count, cum); break;
                    case 8: launch_gen<8, 0>(*this, ops, count, cum); break;
                    case 16: launch_gen<16, 0>(*this, ops, count, cum); break;
                    case 20: launch_gen<20, 0>(*this, ops, count, cum); break;
                    case 32: launch_gen<32, 0>(*this, ops, count, cum); break;
                    default: launch_gen<64, 0>(*this, ops, count, cum); break;
                }
            }
            pendingLaunches += 1;
            if (!fused && anyScale) {
                MBAMD_LAUNCH(k_rescale_gen, dim3(Ppad / 64, count), 64, 0, stream, ops, S, K, Ppad, cum);
                pendingLaunches += 1;
            }
            off += count;
            remaining -= count;
        }
    }
    if (levelEnd < nLevels) {                    // the spine: one launch walks it
        OpTables tabs;
        std::memset(&tabs, 0, sizeof tabs);
        int nt = 0;
        for (auto& ch : plan.spineChains) {
            tabs.ops[nt] = plan.d_table + ch.first;
            tabs.cum[nt] = cum;
            tabs.start[nt] = ch.second;
            ++nt;
        }
        if (!launch_mfma_serial(*this, tabs, nt)) return fail(BEAGLE_ERROR_GENERAL, "no serial MFMA kernel for this shape");
        pendingLaunches += 1;
    }
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int Instance::accumulate(const int* idx, int n, int cumIdx, int sign, bool fresh)
{
    if (cumIdx < 0 || cumIdx >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "scale factors: cumulative index");
    if (n <= 0) return BEAGLE_SUCCESS;
    int rc = ensureScale(cumIdx);
    if (rc) return rc;
    std::vector<const int32_t*> ptrs(n);
    for (int i = 0; i < n; ++i) {
        if (idx[i] < 0 || idx[i] >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "scale factors: index");
        rc = ensureScale(idx[i]);
        if (rc) return rc;
        ptrs[i] = scale[idx[i]];
    }
    const int32_t* const* dptrs = nullptr;
    rc = stageDirect(ptrs.data(), sizeof(void*) * n, (const void**) &dptrs);
    if (rc) return rc;
    MBAMD_LAUNCH_BARRIER(k_scale_accumulate, (unsigned) ((Ppad + 255) / 256), 256, 0, stream, dptrs, n, sign,
                 Ppad, scale[cumIdx], fresh ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int Instance::runDeferredReset()
{
    const int idx = deferredReset;
    deferredReset = -1;
    if (idx < 0 || idx >= nScale || !scale[idx]) return BEAGLE_SUCCESS;
    MBAMD_LAUNCH(k_scale_copy, (unsigned) ((Ppad + 255) / 256), 256, 0, stream, (const int32_t*) nullptr, Ppad, scale[idx]);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

// 4-state path: sources are node-exponent buffers of the arena (int8 per pattern and category) or cumulative ones
int Instance::accumulate4(const int* idx, int n, int cumIdx, int sign)
{
    if (cumIdx < 0 || cumIdx >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "scale factors: cumulative index");
    if (n <= 0) return BEAGLE_SUCCESS;
    std::vector<ExpSource> src;
    src.reserve(n);
    for (int i = 0; i < n; ++i) {
        if (idx[i] < 0 || idx[i] >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "scale factors: index");
        if (scaleState[idx[i]] == 0) continue;                       // never written: zero
        ExpSource e;
        e.wide = scaleState[idx[i]] == 2 ? wideScale[idx[i]] : nullptr;
        e.narrow = idx[i];
        e.pad_ = 0;
        src.push_back(e);
    }
    // (arena buffers in front of the wide ones: the kernel sums them without a branch; the order of an integer sum is free)
    const int nNarrow = (int) (std::stable_partition(src.begin(), src.end(), [](const ExpSource& e) { return e.wide == nullptr; }) - src.begin());
    // a freshly reset cumulative buffer (MrBayes-style rescaling: Reset + Accumulate of every node): the kernel STORES, no zero fill
    const bool fresh = scaleState[cumIdx] == 0 && !src.empty();
    int rc = BEAGLE_SUCCESS;
    if (fresh) {
        if (!wideScale[cumIdx]) HIP_TRY(hipMalloc(&wideScale[cumIdx], (size_t) K * Ppad * sizeof(int32_t)));
        scaleState[cumIdx] = 2;
    } else {
        rc = ensureWide(cumIdx);
        if (rc) return rc;
    }
    if (src.empty()) return BEAGLE_SUCCESS;
    const ExpSource* dsrc = nullptr;
    rc = stageDirect(src.data(), sizeof(ExpSource) * src.size(), (const void**) &dsrc);
    if (rc) return rc;
    MBAMD_LAUNCH_BARRIER(k_exp_accumulate, (unsigned) (((size_t) K * Ppad + 255) / 256), 256, 0, stream, dsrc, (int) src.size(), nNarrow, sign, K, Ppad,
                 (const int8_t*) arenaExp, estride, wideScale[cumIdx], fresh ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int Instance::integrate(const int* parent, const int* child, const int* prob, const int* wIdx, const int* fIdx,
                        const int* cumIdx, int count, double* out)
{
    if (count < 1 || count > MBAMD_MAX_SUBSETS) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "log-likelihood: subset count");
    armSums();
    if (arena()) {
        int rc;
        if (heldPath && count == 1 && parent[0] == heldPathDst && !(child && tipStates[child[0]] == nullptr && child[0] == heldPathDst)) {
            rc = integratePath4(parent, child, prob, wIdx, fIdx, cumIdx);      // the held path and this integration: one launch
        } else {
            if (heldPath) { rc = runHeldPath(); if (rc) return rc; }
            rc = integrate4(parent, child, prob, wIdx, fIdx, cumIdx, count);
        }
        if (rc) return rc;
        rc = spanEnd();
        if (rc) return rc;
        postResultFlag();
        haveSite = true;
        pendingResult = true;
        if (deferred) {
            if (out) *out = 0.0;
            return BEAGLE_SUCCESS;
        }
        return fetchResult(out);
    }
    IntegrateArgs a;
    std::memset(&a, 0, sizeof a);
    a.count = count;
    for (int n = 0; n < count; ++n) {
        if (parent[n] < 0 || parent[n] >= nBuffers || !valid[parent[n]])
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "log-likelihood: parent buffer");
        a.parent[n] = partials[parent[n]];
        if (child) {
            const int ci = child[n];
            if (ci < 0 || ci >= nBuffers || prob[n] < 0 || prob[n] >= nMatrices)
                return fail(BEAGLE_ERROR_OUT_OF_RANGE, "edge log-likelihood: child buffer / matrix");
            if (tipStates[ci]) { a.child[n] = tipStates[ci]; a.child_kind[n] = CHILD_STATES; }
            else if (valid[ci]) { a.child[n] = partials[ci]; a.child_kind[n] = CHILD_PARTIALS; }
            else return fail(BEAGLE_ERROR_OUT_OF_RANGE, "edge log-likelihood: child buffer was never written");
            a.matrix[n] = matrixPtr(prob[n]);
        }
        if (wIdx[n] < 0 || wIdx[n] >= nEigen || fIdx[n] < 0 || fIdx[n] >= nEigen)
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "log-likelihood: weights / frequencies index");
        a.weights[n] = d_weights + (size_t) wIdx[n] * K;
        a.freqs[n] = d_freqs + (size_t) fIdx[n] * S;
        if (cumIdx && cumIdx[n] != BEAGLE_OP_NONE) {
            if (cumIdx[n] < 0 || cumIdx[n] >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "log-likelihood: cumulative scale index");
            int rc = ensureScale(cumIdx[n]);
            if (rc) return rc;
            a.cum[n] = scale[cumIdx[n]];
        }
    }
    double* const siteOut = (siteToHost && h_site_dev) ? h_site_dev : d_site;
    siteOnHost = siteOut != d_site;
    if (S >= 8) MBAMD_LAUNCH_BARRIER(k_integrate_lnl_wide, (unsigned) nblocks, 256, 0, stream, a, S, SP, K, P, Ppad, (const double*) d_pweights, siteOut, h_sums_dev);
    else
        MBAMD_LAUNCH(k_integrate_lnl, (unsigned) nblocks, 64, 0, stream, a, S, SP, K, P, Ppad, (const double*) d_pweights, siteOut, h_sums_dev);
    HIP_TRY(hipGetLastError());
    { int src = spanEnd(); if (src) return src; }
    postResultFlag();
    haveSite = true;
    pendingResult = true;
    if (deferred) {
        if (out) *out = 0.0;
        return BEAGLE_SUCCESS;
    }
    return fetchResult(out);
}

// the held root-ward path (k_path4 plan) and the log-likelihood over its last result as ONE launch (k_path4_lnl, mbamd_walk4.h)
int Instance::integratePath4(const int* parent, const int* child, const int* prob, const int* wIdx, const int* fIdx, const int* cumIdx)
{
    Plan* plan = heldPath;
    PathLnl4 t;
    std::memset(&t, 0, sizeof t);
    bool ok = parent[0] >= 0 && parent[0] < nBuffers && valid[parent[0]] && !tipStates[parent[0]];
    if (ok && child) {
        const int ci = child[0];
        ok = ci >= 0 && ci < nBuffers && prob[0] >= 0 && prob[0] < nMatrices && (tipStates[ci] || valid[ci]);
        if (ok) {
            if (tipStates[ci]) { t.child = tipStates[ci]; t.child_kind = CHILD_STATES; }
            else { t.child = partials[ci]; t.child_kind = CHILD_PARTIALS; }
            t.matrix = matrixPtr(prob[0]);
        }
    }
    ok = ok && wIdx[0] >= 0 && wIdx[0] < nEigen && fIdx[0] >= 0 && fIdx[0] < nEigen;
    if (ok && cumIdx && cumIdx[0] != BEAGLE_OP_NONE) ok = cumIdx[0] >= 0 && cumIdx[0] < nScale;
    if (!ok) {                                   // (let the separate kernels report what is wrong, in their own words)
        int rc = runHeldPath();
        if (rc) return rc;
        return integrate4(parent, child, prob, wIdx, fIdx, cumIdx, 1);
    }
    t.weights = d_weights + (size_t) wIdx[0] * K;
    t.freqs = d_freqs + (size_t) fIdx[0] * S;
    if (cumIdx && cumIdx[0] != BEAGLE_OP_NONE && scaleState[cumIdx[0]] != 0) {
        int rc = ensureWide(cumIdx[0]);
        if (rc) return rc;
        t.cum = wideScale[cumIdx[0]];
    }
    double* const siteOut = (siteToHost && h_site_dev) ? h_site_dev : d_site;
    siteOnHost = siteOut != d_site;
    t.pattern_weights = d_pweights;
    t.site = siteOut;
    t.wsite = h_sums_dev;
    t.P = P;
    heldPath = nullptr;
    fusedPaths++;
    plan->lastLaunch = ++launchClock;
    { int src = spanBegin(); if (src) return src; }
    Walk4ArgsInline ai;
    Walk4Args& a = ai.a;
    a.prog = nullptr;
    a.entries = (int) plan->inlineProg.size();
    a.nslots = 0;
    a.partials = reinterpret_cast<f4*>(arenaPartials);
    a.pstride = geom.pstride;
    a.tips = arenaTips;
    a.tstride = geom.tstride;
    a.exps = arenaExp;
    a.estride = estride;
    a.matrices = matrices;
    a.cum = heldPathCum;
    a.cumFresh = heldPathFresh ? 1 : 0;
    a.K = K;
    a.Ppad = Ppad;
    a.nblocks = Ppad / 64;
    a.tail = 0;
    std::memcpy(ai.inl, plan->inlineProg.data(), plan->inlineProg.size() * sizeof(Walk4Entry));
    hipEvent_t ev0{}, ev1{};
    if (timing) {
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipEventRecord(ev0, stream));
    }
    auto kernel = k_path4_lnl<Walk4ArgsInline>;
    MBAMD_LAUNCH_BARRIER(kernel, 8u * (unsigned) ((Ppad / 64 + 7) / 8), 64 * K, path4_lnl_lds_bytes(a.entries, K), stream, ai, t);
    HIP_TRY(hipGetLastError());
    if (timing) {
        HIP_TRY(hipEventRecord(ev1, stream));
        events.emplace_back(ev0, ev1);
    }
    pendingLaunches += 1;
    return BEAGLE_SUCCESS;
}

int Instance::integrate4(const int* parent, const int* child, const int* prob, const int* wIdx, const int* fIdx,
                         const int* cumIdx, int count)
{
    IntegrateArgs4 a;
    std::memset(&a, 0, sizeof a);
    a.count = count;
    for (int n = 0; n < count; ++n) {
        if (parent[n] < 0 || parent[n] >= nBuffers || !valid[parent[n]] || tipStates[parent[n]])
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "log-likelihood: parent buffer");
        a.parent[n] = reinterpret_cast<const f4*>(partials[parent[n]]);
        if (child) {
            const int ci = child[n];
            if (ci < 0 || ci >= nBuffers || prob[n] < 0 || prob[n] >= nMatrices)
                return fail(BEAGLE_ERROR_OUT_OF_RANGE, "edge log-likelihood: child buffer / matrix");
            if (tipStates[ci]) { a.child[n] = tipStates[ci]; a.child_kind[n] = CHILD_STATES; }
            else if (valid[ci]) { a.child[n] = partials[ci]; a.child_kind[n] = CHILD_PARTIALS; }
            else return fail(BEAGLE_ERROR_OUT_OF_RANGE, "edge log-likelihood: child buffer was never written");
            a.matrix[n] = matrixPtr(prob[n]);
        }
        if (wIdx[n] < 0 || wIdx[n] >= nEigen || fIdx[n] < 0 || fIdx[n] >= nEigen)
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "log-likelihood: weights / frequencies index");
        a.weights[n] = d_weights + (size_t) wIdx[n] * K;
        a.freqs[n] = d_freqs + (size_t) fIdx[n] * S;
        if (cumIdx && cumIdx[n] != BEAGLE_OP_NONE) {
            if (cumIdx[n] < 0 || cumIdx[n] >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "log-likelihood: cumulative scale index");
            if (scaleState[cumIdx[n]] != 0) {
                int rc = ensureWide(cumIdx[n]);
                if (rc) return rc;
                a.cum[n] = wideScale[cumIdx[n]];
            }
        }
    }
    double* const siteOut = (siteToHost && h_site_dev) ? h_site_dev : d_site;
    siteOnHost = siteOut != d_site;
    if (wg) {
        WgGeom g;
        g.tileFloats = wgTileBytes / 4; g.tipTileBytes = wgTipTileBytes; g.TP = wg_pairs_padded(S);
        MBAMD_LAUNCH_BARRIER(k_integrate_lnl_wg_wide, (unsigned) nblocks, MBAMD_INTEGRATE_WG_THREADS, 0, stream, a, S, SP, K, P, Ppad, g, (const double*) d_pweights, siteOut, h_sums_dev);
    } else {
        MBAMD_LAUNCH(k_integrate_lnl_s4, (unsigned) nblocks, 64, 0, stream, a, K, P, Ppad, geom, (const double*) d_pweights, siteOut, h_sums_dev);
    }
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

// the block sums as their own completion signal: fill them with the pattern fetchResult waits to see overwritten
void Instance::armSums()
{
    sumsArmed = pollSums && pollResult && !pendingResult && !deferred;
    if (!sumsArmed) return;
    uint64_t* p = reinterpret_cast<uint64_t*>(h_sums);
    for (int i = 0; i < nblocks; ++i) p[i] = kSumSentinel;
    __atomic_thread_fence(__ATOMIC_RELEASE);
}

// the stream writes the sequence number of this integration behind its kernel: what fetchResult polls
void Instance::postResultFlag()
{
    if (!pollResult) return;
    flagWritten = !sumsArmed;                    // (a result awaited through its block sums needs no stream operation behind the kernel)
    if (!flagWritten) ++flagSeq;                 // (the sequence moves on: nobody will see this number in the flag word, a later one is larger)
    else if (hipStreamWriteValue32(stream, h_flag_dev, ++flagSeq, 0) != hipSuccess) {
        (void) hipGetLastError();
        pollResult = false;
    }
    flagClock = launchClock;                     // what the stream has finished when the flag shows flagSeq -- and nothing younger
    siteSeq = flagSeq;                           // (the integration in front of this flag wrote the site values)
}

int Instance::fetchResult(double* out)
{
    if (!pendingResult) return fail(BEAGLE_ERROR_GENERAL, "no log-likelihood pending");
    {
        StatTimer st_(ST_WAIT);
        bool landed = false, bySums = false;
        if (sumsArmed) {
            // every block of the integration kernel ends with ONE store of its sum: when all have changed, the kernel has done its work
            const volatile uint64_t* p = reinterpret_cast<const volatile uint64_t*>(h_sums);
            const auto t0 = std::chrono::steady_clock::now();
            int i = 0;
            for (long spins = 0; i < nblocks; ++spins) {
                if (p[i] != kSumSentinel) { ++i; continue; }
                if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) break;
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
            }
            landed = bySums = i == nblocks;
            if (landed) __atomic_thread_fence(__ATOMIC_ACQUIRE);
            sumsArmed = false;
        }
        if (!landed && pollResult && flagWritten) {
            // spin on the word the stream writes behind the integration kernel, for about a millisecond of wall time; then the runtime's wait
            volatile uint32_t* f = h_flag;
            const auto t0 = std::chrono::steady_clock::now();
            for (long spins = 0; !(landed = (*f == flagSeq)); ++spins) {
                if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) break;
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
            }
            if (landed) __atomic_thread_fence(__ATOMIC_ACQUIRE);      // the block sums were written before the flag: read them after it
        }
        // the flag covers the launches up to the integration it follows; launches queued behind it in deferred mode (the reduction of
        // mbamdReduceLogLikelihood, further lists) are complete only after a real synchronisation
        if (landed) syncedClock = std::max(syncedClock, flagClock);
        else { HIP_TRY(hipStreamSynchronize(stream)); syncedClock = launchClock; }
        // (seen through the sums: the kernel's other stores -- the site values -- may still be on their way; getSites then synchronises)
        if (!bySums || *reinterpret_cast<volatile uint32_t*>(h_flag) == flagSeq) seenSeq = flagSeq;
    }
    pendingResult = false;
    double s = 0.0;
    for (int i = 0; i < nblocks; ++i) s += h_sums[i];
    if (out) *out = s;
    if (!(s == s) || s > 1.79e308 || s < -1.79e308) return BEAGLE_ERROR_FLOATING_POINT;
    return BEAGLE_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// resources
// ---------------------------------------------------------------------------------------------
static BeagleResourceList g_resources = {nullptr, 0};
static std::vector<BeagleResource> g_resourceVec;
static std::vector<std::string> g_resourceNames, g_resourceDescs;
static const long kSupport = BEAGLE_FLAG_PRECISION_SINGLE | BEAGLE_FLAG_COMPUTATION_SYNCH | BEAGLE_FLAG_EIGEN_REAL |
                             BEAGLE_FLAG_SCALING_MANUAL | BEAGLE_FLAG_SCALING_ALWAYS | BEAGLE_FLAG_SCALING_DYNAMIC |
                             BEAGLE_FLAG_SCALERS_LOG | BEAGLE_FLAG_VECTOR_NONE | BEAGLE_FLAG_THREADING_NONE |
                             BEAGLE_FLAG_PROCESSOR_GPU | BEAGLE_FLAG_INVEVEC_STANDARD | BEAGLE_FLAG_FRAMEWORK_HIP;

static void buildResources()
{
    if (g_resources.list) return;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    g_resourceNames.resize(n);
    g_resourceDescs.resize(n);
    g_resourceVec.resize(std::max(n, 1));
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        std::memset(&prop, 0, sizeof prop);
        (void) hipGetDeviceProperties(&prop, i);
        g_resourceNames[i] = prop.name;
        char buf[256];
        std::snprintf(buf, sizeof buf, "HIP device %d (%s), %.0f GiB, %d CUs", i, prop.gcnArchName,
                      (double) prop.totalGlobalMem / (1 << 30), prop.multiProcessorCount);
        g_resourceDescs[i] = buf;
        g_resourceVec[i].name = const_cast<char*>(g_resourceNames[i].c_str());
        g_resourceVec[i].description = const_cast<char*>(g_resourceDescs[i].c_str());
        g_resourceVec[i].supportFlags = kSupport | BEAGLE_FLAG_PRECISION_DOUBLE;
        g_resourceVec[i].requiredFlags = 0;
    }
    g_resources.list = g_resourceVec.data();
    g_resources.length = n;
}

}  // namespace mbamd

// ---------------------------------------------------------------------------------------------
// per-pattern read-outs as Instance methods (the facade gathers them from its children)
// ---------------------------------------------------------------------------------------------
namespace mbamd {

int Instance::getSites(double* out)
{
    if (!haveSite) return fail(BEAGLE_ERROR_GENERAL, "beagleGetSiteLogLikelihoods: no likelihood computed yet");
    // (a result that was fetched -- the stream's flag behind the integration kernel was seen, or the stream synchronised -- has its
    //  site values in place: no second wait; a runtime synchronisation of an idle stream still costs ~25 us)
    if (!(siteOnHost && siteSeq != 0 && seenSeq == siteSeq && !pendingResult)) HIP_TRY(hipStreamSynchronize(stream));
    if (siteOnHost) {
        std::memcpy(out, h_site, (size_t) P * sizeof(double));
    } else {
        HIP_TRY(hipMemcpy(out, d_site, (size_t) P * sizeof(double), hipMemcpyDeviceToHost));
        if (!h_site && hipHostMalloc((void**) &h_site, (size_t) Ppad * sizeof(double), hipHostMallocDefault) == hipSuccess) {
            if (hipHostGetDevicePointer((void**) &h_site_dev, h_site, 0) != hipSuccess) h_site_dev = nullptr;
        }
        siteToHost = h_site_dev != nullptr && !noSiteHost;   // this client reads them: later evaluations write to the host directly
    }
    return BEAGLE_SUCCESS;
}

// the binary exponents behind a scale buffer, out[k * P + c]
int Instance::getScaleExponents(int idx, int* out)
{
    if (idx < 0 || idx >= nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "scale exponents: index");
    HIP_TRY(hipStreamSynchronize(stream));
    if (!arena()) {
        int rc = ensureScale(idx);
        if (rc) return rc;
        std::vector<int32_t> h(Ppad);
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(h.data(), scale[idx], (size_t) Ppad * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int k = 0; k < K; ++k) for (int c = 0; c < P; ++c) out[(size_t) k * P + c] = h[c];
        return BEAGLE_SUCCESS;
    }
    const int st = scaleState[idx];
    if (st == 0) { std::fill(out, out + (size_t) K * P, 0); return BEAGLE_SUCCESS; }
    std::vector<int32_t> h((size_t) K * Ppad);
    if (st == 2) {
        HIP_TRY(hipMemcpy(h.data(), wideScale[idx], h.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    } else {
        int rc = grow(&d_tmp, &tmpCap, h.size() * sizeof(int32_t));
        if (rc) return rc;
        MBAMD_LAUNCH(k_exp_widen, (unsigned) ((h.size() + 255) / 256), 256, 0, stream, (const int8_t*) arenaExp, estride, idx, K, Ppad,
                     (int32_t*) d_tmp);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(h.data(), d_tmp, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    for (int k = 0; k < K; ++k) for (int c = 0; c < P; ++c) out[(size_t) k * P + c] = h[(size_t) k * Ppad + c];
    return BEAGLE_SUCCESS;
}

// ---- reports (csrc/mbamd_reports.h, include/libhmsbeagle/mbamd_reports.h) -------------------------------------------
int Instance::finalPass(const MbamdFinalOperation* ops, int count)
{
    if (S > MBAMD_REP_MAXS) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdUpdateFinalPartials: more than 64 states");
    for (int o = 0; o < count; ++o) {
        const MbamdFinalOperation& b = ops[o];
        if (b.destinationPartials < 0 || b.destinationPartials >= nBuffers || b.downPartials < 0 || b.downPartials >= nBuffers ||
            b.ancestorFinal >= nBuffers || b.rootTip >= nBuffers)
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdUpdateFinalPartials: partials index");
        if (b.transitionMatrix < 0 || b.transitionMatrix >= nMatrices) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdUpdateFinalPartials: matrix index");
        if (!valid[b.downPartials] || tipStates[b.downPartials]) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdUpdateFinalPartials: down partials were never computed");
        if (b.ancestorFinal >= 0 && (!valid[b.ancestorFinal] || tipStates[b.ancestorFinal]))
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdUpdateFinalPartials: the ancestor's final partials are not there");
        if (tipStates[b.destinationPartials]) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdUpdateFinalPartials: the destination holds compact tip states");
        int rc = ensurePartials(b.destinationPartials);
        if (rc) return rc;
        // the exponents of the final pass: written by the top node's launch, inherited by everything below it
        if (finalExpOf.size() != (size_t) nBuffers) finalExpOf.assign((size_t) nBuffers, nullptr);
        if (b.ancestorFinal < 0) {
            int32_t*& own = finalExpOwn[b.destinationPartials];
            if (!own) HIP_TRY(hipMalloc(&own, (size_t) K * Ppad * sizeof(int32_t)));
            // a new pass from this top node rewrites `own` in place: whatever an EARLIER pass left below it would be read with
            // this pass's exponents from now on -- those buffers no longer hold final partials (they are re-made by this pass, or not)
            for (size_t q = 0; q < finalExpOf.size(); ++q)
                if (finalExpOf[q] == own && (int) q != b.destinationPartials) finalExpOf[q] = nullptr;
            finalExpOf[b.destinationPartials] = own;
        } else {
            if (!finalExpOf[b.ancestorFinal]) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdUpdateFinalPartials: the ancestor's buffer does not hold final partials");
            finalExpOf[b.destinationPartials] = finalExpOf[b.ancestorFinal];
        }
        FinalOp f;
        std::memset(&f, 0, sizeof f);
        f.Ppad = Ppad;
        f.fexp = finalExpOf[b.destinationPartials];
        f.dst = partials[b.destinationPartials];
        f.anc = b.ancestorFinal >= 0 ? partials[b.ancestorFinal] : nullptr;
        f.down = partials[b.downPartials];
        f.matrix = matrixPtr(b.transitionMatrix);
        if (b.ancestorFinal < 0 && b.rootTip >= 0) {
            if (tipStates[b.rootTip]) { f.tip = tipStates[b.rootTip]; f.tipKind = CHILD_STATES; }
            else if (valid[b.rootTip]) { f.tip = partials[b.rootTip]; f.tipKind = CHILD_PARTIALS; }
            else return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdUpdateFinalPartials: the root tip was never set");
        }
        const dim3 grid((unsigned) ((P + 63) / 64), (unsigned) K);
        if (s4) MBAMD_LAUNCH(k_final_pass<1>, grid, 64, 0, stream, f, S, SP, K, P, (size_t) geom.pstride, (size_t) geom.tstride);
        else if (wg) MBAMD_LAUNCH(k_final_pass<2>, grid, 64, 0, stream, f, S, SP, K, P, (size_t) (wgTileBytes / 4), (size_t) wgTipTileBytes);
        else MBAMD_LAUNCH(k_final_pass<0>, grid, 64, 0, stream, f, S, SP, K, P, (size_t) geom.pstride, (size_t) 0);
        HIP_TRY(hipGetLastError());
        valid[b.destinationPartials] = 1;
    }
    return BEAGLE_SUCCESS;
}

int Instance::getScaledPartials(int idx, int cumIdx, float* out, float* outLn)
{
    if (idx < 0 || idx >= nBuffers || !valid[idx] || tipStates[idx]) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdGetScaledPartials: buffer");
    if (cumIdx != BEAGLE_OP_NONE && (cumIdx < 0 || cumIdx >= nScale)) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdGetScaledPartials: cumulative scale index");
    const int32_t *wide = nullptr, *narrow = nullptr;
    if (cumIdx != BEAGLE_OP_NONE) {
        if (arena()) {
            if (scaleState[cumIdx] != 0) {
                int rc = ensureWide(cumIdx);
                if (rc) return rc;
                wide = wideScale[cumIdx];
            }
        } else {
            int rc = ensureScale(cumIdx);
            if (rc) return rc;
            narrow = scale[cumIdx];
        }
    }
    const size_t total = (size_t) K * P * S;
    int rc = grow(&d_tmp, &tmpCap, (total + (size_t) Ppad) * sizeof(float));
    if (rc) return rc;
    float* d_out = static_cast<float*>(d_tmp);
    float* d_ln = d_out + total;
    const unsigned blocks = (unsigned) ((total + 255) / 256);
    const int32_t* extra = (idx < (int) finalExpOf.size()) ? finalExpOf[idx] : nullptr;      // final partials carry their pass's own exponents
    if (s4) MBAMD_LAUNCH(k_export_scaled<1>, blocks, 256, 0, stream, (const float*) partials[idx], wide, narrow, extra, S, K, P, Ppad, (size_t) geom.pstride, d_out, d_ln);
    else if (wg) MBAMD_LAUNCH(k_export_scaled<2>, blocks, 256, 0, stream, (const float*) partials[idx], wide, narrow, extra, S, K, P, Ppad, (size_t) (wgTileBytes / 4), d_out, d_ln);
    else MBAMD_LAUNCH(k_export_scaled<0>, blocks, 256, 0, stream, (const float*) partials[idx], wide, narrow, extra, S, K, P, Ppad, (size_t) geom.pstride, d_out, d_ln);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    syncedClock = launchClock;
    HIP_TRY(hipMemcpy(out, d_out, total * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(outLn, d_ln, (size_t) P * sizeof(float), hipMemcpyDeviceToHost));
    return BEAGLE_SUCCESS;
}


void Instance::destroyChildren()
{
    for (Child& ch : children) { ch.in->destroy(); delete ch.in; }
    children.clear();
}

// Split the patterns into child instances: every partition range (start, count) is cut into one shard per device of
// shardDevices (whole 64-pattern blocks, the last shard takes the remainder), then the recorded tip data and pattern
// weights are replayed into the children.
int Instance::makeChildren(const std::vector<std::pair<int, int>>& ranges)
{
    destroyChildren();
    const int G = std::max<int>(1, (int) shardDevices.size());
    for (size_t p = 0; p < ranges.size(); ++p) {
        const int start = ranges[p].first, count = ranges[p].second;
        const int blocks = (count + 63) / 64;
        const int g = std::max(1, std::min(G, blocks));
        int done = 0;
        for (int i = 0; i < g; ++i) {
            const int b0 = (int) ((long) blocks * i / g), b1 = (int) ((long) blocks * (i + 1) / g);
            const int n = (i == g - 1) ? count - done : (b1 - b0) * 64;
            if (n <= 0) continue;
            Instance* c = nullptr;
            const int dev = shardDevices.empty() ? device : shardDevices[i % shardDevices.size()];
            int rc = new_engine(c, flags, createArgs[0], createArgs[1], createArgs[2], createArgs[3], n, createArgs[5], createArgs[6],
                                createArgs[7], createArgs[8], dev);
            if (rc) { destroyChildren(); return rc; }
            c->logOpen = false;
            children.push_back(Child{c, start + done, n, (int) p});
            done += n;
        }
    }
    for (Child& ch : children) {
        (void) hipSetDevice(ch.in->device);
        for (auto& ts : logTipStates) { int rc = ch.in->setTipStates(ts.first, ts.second.data() + ch.start); if (rc) return rc; }
        for (auto& tp : logTipPartials) { int rc = ch.in->importPartials(tp.first, tp.second.data() + (size_t) ch.start * S, false); if (rc) return rc; }
        if (!logWeights.empty()) { int rc = ch.in->upload(ch.in->d_pweights, logWeights.data() + ch.start, sizeof(double) * ch.count); if (rc) return rc; }
    }
    return BEAGLE_SUCCESS;
}

}  // namespace mbamd

// =============================================================================================
// C ABI
// =============================================================================================
#include "libhmsbeagle/mbamd_parsimony.h"
#include "mbamd_parsimony.h"
#include "mbamd_f64.h"

using namespace mbamd;

#define GET_INSTANCE_RAW(id)                                                                         \
    Instance* in = lookup(id);                                                                       \
    if (!in) return fail(BEAGLE_ERROR_UNINITIALIZED_INSTANCE, "no such instance");                   \
    (void) hipSetDevice(in->device)
// (a beagleResetScaleFactors that is still waiting for its beagleAccumulateScaleFactors -- see there -- runs before anything else)
#define GET_INSTANCE_NOFLUSH(id)                                                                     \
    GET_INSTANCE_RAW(id);                                                                            \
    if (in->deferredReset >= 0) {                                                                    \
        int drc_ = in->runDeferredReset();                                                           \
        if (drc_ != BEAGLE_SUCCESS) return drc_;                                                     \
    }
// every entry point except beagleUpdatePartials first runs the lists deferred so far
#define GET_INSTANCE(id)                                                                             \
    GET_INSTANCE_NOFLUSH(id);                                                                        \
    if (in->hasPending() || !in->pendingJobs.empty()) {                                          \
        int frc_ = in->flushPending();                                                               \
        if (frc_ != BEAGLE_SUCCESS) return frc_;                                                     \
    }
// the calls between a list and its log-likelihood that do not touch partials (category weights, state frequencies) and the
// log-likelihood calls themselves leave a held 4-state path where it is (Instance::heldPath)
#define GET_INSTANCE_KEEPING_PATH(id)                                                                \
    GET_INSTANCE_NOFLUSH(id);                                                                        \
    if (!in->pending.empty() || !in->wgListCum.empty() || !in->pendingJobs.empty()) {                \
        int frc_ = in->flushPending(true);                                                           \
        if (frc_ != BEAGLE_SUCCESS) return frc_;                                                     \
    }
// a facade forwards the call to every child (`c`, its pattern range in `ch`) and returns
#define FACADE_EACH(FLUSH, ...)                                                                      \
    if (in->facade()) {                                                                              \
        for (Instance::Child& ch : in->children) {                                                   \
            Instance* c = ch.in;                                                                     \
            (void) ch;                                                                               \
            (void) hipSetDevice(c->device);                                                          \
            if (FLUSH && (c->hasPending() || !c->pendingJobs.empty())) {                         \
                int frc_ = c->flushPending();                                                        \
                if (frc_ != BEAGLE_SUCCESS) return frc_;                                             \
            }                                                                                        \
            int crc_ = (__VA_ARGS__);                                                                \
            if (crc_ != BEAGLE_SUCCESS) return crc_;                                                 \
        }                                                                                            \
        return BEAGLE_SUCCESS;                                                                       \
    }
#define FACADE_ALL(...) FACADE_EACH(true, __VA_ARGS__)

// the engine proper for one log-likelihood call; facade: per-child sums, FLOATING_POINT if any child says so
static int integrate_any(Instance* in, const int* parent, const int* child, const int* prob, const int* wIdx, const int* fIdx,
                         const int* cumIdx, int count, const int* partitionIndices, int partitionCount, double* outByPartition,
                         double* outSum)
{
    if (!in->facade()) return in->integrate(parent, child, prob, wIdx, fIdx, cumIdx, count, outSum);
    // arrays are laid out [count][partitionCount] when partitionIndices is given (reference src/mbbeagle.c:2781-2800)
    const int pc = partitionIndices ? partitionCount : 1;
    std::vector<int> pa(count), ca(count), pr(count), wa(count), fa(count), cu(count);
    std::vector<Instance*> launched;
    std::vector<int> slotOf;
    for (Instance::Child& ch : in->children) {
        int dpos = 0;
        if (partitionIndices) {
            dpos = -1;
            for (int d = 0; d < pc; ++d) if (partitionIndices[d] == ch.partition) dpos = d;
            if (dpos < 0) continue;                     // this partition is not part of the call
        }
        for (int i = 0; i < count; ++i) {
            const int j = i * pc + dpos;
            pa[i] = parent[j]; wa[i] = wIdx[j]; fa[i] = fIdx[j];
            cu[i] = cumIdx ? cumIdx[j] : BEAGLE_OP_NONE;
            if (child) { ca[i] = child[j]; pr[i] = prob[j]; }
        }
        Instance* c = ch.in;
        (void) hipSetDevice(c->device);
        if (c->hasPending() || !c->pendingJobs.empty()) { int frc = c->flushPending(); if (frc) return frc; }
        const bool was = c->deferred;
        c->deferred = true;                             // launch everywhere first, collect afterwards
        const int rc = c->integrate(pa.data(), child ? ca.data() : nullptr, child ? pr.data() : nullptr, wa.data(), fa.data(),
                                    cumIdx ? cu.data() : nullptr, count, nullptr);
        c->deferred = was;
        if (rc) return rc;
        launched.push_back(c);
        slotOf.push_back(dpos);
    }
    double total = 0.0;
    int result = BEAGLE_SUCCESS;
    if (outByPartition) for (int d = 0; d < pc; ++d) outByPartition[d] = 0.0;
    for (size_t i = 0; i < launched.size(); ++i) {
        (void) hipSetDevice(launched[i]->device);
        double v = 0.0;
        const int rc = launched[i]->fetchResult(&v);
        if (rc == BEAGLE_ERROR_FLOATING_POINT) result = rc;
        else if (rc) return rc;
        total += v;
        if (outByPartition) outByPartition[slotOf[i]] += v;
    }
    if (outSum) *outSum = total;
    in->haveSite = true;
    return result;
}

extern "C" {

const char* beagleGetVersion(void) { return "mbamd-0.2 (HIP/gfx950; BEAGLE API 3.x compatible subset)"; }
const char* beagleGetCitation(void)
{
    return "mrbayes_amd: MI355X-native conditional-likelihood engine behind the BEAGLE API used by MrBayes";
}
const char* mbamdGetLastError(void) { return g_last_error.c_str(); }

BeagleResourceList* beagleGetResourceList(void)
{
    std::lock_guard<std::mutex> lk(g_mutex);
    buildResources();
    return &g_resources;
}

// v3: "benchmark" every resource for a problem size and let the client pick the fastest (reference
// src/mbbeagle.c:220-307).  Every resource is an MI355X running the same kernels: nothing is timed, the list is the
// resource list in order with a nominal, equal result -- the client's "first fastest" rule then picks resource 0
// unless the user named one.
BeagleBenchmarkedResourceList* beagleGetBenchmarkedResourceList(int tipCount, int compactBufferCount, int stateCount, int patternCount,
                                                                int categoryCount, int* resourceList, int resourceCount,
                                                                long preferenceFlags, long requirementFlags, int eigenModelCount,
                                                                int partitionCount, int calculateDerivatives, long benchmarkFlags)
{
    (void) tipCount; (void) compactBufferCount; (void) stateCount; (void) patternCount; (void) categoryCount; (void) resourceList;
    (void) resourceCount; (void) preferenceFlags; (void) requirementFlags; (void) eigenModelCount; (void) partitionCount;
    (void) calculateDerivatives;
    static BeagleBenchmarkedResourceList list = {nullptr, 0};
    static std::vector<BeagleBenchmarkedResource> vec;
    std::lock_guard<std::mutex> lk(g_mutex);
    buildResources();
    vec.resize(std::max(1, g_resources.length));
    for (int i = 0; i < g_resources.length; ++i) {
        BeagleBenchmarkedResource& r = vec[i];
        r.number = i;
        r.name = g_resources.list[i].name;
        r.description = g_resources.list[i].description;
        r.supportFlags = g_resources.list[i].supportFlags;
        r.requiredFlags = 0;
        r.returnCode = BEAGLE_SUCCESS;
        r.implName = const_cast<char*>(MBAMD_IMPL_NAME);
        r.benchedFlags = kSupport | (benchmarkFlags & BEAGLE_BENCHFLAG_SCALING_ALWAYS ? BEAGLE_FLAG_SCALING_ALWAYS : BEAGLE_FLAG_SCALING_DYNAMIC);
        r.benchmarkResult = 1.0;
        r.performanceRatio = 1.0;
    }
    list.list = vec.data();
    list.length = g_resources.length;
    return list.length > 0 ? &list : nullptr;
}

int beagleCreateInstance(int tipCount, int partialsBufferCount, int compactBufferCount, int stateCount,
                         int patternCount, int eigenBufferCount, int matrixBufferCount, int categoryCount,
                         int scaleBufferCount, int* resourceList, int resourceCount, long preferenceFlags,
                         long requirementFlags, BeagleInstanceDetails* returnInfo)
{
    API_TRACE("beagleCreateInstance(tips=%d, partials=%d, compact=%d, states=%d, patterns=%d, eigen=%d, matrices=%d, categories=%d, scale=%d)",
              tipCount, partialsBufferCount, compactBufferCount, stateCount, patternCount, eigenBufferCount, matrixBufferCount,
              categoryCount, scaleBufferCount);
    if (tipCount < 0 || partialsBufferCount < 0 || compactBufferCount < 0 || stateCount < 2 || patternCount < 1 ||
        eigenBufferCount < 0 || matrixBufferCount < 0 || categoryCount < 1 || scaleBufferCount < 0)
        return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleCreateInstance: bad dimensions");
    if (stateCount > 64) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleCreateInstance: more than 64 states");
    if (categoryCount > MBAMD_MAX_RATES) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleCreateInstance: more than 16 rate categories");
    if (requirementFlags & (BEAGLE_FLAG_EIGEN_COMPLEX | BEAGLE_FLAG_PROCESSOR_CPU |
                            BEAGLE_FLAG_FRAMEWORK_CUDA | BEAGLE_FLAG_FRAMEWORK_OPENCL | BEAGLE_FLAG_SCALERS_RAW))
        return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleCreateInstance: unsupported requirement flags");
    if ((requirementFlags & BEAGLE_FLAG_PRECISION_DOUBLE) && (requirementFlags & BEAGLE_FLAG_PRECISION_SINGLE))
        return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleCreateInstance: both precisions required");
    // double precision when it is required, or preferred and single is neither required nor preferred as well (MrBayes puts
    // `set beagleprecision=` into the preference flags, reference src/mbbeagle.c:186; single is the faster engine)
    const bool wantDouble = (requirementFlags & BEAGLE_FLAG_PRECISION_DOUBLE) ||
                            ((preferenceFlags & BEAGLE_FLAG_PRECISION_DOUBLE) && !(preferenceFlags & BEAGLE_FLAG_PRECISION_SINGLE) &&
                             !(requirementFlags & BEAGLE_FLAG_PRECISION_SINGLE));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(BEAGLE_ERROR_NO_RESOURCE, "beagleCreateInstance: no HIP device (this engine has no CPU path)");
    std::vector<int> devices;                      // a resource list with several GPUs = shard the patterns over them
    if (resourceList && resourceCount > 0) {
        for (int i = 0; i < resourceCount; ++i)
            if (resourceList[i] >= 0 && resourceList[i] < ndev) devices.push_back(resourceList[i]);
        if (devices.empty()) return fail(BEAGLE_ERROR_NO_RESOURCE, "beagleCreateInstance: requested resource not available");
    } else {
        devices.push_back(0);
    }
    if (const char* e = std::getenv("MBAMD_SHARD")) {      // MrBayes names at most one resource: shard over g devices from there
        const int g = std::max(1, std::min(64, std::atoi(e)));
        const int first = devices[0];
        devices.clear();
        for (int i = 0; i < g; ++i) devices.push_back((first + i) % ndev);
    }
    const int dev = devices[0];
    Instance* in = new Instance();
    in->flags = kSupport | (requirementFlags & (BEAGLE_FLAG_SCALING_ALWAYS | BEAGLE_FLAG_SCALING_DYNAMIC));
    const int args[9] = {tipCount, partialsBufferCount, compactBufferCount, stateCount, patternCount, eigenBufferCount,
                         matrixBufferCount, categoryCount, scaleBufferCount};
    std::memcpy(in->createArgs, args, sizeof args);
    in->shardDevices = devices;
    int rc;
    if (wantDouble) {
        // the handle only: every entry point forwards to the fp64 engine (mbamd_f64.h); one device, no shards
        in->device = dev;
        in->S = stateCount; in->P = patternCount; in->K = categoryCount; in->nEigen = eigenBufferCount;
        in->released = true;
        in->flags = (in->flags & ~BEAGLE_FLAG_PRECISION_SINGLE) | BEAGLE_FLAG_PRECISION_DOUBLE;
        in->f64 = new Engine64();
        rc = in->f64->create(tipCount, partialsBufferCount, compactBufferCount, stateCount, patternCount, eigenBufferCount,
                             matrixBufferCount, categoryCount, scaleBufferCount, dev);
    } else if (devices.size() > 1 && patternCount > 64) {
        // facade from the start: dimensions only, the children own the device memory
        in->device = dev;
        in->tipCount = tipCount; in->nBuffers = partialsBufferCount + compactBufferCount; in->S = stateCount; in->P = patternCount;
        in->Ppad = round_up(patternCount, 64); in->K = categoryCount; in->nEigen = eigenBufferCount; in->nMatrices = matrixBufferCount;
        in->nScale = scaleBufferCount;
        in->released = true;
        rc = in->makeChildren(std::vector<std::pair<int, int>>(1, std::make_pair(0, patternCount)));
    } else {
        in->shardDevices.assign(1, dev);
        rc = in->create(tipCount, partialsBufferCount, compactBufferCount, stateCount, patternCount, eigenBufferCount,
                        matrixBufferCount, categoryCount, scaleBufferCount, dev);
        if (rc == BEAGLE_ERROR_OUT_OF_MEMORY && in->wg) {
            // the arenas of the 20/61-state tree walk did not fit: once more on the level kernels (buffers allocated on first use)
            const long fl = in->flags;
            in->destroy();
            delete in;
            (void) hipGetLastError();
            in = new Instance();
            in->flags = fl;
            in->noWalkG = true;
            std::memcpy(in->createArgs, args, sizeof args);
            in->shardDevices.assign(1, dev);
            rc = in->create(tipCount, partialsBufferCount, compactBufferCount, stateCount, patternCount, eigenBufferCount,
                            matrixBufferCount, categoryCount, scaleBufferCount, dev);
        }
    }
    if (rc != BEAGLE_SUCCESS) {
        in->destroyChildren();
        if (!in->released) in->destroy();
        delete in->f64;
        delete in;
        return rc;
    }
    int id;
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        buildResources();
        id = -1;
        for (size_t i = 0; i < g_instances.size(); ++i)
            if (!g_instances[i]) { id = (int) i; break; }
        if (id < 0) { id = (int) g_instances.size(); g_instances.push_back(nullptr); }
        g_instances[id] = in;
    }
    if (returnInfo) {
        const Instance* first = in->facade() ? in->children[0].in : in;
        returnInfo->resourceNumber = dev;
        returnInfo->resourceName = (dev < g_resources.length) ? g_resources.list[dev].name : const_cast<char*>("HIP device");
        returnInfo->implName = const_cast<char*>(first->f64 ? (first->S == 4 ? MBAMD_IMPL_NAME ": double-precision kernels (four states: tree walk)" : MBAMD_IMPL_NAME ": double-precision level kernels")
                                                 : first->s4 ? MBAMD_IMPL_NAME ": 4-state tree-walk kernels"
                                                 : first->wg ? (wg_bf16(first->S) ? MBAMD_IMPL_NAME ": general-state tree-walk kernels (fp32 arithmetic as three exact bf16 pieces on v_mfma_f32_32x32x16_bf16)"
                                                                                  : MBAMD_IMPL_NAME ": general-state tree-walk kernels (v_mfma_f32_32x32x2_f32)")
                                                 : first->mfma ? MBAMD_IMPL_NAME ": general-state MFMA (v_mfma_f32_32x32x2_f32) kernels"
                                                               : MBAMD_IMPL_NAME ": general-state vector kernels");
        returnInfo->implDescription = const_cast<char*>("hand-written HIP kernels for AMD CDNA4 (MI355X)");
        returnInfo->flags = in->flags;
    }
    return id;
}

int beagleFinalizeInstance(int instance)
{
    Instance* in;
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        if (instance < 0 || instance >= (int) g_instances.size() || !g_instances[instance])
            return fail(BEAGLE_ERROR_UNINITIALIZED_INSTANCE, "beagleFinalizeInstance: no such instance");
        in = g_instances[instance];
        g_instances[instance] = nullptr;
    }
    if (g_statsOn) {
        std::fprintf(stderr, "[mbamd] instance %d: plan cache %ld hits / %ld misses; tree-walk schedules re-used %llu / built %llu; root-ward paths held %ld, run with their log-likelihood as one launch %ld\n", instance,
                     in->planHits, in->planMisses, (unsigned long long) in->scheduleHits, (unsigned long long) in->scheduleMisses, in->heldPaths, in->fusedPaths);
        if (in->listsTotal)
            std::fprintf(stderr, "[mbamd] instance %d: 4-state lists %ld: root-ward paths %ld (of them forked %ld), tree walks %ld (%.1f operations each)\n", instance,
                         in->listsTotal, in->listsPath, in->forkedPaths, in->listsWalked, in->listsWalked ? (double) in->opsWalked / in->listsWalked : 0.0);
        for (const ApiStats& a : g_stats)
            std::fprintf(stderr, "[mbamd]   %-34s %9ld calls %10.3f ms total %9.2f us/call\n", a.name, a.calls,
                         a.seconds * 1e3, a.calls ? a.seconds * 1e6 / a.calls : 0.0);
    }
    in->destroyChildren();
    if (!in->released) in->destroy();
    delete in->f64;
    delete in;
    return BEAGLE_SUCCESS;
}

int beagleFinalize(void)
{
    std::vector<Instance*> all;
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        all.swap(g_instances);
    }
    for (Instance* in : all)
        if (in) { in->destroyChildren(); if (!in->released) in->destroy(); delete in->f64; delete in; }
    return BEAGLE_SUCCESS;
}

// v3: only meaningful for a CPU implementation (reference src/mbbeagle.c:384-387); its presence in the library is what
// makes MrBayes' configure compile the v3 code path (configure.ac:220-223)
int beagleSetCPUThreadCount(int instance, int threadCount)
{
    (void) threadCount;
    GET_INSTANCE_NOFLUSH(instance);
    if (in->f64) return BEAGLE_SUCCESS;
    return BEAGLE_SUCCESS;
}

// v3 multi-partition mode (reference src/mcmc.c:6464): patternPartitions[c] = partition of pattern c, partitions are
// contiguous pattern ranges (MrBayes lists its divisions one after the other).  From here on the instance is a facade over
// one child per partition (times the shards); tip data and pattern weights set so far are replayed into the children.
int beagleSetPatternPartitions(int instance, int partitionCount, const int* inPatternPartitions)
{
    GET_INSTANCE(instance);
    if (in->f64) return (partitionCount < 1 || !inPatternPartitions) ? fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetPatternPartitions: arguments") : in->f64->setPartitions(partitionCount, inPatternPartitions);
    API_TRACE("beagleSetPatternPartitions(%d partitions)", partitionCount);
    if (partitionCount < 1 || !inPatternPartitions) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetPatternPartitions: arguments");
    if (!in->logOpen) return fail(BEAGLE_ERROR_GENERAL, "beagleSetPatternPartitions: call it before the first matrix / partials update");
    std::vector<std::pair<int, int>> ranges;
    for (int c = 0; c < in->P; ++c) {
        const int p = inPatternPartitions[c];
        if (p < 0 || p >= partitionCount) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetPatternPartitions: partition index");
        if ((int) ranges.size() == p) ranges.emplace_back(c, 1);
        else if ((int) ranges.size() == p + 1 && ranges[p].first + ranges[p].second == c) ranges[p].second++;
        else return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleSetPatternPartitions: partitions must be contiguous, increasing pattern ranges");
    }
    if ((int) ranges.size() != partitionCount) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetPatternPartitions: empty partition");
    in->partitionCount = partitionCount;
    if (partitionCount == 1 && !in->facade()) return BEAGLE_SUCCESS;
    if (!in->released) {                          // hand the single-partition buffers back
        in->destroy();
        in->released = true;
    }
    return in->makeChildren(ranges);
}

int beagleSetTipStates(int instance, int tipIndex, const int* inStates)
{
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->setTipStates(tipIndex, inStates);
    API_TRACE("beagleSetTipStates(tip=%d, states=%s...)", tipIndex, trace_ints(inStates, std::min(8, in->P)).c_str());
    if (in->logOpen) in->logTipStates.emplace_back(tipIndex, std::vector<int>(inStates, inStates + in->P));
    FACADE_ALL(c->setTipStates(tipIndex, inStates + ch.start));
    return in->setTipStates(tipIndex, inStates);
}
int beagleSetTipPartials(int instance, int tipIndex, const double* inPartials)
{
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->setPartials(tipIndex, inPartials, false);
    API_TRACE("beagleSetTipPartials(tip=%d, %s...)", tipIndex, trace_doubles(inPartials, std::min(8, in->S)).c_str());
    if (in->logOpen) in->logTipPartials.emplace_back(tipIndex, std::vector<double>(inPartials, inPartials + (size_t) in->P * in->S));
    FACADE_ALL(c->importPartials(tipIndex, inPartials + (size_t) ch.start * in->S, false));
    return in->importPartials(tipIndex, inPartials, false);
}
int beagleSetPartials(int instance, int bufferIndex, const double* inPartials)
{
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->setPartials(bufferIndex, inPartials, true);
    if (in->facade()) {
        std::vector<double> part;
        for (Instance::Child& ch : in->children) {
            part.resize((size_t) in->K * ch.count * in->S);
            for (int k = 0; k < in->K; ++k)
                std::memcpy(part.data() + (size_t) k * ch.count * in->S, inPartials + ((size_t) k * in->P + ch.start) * in->S,
                            sizeof(double) * ch.count * in->S);
            (void) hipSetDevice(ch.in->device);
            const int rc = ch.in->importPartials(bufferIndex, part.data(), true);
            if (rc) return rc;
        }
        return BEAGLE_SUCCESS;
    }
    return in->importPartials(bufferIndex, inPartials, true);
}
int beagleGetPartials(int instance, int bufferIndex, int scaleIndex, double* outPartials)
{
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->getPartials(bufferIndex, outPartials);
    if (scaleIndex != BEAGLE_OP_NONE) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleGetPartials: scaleIndex must be BEAGLE_OP_NONE");
    if (in->facade()) {
        std::vector<double> part;
        for (Instance::Child& ch : in->children) {
            part.resize((size_t) in->K * ch.count * in->S);
            (void) hipSetDevice(ch.in->device);
            if (ch.in->hasPending() || !ch.in->pendingJobs.empty()) { int frc = ch.in->flushPending(); if (frc) return frc; }
            const int rc = ch.in->getPartials(bufferIndex, part.data());
            if (rc) return rc;
            for (int k = 0; k < in->K; ++k)
                std::memcpy(outPartials + ((size_t) k * in->P + ch.start) * in->S, part.data() + (size_t) k * ch.count * in->S,
                            sizeof(double) * ch.count * in->S);
        }
        return BEAGLE_SUCCESS;
    }
    return in->getPartials(bufferIndex, outPartials);
}
int beagleSetEigenDecomposition(int instance, int eigenIndex, const double* inEigenVectors,
                                const double* inInverseEigenVectors, const double* inEigenValues)
{
    StatTimer st_(ST_SET);
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->setEigen(eigenIndex, inEigenVectors, inInverseEigenVectors, inEigenValues);
    API_TRACE("beagleSetEigenDecomposition(eigen=%d, values=%s...)", eigenIndex, trace_doubles(inEigenValues, std::min(6, in->S)).c_str());
    FACADE_ALL(c->setEigen(eigenIndex, inEigenVectors, inInverseEigenVectors, inEigenValues));
    return in->setEigen(eigenIndex, inEigenVectors, inInverseEigenVectors, inEigenValues);
}
// extension: eigen-systems of `count` reversible rate matrices computed on the device and stored in the eigen
// buffers firstEigenIndex ...; q: count x S x S row-major rates (mode 0) or exchangeabilities (mode 1: Q is built and
// normalised on the device too); pi: S state frequencies, all positive.  Replaces beagleSetEigenDecomposition + the host's
// eigen-solver for these models.
int mbamdSetRateMatrices(int instance, int firstEigenIndex, int count, const double* q, const double* pi, int mode)
{
    GET_INSTANCE(instance);
    if (in->f64) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdSetRateMatrices: not on a double-precision instance");
    if (!q || !pi || (mode != 0 && mode != 1)) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdSetRateMatrices: arguments");
    FACADE_ALL(c->setRateMatrices(firstEigenIndex, count, q, pi, mode));
    return in->setRateMatrices(firstEigenIndex, count, q, pi, mode);
}
int mbamdSetRateMatricesFrom(int instance, int firstEigenIndex, int count, const double* q, const double* pi, int mode, int warmFirstEigenIndex)
{
    GET_INSTANCE(instance);
    if (in->f64) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdSetRateMatricesFrom: not on a double-precision instance");
    if (!q || !pi) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdSetRateMatricesFrom: null");
    FACADE_ALL(c->setRateMatrices(firstEigenIndex, count, q, pi, mode, warmFirstEigenIndex));
    return in->setRateMatrices(firstEigenIndex, count, q, pi, mode, warmFirstEigenIndex);
}
int beagleSetStateFrequencies(int instance, int idx, const double* f)
{
    StatTimer st_(ST_SET);
    GET_INSTANCE_KEEPING_PATH(instance);
    if (in->f64) return in->f64->setFreqs(idx, f);
    API_TRACE("beagleSetStateFrequencies(%d, %s...)", idx, trace_doubles(f, std::min(6, in->S)).c_str());
    if (idx < 0 || idx >= in->nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetStateFrequencies: index");
    FACADE_ALL(c->uploadIfChanged(c->h_freqs, (size_t) idx * c->S, c->d_freqs, f, c->S));
    return in->uploadIfChanged(in->h_freqs, (size_t) idx * in->S, in->d_freqs, f, in->S);
}
int beagleSetCategoryWeights(int instance, int idx, const double* w)
{
    StatTimer st_(ST_SET);
    GET_INSTANCE_KEEPING_PATH(instance);
    if (in->f64) return in->f64->setWeights(idx, w);
    API_TRACE("beagleSetCategoryWeights(%d, %s)", idx, trace_doubles(w, in->K).c_str());
    if (idx < 0 || idx >= in->nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetCategoryWeights: index");
    FACADE_ALL(c->uploadIfChanged(c->h_weights, (size_t) idx * c->K, c->d_weights, w, c->K));
    return in->uploadIfChanged(in->h_weights, (size_t) idx * in->K, in->d_weights, w, in->K);
}
int beagleSetCategoryRates(int instance, const double* r)
{
    StatTimer st_(ST_SET);
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->setRates(0, r);
    API_TRACE("beagleSetCategoryRates(%s)", trace_doubles(r, in->K).c_str());
    FACADE_ALL(c->setRates(0, r));
    return in->setRates(0, r);
}
// v3 (reference src/mbbeagle.c:2055): one rate vector per partition, named by index in beagleUpdateTransitionMatricesWithMultipleModels
int beagleSetCategoryRatesWithIndex(int instance, int categoryRatesIndex, const double* r)
{
    StatTimer st_(ST_SET);
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->setRates(categoryRatesIndex, r);
    API_TRACE("beagleSetCategoryRatesWithIndex(%d, %s)", categoryRatesIndex, trace_doubles(r, in->K).c_str());
    FACADE_ALL(c->setRates(categoryRatesIndex, r));
    return in->setRates(categoryRatesIndex, r);
}
int beagleSetPatternWeights(int instance, const double* w)
{
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->setPatternWeights(w);
    if (in->logOpen) in->logWeights.assign(w, w + in->P);
    FACADE_ALL(c->upload(c->d_pweights, w + ch.start, sizeof(double) * ch.count));
    return in->upload(in->d_pweights, w, sizeof(double) * in->P);
}
int beagleUpdateTransitionMatrices(int instance, int eigenIndex, const int* probabilityIndices,
                                   const int* firstDerivativeIndices, const int* secondDerivativeIndices,
                                   const double* edgeLengths, int count)
{
    StatTimer st_(ST_MATRICES);
    GET_INSTANCE_NOFLUSH(instance);
    if (in->f64) return in->f64->updateMatrices(eigenIndex, 0, probabilityIndices, edgeLengths, count);
    API_TRACE("beagleUpdateTransitionMatrices(eigen=%d, count=%d, indices=%s..., lengths=%s...)", eigenIndex, count,
              trace_ints(probabilityIndices, std::min(6, count)).c_str(), trace_doubles(edgeLengths, std::min(6, count)).c_str());
    if (firstDerivativeIndices || secondDerivativeIndices)
        return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleUpdateTransitionMatrices: derivatives");
    in->closeLog();
    FACADE_EACH(false, (!c->hasPending() ? BEAGLE_SUCCESS : c->flushPending()) != BEAGLE_SUCCESS
                           ? BEAGLE_ERROR_GENERAL : c->updateMatrices(eigenIndex, probabilityIndices, edgeLengths, count));
    if (in->hasPending()) {                      // deferred lists read the matrices about to be replaced
        int frc_ = in->flushPending();
        if (frc_ != BEAGLE_SUCCESS) return frc_;
    }
    return in->updateMatrices(eigenIndex, probabilityIndices, edgeLengths, count);
}
// v3 (reference src/mbbeagle.c:2140-2147): every matrix names its own eigen-system and category-rate vector
int beagleUpdateTransitionMatricesWithMultipleModels(int instance, const int* eigenIndices, const int* categoryRateIndices,
                                                     const int* probabilityIndices, const int* firstDerivativeIndices,
                                                     const int* secondDerivativeIndices, const double* edgeLengths, int count)
{
    StatTimer st_(ST_MATRICES);
    GET_INSTANCE_NOFLUSH(instance);
    if (in->f64) return (firstDerivativeIndices || secondDerivativeIndices) ? fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleUpdateTransitionMatricesWithMultipleModels: derivatives") : in->f64->updateMatricesMulti(eigenIndices, categoryRateIndices, probabilityIndices, edgeLengths, count);
    API_TRACE("beagleUpdateTransitionMatricesWithMultipleModels(count=%d, eigen=%s..., rates=%s...)", count,
              trace_ints(eigenIndices, std::min(6, count)).c_str(), trace_ints(categoryRateIndices, std::min(6, count)).c_str());
    if (firstDerivativeIndices || secondDerivativeIndices)
        return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleUpdateTransitionMatricesWithMultipleModels: derivatives");
    in->closeLog();
    auto run = [&](Instance* c) {
        if (c->hasPending()) { int frc = c->flushPending(); if (frc) return frc; }
        int i = 0;
        while (i < count) {                      // runs of equal (eigen-system, rate vector)
            int j = i + 1;
            while (j < count && eigenIndices[j] == eigenIndices[i] && categoryRateIndices[j] == categoryRateIndices[i]) ++j;
            const int rc = c->updateMatrices(eigenIndices[i], probabilityIndices + i, edgeLengths + i, j - i, categoryRateIndices[i]);
            if (rc) return rc;
            i = j;
        }
        return (int) BEAGLE_SUCCESS;
    };
    FACADE_EACH(false, run(c));
    return run(in);
}
int beagleSetTransitionMatrix(int instance, int matrixIndex, const double* inMatrix, double paddedValue)
{
    (void) paddedValue;
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->setMatrix(matrixIndex, inMatrix);
    FACADE_ALL(c->setMatrix(matrixIndex, inMatrix));
    return in->setMatrix(matrixIndex, inMatrix);
}
int beagleGetTransitionMatrix(int instance, int matrixIndex, double* outMatrix)
{
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->getMatrix(matrixIndex, outMatrix);
    if (in->facade()) {
        Instance* c = in->children[0].in;
        (void) hipSetDevice(c->device);
        if (c->hasPending() || !c->pendingJobs.empty()) { int frc = c->flushPending(); if (frc) return frc; }
        return c->getMatrix(matrixIndex, outMatrix);
    }
    return in->getMatrix(matrixIndex, outMatrix);
}
int beagleUpdatePartials(int instance, const BeagleOperation* operations, int operationCount, int cumulativeScaleIndex)
{
    StatTimer st_(ST_PARTIALS);
    GET_INSTANCE_NOFLUSH(instance);
    if (in->f64) return in->f64->updatePartials(operations, operationCount, cumulativeScaleIndex);
    API_TRACE("beagleUpdatePartials(count=%d, cumulative=%d, first=%s, last=%s)", operationCount, cumulativeScaleIndex,
              trace_ints(reinterpret_cast<const int*>(operations), operationCount > 0 ? 7 : 0).c_str(),
              trace_ints(reinterpret_cast<const int*>(operations + std::max(0, operationCount - 1)), operationCount > 0 ? 7 : 0).c_str());
    in->closeLog();
    FACADE_EACH(false, c->updatePartials(operations, operationCount, cumulativeScaleIndex));
    return in->updatePartials(operations, operationCount, cumulativeScaleIndex);
}
// v3 (reference src/mbbeagle.c:2292, 2440, 2616): operations of several partitions in one array; every operation names its
// partition and its cumulative scale buffer.  Each partition's operations, in order, are one list for that partition's child.
int beagleUpdatePartialsByPartition(int instance, const BeagleOperationByPartition* operations, int operationCount)
{
    StatTimer st_(ST_PARTIALS);
    GET_INSTANCE_NOFLUSH(instance);
    if (in->f64) {
        std::vector<int> part((size_t) std::max(operationCount, 0)), cum((size_t) std::max(operationCount, 0));
        for (int i = 0; i < operationCount; ++i) { part[i] = operations[i].partition; cum[i] = operations[i].cumulativeScaleIndex; }
        return in->f64->updatePartialsEx(operations, sizeof(BeagleOperationByPartition), operationCount, part.data(), cum.data());
    }
    API_TRACE("beagleUpdatePartialsByPartition(count=%d)", operationCount);
    in->closeLog();
    std::vector<BeagleOperation> list;
    auto runFor = [&](Instance* c, int partition) {
        int i = 0;
        while (i < operationCount) {
            while (i < operationCount && partition >= 0 && operations[i].partition != partition) ++i;
            if (i >= operationCount) break;
            const int cum = operations[i].cumulativeScaleIndex;
            list.clear();
            while (i < operationCount && operations[i].cumulativeScaleIndex == cum) {
                if (partition < 0 || operations[i].partition == partition) {
                    BeagleOperation o;
                    std::memcpy(&o, &operations[i], sizeof o);       // the first seven ints are the plain operation
                    list.push_back(o);
                }
                ++i;
            }
            if (!list.empty()) { const int rc = c->updatePartials(list.data(), (int) list.size(), cum); if (rc) return rc; }
        }
        return (int) BEAGLE_SUCCESS;
    };
    FACADE_EACH(false, runFor(c, in->partitionCount > 1 ? ch.partition : -1));
    for (int i = 0; i < operationCount; ++i)
        if (operations[i].partition != 0) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartialsByPartition: partition index (no partitions were set)");
    return runFor(in, -1);
}
int beagleWaitForPartials(int instance, const int* destinationPartials, int destinationPartialsCount)
{
    (void) destinationPartials; (void) destinationPartialsCount;
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->synchronize();
    FACADE_ALL(hipStreamSynchronize(c->stream) == hipSuccess ? BEAGLE_SUCCESS : BEAGLE_ERROR_GENERAL);
    HIP_TRY(hipStreamSynchronize(in->stream));
    return BEAGLE_SUCCESS;
}

// scale-factor bookkeeping; `partition` < 0: all patterns
static int scale_accumulate(Instance* in, const int* scaleIndices, int count, int cumulativeScaleIndex, int sign, int partition)
{
    // queued 20/61-state lists run first unless the call commutes with them (MrBayes removes the old node factors of
    // eigen-system part j+1 between the lists of parts j and j+1: flushing there would undo the merge of the parts)
    auto one = [&](Instance* c) -> int {
        (void) hipSetDevice(c->device);
        if (c->hasPending() && !(c->wg && c->scaleOpsIndependentOfPending(scaleIndices, count, cumulativeScaleIndex))) {
            int frc = c->flushPending();
            if (frc != BEAGLE_SUCCESS) return frc;
        }
        return c->arena() ? c->accumulate4(scaleIndices, count, cumulativeScaleIndex, sign)
                          : c->accumulate(scaleIndices, count, cumulativeScaleIndex, sign);
    };
    if (in->facade()) {
        for (Instance::Child& ch : in->children) {
            if (partition >= 0 && in->partitionCount > 1 && ch.partition != partition) continue;
            const int rc = one(ch.in);
            if (rc != BEAGLE_SUCCESS) return rc;
        }
        return BEAGLE_SUCCESS;
    }
    return one(in);
}
static int scale_reset(Instance* in, int idx)
{
    if (idx < 0 || idx >= in->nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleResetScaleFactors: index");
    if (in->arena()) {
        // MrBayes resets every scale buffer once at start-up (reference src/mcmc.c:6270): nothing is allocated or
        // launched for a buffer until it is used -- "never written" reads as zero everywhere
        if (in->scaleState[idx] == 1) {           // node exponents in the arena: later reads must see zeros
            MBAMD_LAUNCH(k_exp_copy, (unsigned) (((size_t) in->K * in->Ppad + 255) / 256), 256, 0, in->stream, in->arenaExp, in->estride,
                         -1, idx, in->K, in->Ppad);
            HIP_TRY(hipGetLastError());
        }
        in->scaleState[idx] = 0;
        return BEAGLE_SUCCESS;
    }
    if (!in->scale[idx]) return in->ensureScale(idx);   // allocated zeroed
    MBAMD_LAUNCH(k_scale_copy, (unsigned) ((in->Ppad + 255) / 256), 256, 0, in->stream, (const int32_t*) nullptr, in->Ppad, in->scale[idx]);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}
static int scale_copy(Instance* in, int dst, int src)
{
    if (dst < 0 || dst >= in->nScale || src < 0 || src >= in->nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleCopyScaleFactors: index");
    if (in->arena()) {
        const int st = in->scaleState[src];
        if (st == 2) {
            in->scaleState[dst] = 0;
            int rc4 = in->ensureWide(dst);
            if (rc4) return rc4;
            HIP_TRY(hipMemcpyAsync(in->wideScale[dst], in->wideScale[src], (size_t) in->K * in->Ppad * sizeof(int32_t),
                                   hipMemcpyDeviceToDevice, in->stream));
        } else if (st == 1) {
            MBAMD_LAUNCH(k_exp_copy, (unsigned) (((size_t) in->K * in->Ppad + 255) / 256), 256, 0, in->stream, in->arenaExp, in->estride,
                         src, dst, in->K, in->Ppad);
            HIP_TRY(hipGetLastError());
            in->scaleState[dst] = 1;
        } else {
            return scale_reset(in, dst);
        }
        return BEAGLE_SUCCESS;
    }
    int rc = in->ensureScale(dst);
    if (rc) return rc;
    rc = in->ensureScale(src);
    if (rc) return rc;
    MBAMD_LAUNCH(k_scale_copy, (unsigned) ((in->Ppad + 255) / 256), 256, 0, in->stream, (const int32_t*) in->scale[src], in->Ppad, in->scale[dst]);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}
int beagleAccumulateScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex)
{
    StatTimer st_(ST_SCALE);
    GET_INSTANCE_RAW(instance);
    // Reset + Accumulate of one cumulative buffer, back to back (rescaling the MrBayes way, reference src/mbbeagle.c:1080-1098): the
    // reset was not launched -- this launch stores instead of adding
    if (in->deferredReset >= 0) {
        bool fuse = in->deferredReset == cumulativeScaleIndex && count > 0;
        for (int i = 0; fuse && i < count; ++i)
            if (scaleIndices[i] == cumulativeScaleIndex) fuse = false;     // (the buffer among its own sources: reset-then-add, not a store)
        if (fuse) {
            if (in->hasPending()) { int frc = in->flushPending(); if (frc != BEAGLE_SUCCESS) return frc; }
            const int arc = in->accumulate(scaleIndices, count, cumulativeScaleIndex, +1, true);
            if (arc == BEAGLE_SUCCESS) { in->deferredReset = -1; return arc; }
            // the store did not happen (an index out of range, ...): the reset the caller asked for still does, then the error is theirs
            const int drc = in->runDeferredReset();
            return drc != BEAGLE_SUCCESS ? drc : arc;
        }
        int drc = in->runDeferredReset();
        if (drc != BEAGLE_SUCCESS) return drc;
    }
    if (in->f64) return in->f64->accumulateScale(scaleIndices, count, cumulativeScaleIndex, +1);
    return scale_accumulate(in, scaleIndices, count, cumulativeScaleIndex, +1, -1);
}
int beagleRemoveScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex)
{
    StatTimer st_(ST_SCALE);
    GET_INSTANCE_NOFLUSH(instance);
    if (in->f64) return in->f64->accumulateScale(scaleIndices, count, cumulativeScaleIndex, -1);
    return scale_accumulate(in, scaleIndices, count, cumulativeScaleIndex, -1, -1);
}
int beagleResetScaleFactors(int instance, int cumulativeScaleIndex)
{
    StatTimer st_(ST_SCALE);
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->resetScale(cumulativeScaleIndex);
    FACADE_ALL(scale_reset(c, cumulativeScaleIndex));
    // the level-kernel path (int32 buffers): wait for the call that follows -- if it is beagleAccumulateScaleFactors into this buffer,
    // one launch does both (any other entry point runs the reset first: GET_INSTANCE_NOFLUSH)
    if (!in->arena() && cumulativeScaleIndex >= 0 && cumulativeScaleIndex < in->nScale && in->scale[cumulativeScaleIndex]) {
        in->deferredReset = cumulativeScaleIndex;
        return BEAGLE_SUCCESS;
    }
    return scale_reset(in, cumulativeScaleIndex);
}
int beagleCopyScaleFactors(int instance, int destScalingIndex, int srcScalingIndex)
{
    StatTimer st_(ST_SCALE);
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->copyScale(destScalingIndex, srcScalingIndex);
    FACADE_ALL(scale_copy(c, destScalingIndex, srcScalingIndex));
    return scale_copy(in, destScalingIndex, srcScalingIndex);
}
// v3 (reference src/likelihood.c:8096-8103, src/mbbeagle.c:2566): the same on one partition's patterns
int beagleAccumulateScaleFactorsByPartition(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex, int partitionIndex)
{
    StatTimer st_(ST_SCALE);
    GET_INSTANCE_NOFLUSH(instance);
    if (in->f64) return in->f64->accumulateScale(scaleIndices, count, cumulativeScaleIndex, +1, partitionIndex);
    return scale_accumulate(in, scaleIndices, count, cumulativeScaleIndex, +1, partitionIndex);
}
int beagleRemoveScaleFactorsByPartition(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex, int partitionIndex)
{
    StatTimer st_(ST_SCALE);
    GET_INSTANCE_NOFLUSH(instance);
    if (in->f64) return in->f64->accumulateScale(scaleIndices, count, cumulativeScaleIndex, -1, partitionIndex);
    return scale_accumulate(in, scaleIndices, count, cumulativeScaleIndex, -1, partitionIndex);
}
int beagleResetScaleFactorsByPartition(int instance, int cumulativeScaleIndex, int partitionIndex)
{
    StatTimer st_(ST_SCALE);
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->resetScale(cumulativeScaleIndex, partitionIndex);
    FACADE_ALL((in->partitionCount > 1 && ch.partition != partitionIndex) ? BEAGLE_SUCCESS : scale_reset(c, cumulativeScaleIndex));
    return scale_reset(in, cumulativeScaleIndex);
}
// engine extension: the binary exponents behind a scale buffer, out[k * patternCount + c] (the general-state
// path keeps one exponent per pattern: every category row is the same)
int mbamdGetScaleExponents(int instance, int srcScalingIndex, int* out)
{
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->getScaleExponents(srcScalingIndex, out);
    if (in->facade()) {
        std::vector<int> part;
        for (Instance::Child& ch : in->children) {
            part.resize((size_t) in->K * ch.count);
            (void) hipSetDevice(ch.in->device);
            if (ch.in->hasPending() || !ch.in->pendingJobs.empty()) { int frc = ch.in->flushPending(); if (frc) return frc; }
            const int rc = ch.in->getScaleExponents(srcScalingIndex, part.data());
            if (rc) return rc;
            for (int k = 0; k < in->K; ++k)
                std::memcpy(out + (size_t) k * in->P + ch.start, part.data() + (size_t) k * ch.count, sizeof(int) * ch.count);
        }
        return BEAGLE_SUCCESS;
    }
    return in->getScaleExponents(srcScalingIndex, out);
}
// BEAGLE's scale factors are one log value per pattern.  The 4-state path keeps an exponent per (pattern, category):
// reported here is the largest of a pattern's exponents (times ln 2), the factor a per-pattern scaler would have used.
int beagleGetScaleFactors(int instance, int srcScalingIndex, double* outScaleFactors)
{
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->getScaleFactors(srcScalingIndex, outScaleFactors);
    if (srcScalingIndex < 0 || srcScalingIndex >= in->nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleGetScaleFactors: index");
    std::vector<int> e((size_t) in->K * in->P);
    int rc = mbamdGetScaleExponents(instance, srcScalingIndex, e.data());
    if (rc) return rc;
    for (int c = 0; c < in->P; ++c) {
        int m = e[c];
        for (int k = 1; k < in->K; ++k) m = std::max(m, e[(size_t) k * in->P + c]);
        outScaleFactors[c] = (double) m * 0.69314718055994530942;
    }
    return BEAGLE_SUCCESS;
}
int beagleCalculateRootLogLikelihoods(int instance, const int* bufferIndices, const int* categoryWeightsIndices,
                                      const int* stateFrequenciesIndices, const int* cumulativeScaleIndices, int count,
                                      double* outSumLogLikelihood)
{
    StatTimer st_(ST_LNL);
    GET_INSTANCE_KEEPING_PATH(instance);
    if (in->f64) return in->f64->logLikelihoods(bufferIndices, nullptr, nullptr, categoryWeightsIndices, stateFrequenciesIndices, cumulativeScaleIndices, count, outSumLogLikelihood);
    const int rc_ = integrate_any(in, bufferIndices, nullptr, nullptr, categoryWeightsIndices, stateFrequenciesIndices,
                                  cumulativeScaleIndices, count, nullptr, 1, nullptr, outSumLogLikelihood);
    API_TRACE("beagleCalculateRootLogLikelihoods(buffers=%s, weights=%s, freqs=%s, cumulative=%s) -> %d, lnL %.6f",
              trace_ints(bufferIndices, count).c_str(), trace_ints(categoryWeightsIndices, count).c_str(),
              trace_ints(stateFrequenciesIndices, count).c_str(), trace_ints(cumulativeScaleIndices, count).c_str(), rc_,
              outSumLogLikelihood ? *outSumLogLikelihood : 0.0);
    return rc_;
}
int beagleCalculateEdgeLogLikelihoods(int instance, const int* parentBufferIndices, const int* childBufferIndices,
                                      const int* probabilityIndices, const int* firstDerivativeIndices,
                                      const int* secondDerivativeIndices, const int* categoryWeightsIndices,
                                      const int* stateFrequenciesIndices, const int* cumulativeScaleIndices, int count,
                                      double* outSumLogLikelihood, double* outSumFirstDerivative,
                                      double* outSumSecondDerivative)
{
    StatTimer st_(ST_LNL);
    GET_INSTANCE_KEEPING_PATH(instance);
    if (in->f64) return (firstDerivativeIndices || secondDerivativeIndices || outSumFirstDerivative || outSumSecondDerivative) ? fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleCalculateEdgeLogLikelihoods: derivatives") : in->f64->logLikelihoods(parentBufferIndices, childBufferIndices, probabilityIndices, categoryWeightsIndices, stateFrequenciesIndices, cumulativeScaleIndices, count, outSumLogLikelihood);
    if (firstDerivativeIndices || secondDerivativeIndices || outSumFirstDerivative || outSumSecondDerivative)
        return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleCalculateEdgeLogLikelihoods: derivatives");
    const int rc_ = integrate_any(in, parentBufferIndices, childBufferIndices, probabilityIndices, categoryWeightsIndices,
                                  stateFrequenciesIndices, cumulativeScaleIndices, count, nullptr, 1, nullptr, outSumLogLikelihood);
    API_TRACE("beagleCalculateEdgeLogLikelihoods(parents=%s, children=%s, matrices=%s, weights=%s, freqs=%s, cumulative=%s) -> %d, lnL %.6f",
              trace_ints(parentBufferIndices, count).c_str(), trace_ints(childBufferIndices, count).c_str(),
              trace_ints(probabilityIndices, count).c_str(), trace_ints(categoryWeightsIndices, count).c_str(),
              trace_ints(stateFrequenciesIndices, count).c_str(), trace_ints(cumulativeScaleIndices, count).c_str(), rc_,
              outSumLogLikelihood ? *outSumLogLikelihood : 0.0);
    return rc_;
}
// v3 (reference src/mbbeagle.c:2817-2850): index arrays are [count][partitionCount]; one sum per named partition + the total
int beagleCalculateRootLogLikelihoodsByPartition(int instance, const int* bufferIndices, const int* categoryWeightsIndices,
                                                 const int* stateFrequenciesIndices, const int* cumulativeScaleIndices,
                                                 const int* partitionIndices, int partitionCount, int count,
                                                 double* outSumLogLikelihoodByPartition, double* outSumLogLikelihood)
{
    StatTimer st_(ST_LNL);
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->logLikelihoods(bufferIndices, nullptr, nullptr, categoryWeightsIndices, stateFrequenciesIndices, cumulativeScaleIndices, count, outSumLogLikelihood, partitionIndices, partitionCount, outSumLogLikelihoodByPartition);
    if (!in->facade() && (partitionCount != 1 || partitionIndices[0] != 0))
        return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleCalculateRootLogLikelihoodsByPartition: no partitions were set");
    double total = 0.0;
    const int rc_ = integrate_any(in, bufferIndices, nullptr, nullptr, categoryWeightsIndices, stateFrequenciesIndices,
                                  cumulativeScaleIndices, count, partitionIndices, partitionCount, outSumLogLikelihoodByPartition, &total);
    if (!in->facade() && outSumLogLikelihoodByPartition) outSumLogLikelihoodByPartition[0] = total;
    if (outSumLogLikelihood) *outSumLogLikelihood = total;
    API_TRACE("beagleCalculateRootLogLikelihoodsByPartition(%d partitions) -> %d, lnL %.6f", partitionCount, rc_, total);
    return rc_;
}
int beagleCalculateEdgeLogLikelihoodsByPartition(int instance, const int* parentBufferIndices, const int* childBufferIndices,
                                                 const int* probabilityIndices, const int* firstDerivativeIndices,
                                                 const int* secondDerivativeIndices, const int* categoryWeightsIndices,
                                                 const int* stateFrequenciesIndices, const int* cumulativeScaleIndices,
                                                 const int* partitionIndices, int partitionCount, int count,
                                                 double* outSumLogLikelihoodByPartition, double* outSumLogLikelihood,
                                                 double* outSumFirstDerivativeByPartition, double* outSumFirstDerivative,
                                                 double* outSumSecondDerivativeByPartition, double* outSumSecondDerivative)
{
    StatTimer st_(ST_LNL);
    GET_INSTANCE(instance);
    if (in->f64)
        return (firstDerivativeIndices || secondDerivativeIndices || outSumFirstDerivativeByPartition || outSumFirstDerivative || outSumSecondDerivativeByPartition || outSumSecondDerivative)
                   ? fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleCalculateEdgeLogLikelihoodsByPartition: derivatives")
                   : in->f64->logLikelihoods(parentBufferIndices, childBufferIndices, probabilityIndices, categoryWeightsIndices, stateFrequenciesIndices,
                                             cumulativeScaleIndices, count, outSumLogLikelihood, partitionIndices, partitionCount, outSumLogLikelihoodByPartition);
    if (firstDerivativeIndices || secondDerivativeIndices || outSumFirstDerivativeByPartition || outSumFirstDerivative ||
        outSumSecondDerivativeByPartition || outSumSecondDerivative)
        return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleCalculateEdgeLogLikelihoodsByPartition: derivatives");
    if (!in->facade() && (partitionCount != 1 || partitionIndices[0] != 0))
        return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleCalculateEdgeLogLikelihoodsByPartition: no partitions were set");
    double total = 0.0;
    const int rc_ = integrate_any(in, parentBufferIndices, childBufferIndices, probabilityIndices, categoryWeightsIndices,
                                  stateFrequenciesIndices, cumulativeScaleIndices, count, partitionIndices, partitionCount,
                                  outSumLogLikelihoodByPartition, &total);
    if (!in->facade() && outSumLogLikelihoodByPartition) outSumLogLikelihoodByPartition[0] = total;
    if (outSumLogLikelihood) *outSumLogLikelihood = total;
    API_TRACE("beagleCalculateEdgeLogLikelihoodsByPartition(%d partitions) -> %d, lnL %.6f", partitionCount, rc_, total);
    return rc_;
}
int beagleGetSiteLogLikelihoods(int instance, double* outLogLikelihoods)
{
    StatTimer st_(ST_SITE);
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->getSites(outLogLikelihoods);
    FACADE_ALL(c->haveSite ? c->getSites(outLogLikelihoods + ch.start) : BEAGLE_SUCCESS);
    return in->getSites(outLogLikelihoods);
}

// ---- engine extensions ---------------------------------------------------------------------
int mbamdSynchronize(int instance)
{
    GET_INSTANCE(instance);
    if (in->f64) return in->f64->synchronize();
    FACADE_ALL(hipStreamSynchronize(c->stream) == hipSuccess ? BEAGLE_SUCCESS : BEAGLE_ERROR_GENERAL);
    HIP_TRY(hipStreamSynchronize(in->stream));
    return BEAGLE_SUCCESS;
}
int mbamdKernelTiming(int instance, int enable)
{
    GET_INSTANCE(instance);
    if (in->f64) return BEAGLE_SUCCESS;
    if (in->facade()) { for (Instance::Child& ch : in->children) ch.in->timing = enable != 0; return BEAGLE_SUCCESS; }
    in->timing = enable != 0;
    return BEAGLE_SUCCESS;
}
static int kernel_timing_of(Instance* in, double* ms, long* launches, int reset)
{
    (void) hipSetDevice(in->device);
    HIP_TRY(hipStreamSynchronize(in->stream));
    for (auto& ev : in->events) {
        float t = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t, ev.first, ev.second));
        in->timedMs += t;
        (void) hipEventDestroy(ev.first);
        (void) hipEventDestroy(ev.second);
    }
    in->events.clear();
    in->timedLaunches += in->pendingLaunches;
    in->pendingLaunches = 0;
    *ms += in->timedMs;
    *launches += in->timedLaunches;
    if (reset) { in->timedMs = 0.0; in->timedLaunches = 0; }
    return BEAGLE_SUCCESS;
}
static int step_timing_of(Instance* in, double* ms, long* steps, int reset)
{
    (void) hipSetDevice(in->device);
    HIP_TRY(hipStreamSynchronize(in->stream));
    in->spanFold();
    *ms += in->spanMs;
    *steps += in->spanCount;
    if (reset) { in->spanMs = 0.0; in->spanCount = 0; }
    return BEAGLE_SUCCESS;
}
// (a facade reports the LARGEST kernel time of its children -- they run side by side -- and the sum of the launches)
int mbamdGetKernelTiming(int instance, double* outMilliseconds, long* outLaunches, int reset)
{
    GET_INSTANCE(instance);
    if (in->f64) {                               // (no device timing on a double-precision instance: the partials launches are counted)
        { const int rcq = in->f64->flushQueue(); if (rcq) return rcq; }
        if (outMilliseconds) *outMilliseconds = 0.0;
        if (outLaunches) *outLaunches = (long) (in->f64->walkLaunches + in->f64->levelLaunches);
        if (reset) in->f64->walkLaunches = in->f64->levelLaunches = 0;
        return BEAGLE_SUCCESS;
    }
    double ms = 0.0;
    long launches = 0;
    if (in->facade()) {
        for (Instance::Child& ch : in->children) {
            double m = 0.0;
            int rc = kernel_timing_of(ch.in, &m, &launches, reset);
            if (rc) return rc;
            ms = std::max(ms, m);
        }
    } else {
        int rc = kernel_timing_of(in, &ms, &launches, reset);
        if (rc) return rc;
    }
    if (outMilliseconds) *outMilliseconds = ms;
    if (outLaunches) *outLaunches = launches;
    return BEAGLE_SUCCESS;
}
int mbamdGetListCounts(int instance, long* out6)
{
    GET_INSTANCE(instance);
    if (!out6) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdGetListCounts: null output");
    for (int i = 0; i < 6; ++i) out6[i] = 0;
    if (in->f64 || in->facade()) return BEAGLE_SUCCESS;
    out6[0] = in->listsTotal; out6[1] = in->listsPath; out6[2] = in->forkedPaths; out6[3] = in->fusedPaths;
    out6[4] = in->listsWalked; out6[5] = in->opsWalked;
    return BEAGLE_SUCCESS;
}
// Device time of whole evaluations while mbamdKernelTiming is on: from the first kernel launched after a log-likelihood
// call to the end of the next integration kernel -- every kernel of a step and the gaps between them (HIP events on the
// engine's stream).  A facade reports the largest of its children.
int mbamdGetStepTiming(int instance, double* outMilliseconds, long* outSteps, int reset)
{
    GET_INSTANCE(instance);
    if (in->f64) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdGetStepTiming: not on a double-precision instance");
    double ms = 0.0;
    long steps = 0;
    if (in->facade()) {
        for (Instance::Child& ch : in->children) {
            double m = 0.0;
            long st = 0;
            int rc = step_timing_of(ch.in, &m, &st, reset);
            if (rc) return rc;
            ms = std::max(ms, m);
            steps = std::max(steps, st);
        }
    } else {
        int rc = step_timing_of(in, &ms, &steps, reset);
        if (rc) return rc;
    }
    if (outMilliseconds) *outMilliseconds = ms;
    if (outSteps) *outSteps = steps;
    return BEAGLE_SUCCESS;
}
int mbamdSetKernelPath(int instance, int path)
{
    GET_INSTANCE(instance);
    if (in->f64) return BEAGLE_SUCCESS;
    if (path < 0 || path > 3) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdSetKernelPath");
    in->path = path;
    return BEAGLE_SUCCESS;
}
int mbamdWalkTrace(int instance, long long* out, int maxSteps, int* outSteps, int* outWaves)
{
    GET_INSTANCE(instance);
    if (in->f64) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdWalkTrace: not on a double-precision instance");
    if (!in->d_trace) return fail(BEAGLE_ERROR_GENERAL, "set MBAMD_WALK_TRACE before creating the instance");
    HIP_TRY(hipStreamSynchronize(in->stream));
    const int n = std::min(maxSteps, std::min(4096, in->lastWalkSteps));
    HIP_TRY(hipMemcpy(out, in->d_trace, (size_t) n * 8 * 3 * sizeof(long long), hipMemcpyDeviceToHost));
    if (outSteps) *outSteps = n;
    if (outWaves) *outWaves = in->walkWaves + 1;
    return BEAGLE_SUCCESS;
}
// the number of child instances behind this instance (1: none) -- pattern partitions x shards
int mbamdGetChildCount(int instance)
{
    GET_INSTANCE_NOFLUSH(instance);
    if (in->f64) return std::max<int>(1, (int) in->f64->parts.size());
    return in->facade() ? (int) in->children.size() : 1;
}
int mbamdSetDeferredResult(int instance, int enable)
{
    GET_INSTANCE(instance);
    if (in->f64) return enable ? fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdSetDeferredResult: not on a double-precision instance") : BEAGLE_SUCCESS;
    if (in->facade()) return enable ? fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdSetDeferredResult: not on a partitioned / sharded instance") : BEAGLE_SUCCESS;
    in->deferred = enable != 0;
    return BEAGLE_SUCCESS;
}
int mbamdReduceLogLikelihood(int instance, double* deviceOut, void* waitingStream)
{
    GET_INSTANCE(instance);
    if (in->f64 || in->facade()) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdReduceLogLikelihood: plain single-precision instances only");
    if (!in->pendingResult) return fail(BEAGLE_ERROR_GENERAL, "no log-likelihood pending");
    if (deviceOut == nullptr) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdReduceLogLikelihood: null output");
    MBAMD_LAUNCH_BARRIER(k_sum_block_sums, 1u, 256, 256 * sizeof(double), in->stream, (const double*) in->h_sums_dev, in->nblocks, deviceOut);
    HIP_TRY(hipGetLastError());
    if (waitingStream != nullptr) {
        if (!in->reduceEvent) HIP_TRY(hipEventCreateWithFlags(&in->reduceEvent, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(in->reduceEvent, in->stream));
        hipStream_t ws{};
        std::memcpy(&ws, &waitingStream, std::min(sizeof ws, sizeof waitingStream));
        HIP_TRY(hipStreamWaitEvent(ws, in->reduceEvent, 0));
    }
    return BEAGLE_SUCCESS;
}
int mbamdGetResourcePciBusId(int resource, char* out, int length)
{
    if (out == nullptr || length < 2) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdGetResourcePciBusId: buffer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || resource < 0 || resource >= n) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdGetResourcePciBusId: resource");
    HIP_TRY(hipDeviceGetPCIBusId(out, length, resource));
    return BEAGLE_SUCCESS;
}
int mbamdGetInstanceDevices(int instance, int* outResources, int maxCount)
{
    GET_INSTANCE_NOFLUSH(instance);
    if (in->f64) { if (outResources && maxCount > 0) outResources[0] = in->device; return 1; }
    if (!in->facade()) { if (outResources && maxCount > 0) outResources[0] = in->device; return 1; }
    int n = 0;
    for (const Instance::Child& c : in->children) { if (outResources && n < maxCount) outResources[n] = c.in->device; ++n; }
    return n;
}
int mbamdFetchLogLikelihood(int instance, double* outSumLogLikelihood)
{
    GET_INSTANCE(instance);
    if (in->f64) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdFetchLogLikelihood: not on a double-precision instance");
    if (in->facade()) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdFetchLogLikelihood: not on a partitioned / sharded instance");
    return in->fetchResult(outSumLogLikelihood);
}


// ---- reports (include/libhmsbeagle/mbamd_reports.h) -------------------------------------------------------------
int mbamdUpdateFinalPartials(int instance, const MbamdFinalOperation* operations, int operationCount)
{
    GET_INSTANCE(instance);
    if (in->f64) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdUpdateFinalPartials: not on a double-precision instance");
    if (operationCount <= 0) return BEAGLE_SUCCESS;
    if (!operations) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdUpdateFinalPartials: null");
    FACADE_ALL(c->finalPass(operations, operationCount));          // (site patterns are independent: every child does its range)
    return in->finalPass(operations, operationCount);
}
int mbamdGetScaledPartials(int instance, int bufferIndex, int cumulativeScaleIndex, float* outPartials, float* outLnScale)
{
    GET_INSTANCE(instance);
    if (in->f64) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdGetScaledPartials: not on a double-precision instance");
    if (!outPartials || !outLnScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdGetScaledPartials: null");
    if (!in->facade()) return in->getScaledPartials(bufferIndex, cumulativeScaleIndex, outPartials, outLnScale);
    if (in->partitionCount > 1) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "mbamdGetScaledPartials: not on a multi-partition instance");
    // pattern shards: each child's [K][count][S] block goes to its pattern range of the caller's [K][P][S] array
    const int S = in->createArgs[3], P = in->createArgs[4], K = in->createArgs[7];
    std::vector<float> part, ln;
    for (Instance::Child& ch : in->children) {
        Instance* c = ch.in;
        (void) hipSetDevice(c->device);
        if (c->hasPending() || !c->pendingJobs.empty()) { int frc = c->flushPending(); if (frc) return frc; }
        part.resize((size_t) K * ch.count * S);
        ln.resize((size_t) ch.count);
        int rc = c->getScaledPartials(bufferIndex, cumulativeScaleIndex, part.data(), ln.data());
        if (rc) return rc;
        for (int k = 0; k < K; ++k)
            std::memcpy(outPartials + ((size_t) k * P + ch.start) * S, part.data() + (size_t) k * ch.count * S, (size_t) ch.count * S * sizeof(float));
        std::memcpy(outLnScale + ch.start, ln.data(), (size_t) ch.count * sizeof(float));
    }
    return BEAGLE_SUCCESS;
}


// ---- Fitch parsimony (include/libhmsbeagle/mbamd_parsimony.h) --------------------------------------------------
#define GET_PARS(id)                                                                                 \
    ParsInstance* pi = pars_lookup(id);                                                              \
    if (!pi) return fail(BEAGLE_ERROR_UNINITIALIZED_INSTANCE, "no such parsimony instance");         \
    (void) hipSetDevice(pi->device)

int mbamdParsCreateInstance(int setCount, int patternCount, int wordsPerSet, int setBits, int likelihoodInstance)
{
    API_TRACE("mbamdParsCreateInstance(sets=%d, patterns=%d, words=%d, bits=%d, like=%d)", setCount, patternCount, wordsPerSet, setBits, likelihoodInstance);
    if (setCount < 1 || patternCount < 1 || (wordsPerSet != 1 && wordsPerSet != 2) || setBits < 1 || setBits > 64 * wordsPerSet)
        return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdParsCreateInstance: bad dimensions");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(BEAGLE_ERROR_NO_RESOURCE, "mbamdParsCreateInstance: no HIP device (this engine has no CPU path)");
    int dev = 0;
    if (likelihoodInstance >= 0) {
        Instance* in = lookup(likelihoodInstance);
        if (!in) return fail(BEAGLE_ERROR_UNINITIALIZED_INSTANCE, "mbamdParsCreateInstance: no such likelihood instance");
        dev = in->facade() ? in->children[0].in->device : in->device;
    }
    ParsInstance* pi = new ParsInstance();
    int rc = pi->create(setCount, patternCount, wordsPerSet, setBits, dev);
    if (rc != BEAGLE_SUCCESS) {
        delete pi;
        return rc;
    }
    std::lock_guard<std::mutex> lock(g_parsMutex);
    for (size_t i = 0; i < g_pars.size(); ++i)
        if (!g_pars[i]) {
            g_pars[i] = pi;
            return (int) i;
        }
    g_pars.push_back(pi);
    return (int) g_pars.size() - 1;
}
int mbamdParsFinalizeInstance(int pars)
{
    ParsInstance* pi = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_parsMutex);
        if (pars < 0 || pars >= (int) g_pars.size() || !g_pars[pars])
            return fail(BEAGLE_ERROR_UNINITIALIZED_INSTANCE, "no such parsimony instance");
        pi = g_pars[pars];
        g_pars[pars] = nullptr;
    }
    delete pi;
    return BEAGLE_SUCCESS;
}
int mbamdParsSetSets(int pars, int setIndex, const unsigned long long* sets)
{
    GET_PARS(pars);
    if (!sets) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdParsSetSets: null");
    return pi->setSets(setIndex, sets);
}
int mbamdParsGetSets(int pars, int setIndex, unsigned long long* outSets)
{
    GET_PARS(pars);
    if (!outSets) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdParsGetSets: null");
    return pi->getSets(setIndex, outSets);
}
int mbamdParsSetPatternWeights(int pars, const float* weights)
{
    GET_PARS(pars);
    if (!weights) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdParsSetPatternWeights: null");
    return pi->setWeights(weights);
}
int mbamdParsDownPass(int pars, const int* ops, int count, double* outLength)
{
    StatTimer st_(ST_PARS_PASS);
    GET_PARS(pars);
    API_TRACE("mbamdParsDownPass(%d ops%s)", count, outLength ? ", length" : "");
    if (count < 0 || (count > 0 && !ops)) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdParsDownPass: arguments");
    return pi->downPass(ops, count, outLength);
}
int mbamdParsFinalPass(int pars, const int* ops, int count)
{
    StatTimer st_(ST_PARS_PASS);
    GET_PARS(pars);
    API_TRACE("mbamdParsFinalPass(%d nodes)", count);
    if (count < 0 || (count > 0 && !ops)) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdParsFinalPass: arguments");
    return pi->finalPass(ops, count);
}
int mbamdParsScore(int pars, const int* tuples, int count, double* outLengths)
{
    StatTimer st_(ST_PARS_SCORE);
    GET_PARS(pars);
    API_TRACE("mbamdParsScore(%d tuples)", count);
    if (count < 0 || (count > 0 && (!tuples || !outLengths))) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdParsScore: arguments");
    return pi->score(tuples, count, outLengths);
}

}  // extern "C"
