// mbamd_engine.cpp -- host side of the MI355X conditional-likelihood engine and its C ABI
// (include/libhmsbeagle/beagle.h).  One Instance = one MrBayes data division: it owns every
// partials / transition-matrix / scale buffer of all local chains in HBM (the reference's
// condLikes/tiProbs/scalers arrays, src/mcmc.c:5703-6510) and turns each BEAGLE call coming from
// src/mbbeagle.c into HIP kernel launches on a private stream.
//
// Built by hipcc for gfx950 (see mrbayes_amd/build.py).  There is no CPU code path in the product:
// without a HIP device beagleCreateInstance fails with BEAGLE_ERROR_NO_RESOURCE.
#if defined(MBAMD_HOST_EMU)
#include <memory>
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define MBAMD_LAUNCH(kernel, grid, block, lds, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, stream, __VA_ARGS__)
#endif

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "libhmsbeagle/beagle.h"
#include "mbamd_kernels.h"

namespace mbamd {

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

enum KernelPath { PATH_AUTO = 0, PATH_GENERIC = 1, PATH_WALK = 2, PATH_MFMA = 3 };

struct Instance {
    int device = 0;
    hipStream_t stream{};
    int tipCount = 0, nBuffers = 0, S = 0, SP = 0, P = 0, Ppad = 0, nEigen = 0, nMatrices = 0, K = 0, nScale = 0;
    bool s4 = false;                 // 4-state float4 layout + tree-walk kernel
    long flags = 0;
    size_t partialsFloats = 0, matrixFloats = 0, eigenDoubles = 0;
    int path = PATH_AUTO;

    std::vector<float*> partials;
    std::vector<uint8_t*> tipStates;
    std::vector<int32_t*> scale;
    float* matrices = nullptr;
    double *d_eigen = nullptr, *d_freqs = nullptr, *d_weights = nullptr, *d_rates = nullptr, *d_pweights = nullptr;
    double *d_site = nullptr, *d_wsite = nullptr, *d_sums = nullptr;
    int nchunks = 0, chunk = 128;
    bool haveSite = false;

    // growable device scratch
    PartialsOp* d_ops = nullptr;      size_t opsCap = 0;
    MatrixJob* d_jobs = nullptr;      size_t jobsCap = 0;
    double* d_ev = nullptr;           size_t evCap = 0;
    void* d_tmp = nullptr;            size_t tmpCap = 0;
    const int32_t** d_ptrs = nullptr; size_t ptrsCap = 0;

    // pinned staging ring for small asynchronous uploads / downloads
    unsigned char* stage = nullptr;
    size_t stageCap = 0, stageOff = 0;
    double* h_sums = nullptr;         // pinned, nchunks doubles

    // timing of the partials kernels
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double timedMs = 0.0;
    long timedLaunches = 0, pendingLaunches = 0;

    bool deferred = false, pendingResult = false;

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

    int ensurePartials(int idx)
    {
        if (partials[idx]) return BEAGLE_SUCCESS;
        float* p = nullptr;
        HIP_TRY(hipMalloc(&p, partialsFloats * sizeof(float)));
        HIP_TRY(hipMemsetAsync(p, 0, partialsFloats * sizeof(float), stream));
        partials[idx] = p;
        return BEAGLE_SUCCESS;
    }
    int ensureScale(int idx)
    {
        if (scale[idx]) return BEAGLE_SUCCESS;
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

    int setTipStates(int tip, const int* states);
    int importPartials(int idx, const double* in, bool hasCategories);
    int getPartials(int idx, double* out);
    int setEigen(int idx, const double* U, const double* Ui, const double* lam);
    int updateMatrices(int eigenIndex, const int* probIdx, const double* lengths, int count);
    int setMatrix(int idx, const double* in);
    int getMatrix(int idx, double* out);
    int updatePartials(const BeagleOperation* ops, int n, int cumIdx);
    int launchWalk(std::vector<PartialsOp>& dev, const std::vector<int>& dstIdx, const std::vector<int>& c1Idx,
                   const std::vector<int>& c2Idx, int32_t* cum);
    int launchGeneric(std::vector<PartialsOp>& dev, const std::vector<int>& dstIdx, const std::vector<int>& c1Idx,
                      const std::vector<int>& c2Idx, int32_t* cum);
    int accumulate(const int* idx, int n, int cumIdx, int sign);
    int integrate(const int* parent, const int* child, const int* prob, const int* wIdx, const int* fIdx,
                  const int* cumIdx, int count, double* out);
    int fetchResult(double* out);
};

static std::mutex g_mutex;
static std::vector<Instance*> g_instances;

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
    s4 = (S == 4 && K >= 1 && K <= 8 && !forceGeneric);
    if (s4) SP = 4;
    else if (S <= 4) SP = 4;
    else if (S <= 8) SP = 8;
    else if (S <= 16) SP = 16;
    else if (S <= 20) SP = 20;
    else if (S <= 32) SP = 32;
    else SP = 64;
    partialsFloats = s4 ? (size_t) K * Ppad * 4 : (size_t) K * S * Ppad;
    matrixFloats = (size_t) K * SP * SP;
    eigenDoubles = (size_t) 2 * S * S + S;
    partials.assign(nBuffers, nullptr);
    tipStates.assign(nBuffers, nullptr);
    scale.assign(std::max(nScale, 1), nullptr);

    HIP_TRY(hipMalloc(&matrices, std::max<size_t>(1, (size_t) nMatrices * matrixFloats) * sizeof(float)));
    HIP_TRY(hipMemsetAsync(matrices, 0, std::max<size_t>(1, (size_t) nMatrices * matrixFloats) * sizeof(float), stream));
    HIP_TRY(hipMalloc(&d_eigen, std::max<size_t>(1, (size_t) nEigen * eigenDoubles) * sizeof(double)));
    HIP_TRY(hipMalloc(&d_freqs, std::max<size_t>(1, (size_t) nEigen * S) * sizeof(double)));
    HIP_TRY(hipMalloc(&d_weights, std::max<size_t>(1, (size_t) nEigen * K) * sizeof(double)));
    HIP_TRY(hipMalloc(&d_rates, (size_t) K * sizeof(double)));
    HIP_TRY(hipMalloc(&d_pweights, (size_t) Ppad * sizeof(double)));
    HIP_TRY(hipMalloc(&d_site, (size_t) Ppad * sizeof(double)));
    HIP_TRY(hipMalloc(&d_wsite, (size_t) Ppad * sizeof(double)));
    nchunks = (P + chunk - 1) / chunk;
    HIP_TRY(hipMalloc(&d_sums, (size_t) nchunks * sizeof(double)));
    HIP_TRY(hipHostMalloc(&h_sums, (size_t) nchunks * sizeof(double), hipHostMallocDefault));
    stageCap = (size_t) 8 << 20;
    HIP_TRY(hipHostMalloc(&stage, stageCap, hipHostMallocDefault));

    // defaults: unit rates, uniform category weights, unit pattern weights (BEAGLE clients normally set them)
    std::vector<double> ones(std::max(Ppad, K), 1.0);
    HIP_TRY(hipMemcpy(d_rates, ones.data(), (size_t) K * sizeof(double), hipMemcpyHostToDevice));
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
    for (float* p : partials) if (p) (void) hipFree(p);
    for (uint8_t* p : tipStates) if (p) (void) hipFree(p);
    for (int32_t* p : scale) if (p) (void) hipFree(p);
    void* bufs[] = {matrices, d_eigen, d_freqs, d_weights, d_rates, d_pweights, d_site, d_wsite, d_sums,
                    d_ops, d_jobs, d_ev, d_tmp, (void*) d_ptrs};
    for (void* b : bufs) if (b) (void) hipFree(b);
    if (h_sums) (void) hipHostFree(h_sums);
    if (stage) (void) hipHostFree(stage);
    for (auto& ev : events) { (void) hipEventDestroy(ev.first); (void) hipEventDestroy(ev.second); }
    (void) hipStreamDestroy(stream);
}

// ---------------------------------------------------------------------------------------------
int Instance::setTipStates(int tip, const int* states)
{
    if (tip < 0 || tip >= nBuffers) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetTipStates: tip index");
    std::vector<uint8_t> h(Ppad, (uint8_t) S);
    for (int c = 0; c < P; ++c) h[c] = (uint8_t) ((states[c] < 0 || states[c] >= S) ? S : states[c]);
    if (!tipStates[tip]) HIP_TRY(hipMalloc(&tipStates[tip], (size_t) Ppad));
    return upload(tipStates[tip], h.data(), (size_t) Ppad);
}

int Instance::importPartials(int idx, const double* in, bool hasCategories)
{
    if (idx < 0 || idx >= nBuffers) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "partials buffer index");
    int rc = ensurePartials(idx);
    if (rc) return rc;
    const size_t nIn = (size_t) (hasCategories ? K : 1) * P * S;
    rc = grow(&d_tmp, &tmpCap, nIn * sizeof(double));
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(d_tmp, in, nIn * sizeof(double), hipMemcpyHostToDevice));
    const size_t total = (size_t) K * P * S;
    const unsigned blocks = (unsigned) ((total + 255) / 256);
    if (s4) MBAMD_LAUNCH(k_import_partials<true>, blocks, 256, 0, stream, (const double*) d_tmp, hasCategories ? 1 : 0, S, K, P, Ppad, partials[idx]);
    else    MBAMD_LAUNCH(k_import_partials<false>, blocks, 256, 0, stream, (const double*) d_tmp, hasCategories ? 1 : 0, S, K, P, Ppad, partials[idx]);
    HIP_TRY(hipGetLastError());
    if (idx < tipCount && tipStates[idx]) {      // a tip switches from compact to partials form
        HIP_TRY(hipStreamSynchronize(stream));
        (void) hipFree(tipStates[idx]);
        tipStates[idx] = nullptr;
    }
    return BEAGLE_SUCCESS;
}

int Instance::getPartials(int idx, double* out)
{
    if (idx < 0 || idx >= nBuffers || !partials[idx]) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleGetPartials: buffer");
    const size_t total = (size_t) K * P * S;
    int rc = grow(&d_tmp, &tmpCap, total * sizeof(double));
    if (rc) return rc;
    const unsigned blocks = (unsigned) ((total + 255) / 256);
    if (s4) MBAMD_LAUNCH(k_export_partials<true>, blocks, 256, 0, stream, (const float*) partials[idx], S, K, P, Ppad, (double*) d_tmp);
    else    MBAMD_LAUNCH(k_export_partials<false>, blocks, 256, 0, stream, (const float*) partials[idx], S, K, P, Ppad, (double*) d_tmp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(out, d_tmp, total * sizeof(double), hipMemcpyDeviceToHost));
    return BEAGLE_SUCCESS;
}

int Instance::setEigen(int idx, const double* U, const double* Ui, const double* lam)
{
    if (idx < 0 || idx >= nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetEigenDecomposition: eigen index");
    std::vector<double> h(eigenDoubles);
    std::memcpy(h.data(), U, sizeof(double) * S * S);
    std::memcpy(h.data() + (size_t) S * S, Ui, sizeof(double) * S * S);
    std::memcpy(h.data() + (size_t) 2 * S * S, lam, sizeof(double) * S);
    return upload(d_eigen + (size_t) idx * eigenDoubles, h.data(), eigenDoubles * sizeof(double));
}

int Instance::updateMatrices(int eigenIndex, const int* probIdx, const double* lengths, int count)
{
    if (eigenIndex < 0 || eigenIndex >= nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdateTransitionMatrices: eigen index");
    if (count <= 0) return BEAGLE_SUCCESS;
    std::vector<MatrixJob> jobs(count);
    for (int i = 0; i < count; ++i) {
        if (probIdx[i] < 0 || probIdx[i] >= nMatrices) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdateTransitionMatrices: matrix index");
        jobs[i].out = matrixPtr(probIdx[i]);
        jobs[i].length = lengths[i];
    }
    int rc = grow((void**) &d_jobs, &jobsCap, sizeof(MatrixJob) * count);
    if (rc) return rc;
    const size_t nev = (size_t) count * K * S;
    rc = grow((void**) &d_ev, &evCap, nev * sizeof(double));
    if (rc) return rc;
    rc = upload(d_jobs, jobs.data(), sizeof(MatrixJob) * count);
    if (rc) return rc;
    const double* eig = d_eigen + (size_t) eigenIndex * eigenDoubles;
    MBAMD_LAUNCH(k_eigen_exponentials, (unsigned) ((nev + 255) / 256), 256, 0, stream, (const MatrixJob*) d_jobs, eig,
                 (const double*) d_rates, S, K, (int) nev, d_ev);
    const int threads = std::min(256, round_up(S * S, 64));
    MBAMD_LAUNCH(k_transition_matrices_ev, (unsigned) (count * K), threads, 0, stream, (const MatrixJob*) d_jobs, eig,
                 (const double*) d_ev, S, SP, K, s4 ? 0 : 1);
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
                if (s4) h[(size_t) k * 16 + i * 4 + j] = v;
                else    h[(size_t) k * SP * SP + (size_t) j * SP + i] = v;
            }
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
                out[((size_t) k * S + i) * S + j] =
                    s4 ? h[(size_t) k * 16 + i * 4 + j] : h[(size_t) k * SP * SP + (size_t) j * SP + i];
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
                if (!partials[ci[s]]) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleUpdatePartials: child buffer was never written");
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
    int32_t* cum = nullptr;
    if (cumIdx != BEAGLE_OP_NONE) {
        int rc = ensureScale(cumIdx);
        if (rc) return rc;
        cum = scale[cumIdx];
    }
    hipEvent_t ev0{}, ev1{};
    if (timing) {
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipEventRecord(ev0, stream));
    }
    int rc;
    if (s4) rc = launchWalk(dev, dstIdx, c1Idx, c2Idx, cum);
    else                            rc = launchGeneric(dev, dstIdx, c1Idx, c2Idx, cum);
    if (timing) {
        HIP_TRY(hipEventRecord(ev1, stream));
        events.emplace_back(ev0, ev1);
    }
    return rc;
}

// Tree-walk path: give every freshly produced partial that is consumed later in the same list an LDS
// stack slot (Belady eviction when the stack is full), so the kernel re-reads children from LDS.
int Instance::launchWalk(std::vector<PartialsOp>& dev, const std::vector<int>& dstIdx, const std::vector<int>& c1Idx,
                         const std::vector<int>& c2Idx, int32_t* cum)
{
    const int n = (int) dev.size();
    int maxSlots = std::min(64, (64 * 1024) / (K * 1024));
    if (const char* dbg = std::getenv("MBAMD_MAX_LDS_SLOTS")) maxSlots = std::max(1, std::min(maxSlots, std::atoi(dbg)));
    // next reader of each op's output (before the buffer is overwritten again)
    std::vector<int> nextUse(n, -1);
    {
        std::vector<int> lastReader(nBuffers, -1);
        for (int o = n - 1; o >= 0; --o) {
            // the output of op o is read by the nearest later reader of dstIdx[o] that comes before
            // the next writer of that buffer; scanning backwards, lastReader holds exactly that.
            nextUse[o] = lastReader[dstIdx[o]];
            lastReader[dstIdx[o]] = -1;             // older readers see older contents
            lastReader[c1Idx[o]] = o;
            lastReader[c2Idx[o]] = o;
        }
    }
    std::vector<int> slotOfBuffer(nBuffers, -1);     // buffer -> LDS slot while resident
    std::vector<char> producedHere(nBuffers, 0);
    std::vector<int> slotBuffer(maxSlots, -1), slotNext(maxSlots, -1);
    int slotsUsed = 0;
    for (int o = 0; o < n; ++o) {
        PartialsOp& d = dev[o];
        const int ci[2] = {c1Idx[o], c2Idx[o]};
        uint8_t* kind[2] = {&d.c1_kind, &d.c2_kind};
        uint8_t* slot[2] = {&d.c1_slot, &d.c2_slot};
        for (int s = 0; s < 2; ++s) {
            if (*kind[s] == CHILD_STATES) continue;
            const int b = ci[s];
            if (slotOfBuffer[b] >= 0) {
                *kind[s] = CHILD_LDS;
                *slot[s] = (uint8_t) slotOfBuffer[b];
            } else if (producedHere[b]) {
                *kind[s] = CHILD_RELOAD;
            }
        }
        // release slots whose content was consumed for the last time by this op
        for (int s = 0; s < 2; ++s) {
            const int b = ci[s];
            const int sl = slotOfBuffer[b];
            if (sl >= 0 && slotNext[sl] <= o) {
                slotOfBuffer[b] = -1;
                slotBuffer[sl] = -1;
            }
        }
        // the destination buffer's previous content (if resident) is dead now
        if (slotOfBuffer[dstIdx[o]] >= 0) {
            slotBuffer[slotOfBuffer[dstIdx[o]]] = -1;
            slotOfBuffer[dstIdx[o]] = -1;
        }
        producedHere[dstIdx[o]] = 1;
        if (nextUse[o] >= 0) {
            int sl = -1;
            for (int t = 0; t < maxSlots; ++t)
                if (slotBuffer[t] < 0) { sl = t; break; }
            if (sl < 0) {                             // evict the resident value needed farthest in the future
                int far = -1;
                for (int t = 0; t < maxSlots; ++t)
                    if (far < 0 || slotNext[t] > slotNext[far]) far = t;
                if (slotNext[far] > nextUse[o]) {
                    slotOfBuffer[slotBuffer[far]] = -1;
                    sl = far;
                }
            }
            if (sl >= 0) {
                slotBuffer[sl] = dstIdx[o];
                slotNext[sl] = nextUse[o];
                slotOfBuffer[dstIdx[o]] = sl;
                d.dst_slot = (uint8_t) sl;
                slotsUsed = std::max(slotsUsed, sl + 1);
            }
        }
    }
    int rc = grow((void**) &d_ops, &opsCap, sizeof(PartialsOp) * n);
    if (rc) return rc;
    rc = upload(d_ops, dev.data(), sizeof(PartialsOp) * n);
    if (rc) return rc;
    const size_t lds = (size_t) std::max(1, slotsUsed) * K * 1024;
    const unsigned grid = (unsigned) (Ppad / 64);
    switch (K) {
#define MBAMD_WALK_CASE(KK) \
    case KK: MBAMD_LAUNCH(k_walk_s4<KK>, grid, 64, lds, stream, (const PartialsOp*) d_ops, n, Ppad, cum); break;
        MBAMD_WALK_CASE(1) MBAMD_WALK_CASE(2) MBAMD_WALK_CASE(3) MBAMD_WALK_CASE(4)
        MBAMD_WALK_CASE(5) MBAMD_WALK_CASE(6) MBAMD_WALK_CASE(7) MBAMD_WALK_CASE(8)
#undef MBAMD_WALK_CASE
        default: return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "tree-walk kernel: category count");
    }
    HIP_TRY(hipGetLastError());
    pendingLaunches += 1;
    return BEAGLE_SUCCESS;
}

template <int SP_, int FK_>
static void launch_gen(Instance& in, const PartialsOp* ops, int count, int32_t* cum)
{
    auto kern = k_partials_gen<SP_, FK_>;
    MBAMD_LAUNCH(kern, dim3(in.Ppad / 64, count), 64, 0, in.stream, ops, in.S, in.K, in.Ppad, cum);
}

// General path: order the operations by dependency level (RAW, WAR and WAW on buffer indices) and
// launch one grid per level.
int Instance::launchGeneric(std::vector<PartialsOp>& dev, const std::vector<int>& dstIdx, const std::vector<int>& c1Idx,
                            const std::vector<int>& c2Idx, int32_t* cum)
{
    const int n = (int) dev.size();
    std::vector<int> lastWrite(nBuffers, -1), lastRead(nBuffers, -1), level(n, 0);
    int nLevels = 0;
    for (int o = 0; o < n; ++o) {
        int l = 0;
        l = std::max(l, lastWrite[c1Idx[o]] + 1);
        l = std::max(l, lastWrite[c2Idx[o]] + 1);
        l = std::max(l, lastWrite[dstIdx[o]] + 1);
        l = std::max(l, lastRead[dstIdx[o]] + 1);
        level[o] = l;
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
    int rc = grow((void**) &d_ops, &opsCap, sizeof(PartialsOp) * n);
    if (rc) return rc;
    rc = upload(d_ops, sorted.data(), sizeof(PartialsOp) * n);
    if (rc) return rc;
    bool anyScale = false;
    for (const PartialsOp& d : sorted) anyScale |= d.scale_mode != SCALE_NONE;
    for (int l = 0; l < nLevels; ++l) {
        int off = start[l];
        int remaining = start[l + 1] - start[l];
        while (remaining > 0) {
            const int count = std::min(remaining, 32768);
            const PartialsOp* ops = d_ops + off;
            bool fused = true;
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
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int Instance::accumulate(const int* idx, int n, int cumIdx, int sign)
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
    rc = grow((void**) &d_ptrs, &ptrsCap, sizeof(void*) * n);
    if (rc) return rc;
    rc = upload((void*) d_ptrs, ptrs.data(), sizeof(void*) * n);
    if (rc) return rc;
    MBAMD_LAUNCH(k_scale_accumulate, (unsigned) ((Ppad + 255) / 256), 256, 0, stream, (const int32_t* const*) d_ptrs, n, sign,
                 Ppad, scale[cumIdx]);
    HIP_TRY(hipGetLastError());
    return BEAGLE_SUCCESS;
}

int Instance::integrate(const int* parent, const int* child, const int* prob, const int* wIdx, const int* fIdx,
                        const int* cumIdx, int count, double* out)
{
    if (count < 1 || count > MBAMD_MAX_SUBSETS) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "log-likelihood: subset count");
    IntegrateArgs a;
    std::memset(&a, 0, sizeof a);
    a.count = count;
    for (int n = 0; n < count; ++n) {
        if (parent[n] < 0 || parent[n] >= nBuffers || !partials[parent[n]])
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "log-likelihood: parent buffer");
        a.parent[n] = partials[parent[n]];
        if (child) {
            const int ci = child[n];
            if (ci < 0 || ci >= nBuffers || prob[n] < 0 || prob[n] >= nMatrices)
                return fail(BEAGLE_ERROR_OUT_OF_RANGE, "edge log-likelihood: child buffer / matrix");
            if (tipStates[ci]) { a.child[n] = tipStates[ci]; a.child_kind[n] = CHILD_STATES; }
            else if (partials[ci]) { a.child[n] = partials[ci]; a.child_kind[n] = CHILD_PARTIALS; }
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
    if (s4) MBAMD_LAUNCH(k_integrate_lnl<true>, (unsigned) (Ppad / 64), 64, 0, stream, a, S, SP, K, P, Ppad, (const double*) d_pweights, d_site, d_wsite);
    else    MBAMD_LAUNCH(k_integrate_lnl<false>, (unsigned) (Ppad / 64), 64, 0, stream, a, S, SP, K, P, Ppad, (const double*) d_pweights, d_site, d_wsite);
    MBAMD_LAUNCH(k_chunk_sums, (unsigned) ((nchunks + 63) / 64), 64, 0, stream, (const double*) d_wsite, P, chunk, nchunks, d_sums);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_sums, d_sums, (size_t) nchunks * sizeof(double), hipMemcpyDeviceToHost, stream));
    haveSite = true;
    pendingResult = true;
    if (deferred) {
        if (out) *out = 0.0;
        return BEAGLE_SUCCESS;
    }
    return fetchResult(out);
}

int Instance::fetchResult(double* out)
{
    if (!pendingResult) return fail(BEAGLE_ERROR_GENERAL, "no log-likelihood pending");
    HIP_TRY(hipStreamSynchronize(stream));
    pendingResult = false;
    double s = 0.0;
    for (int i = 0; i < nchunks; ++i) s += h_sums[i];
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
        g_resourceVec[i].supportFlags = kSupport;
        g_resourceVec[i].requiredFlags = 0;
    }
    g_resources.list = g_resourceVec.data();
    g_resources.length = n;
}

}  // namespace mbamd

// =============================================================================================
// C ABI
// =============================================================================================
using namespace mbamd;

#define GET_INSTANCE(id)                                                                             \
    Instance* in = lookup(id);                                                                       \
    if (!in) return fail(BEAGLE_ERROR_UNINITIALIZED_INSTANCE, "no such instance");                   \
    (void) hipSetDevice(in->device)

extern "C" {

const char* beagleGetVersion(void) { return "mbamd-0.1 (HIP/gfx950; BEAGLE API 3.x compatible subset)"; }
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

int beagleCreateInstance(int tipCount, int partialsBufferCount, int compactBufferCount, int stateCount,
                         int patternCount, int eigenBufferCount, int matrixBufferCount, int categoryCount,
                         int scaleBufferCount, int* resourceList, int resourceCount, long preferenceFlags,
                         long requirementFlags, BeagleInstanceDetails* returnInfo)
{
    (void) preferenceFlags;
    if (tipCount < 0 || partialsBufferCount < 0 || compactBufferCount < 0 || stateCount < 2 || patternCount < 1 ||
        eigenBufferCount < 0 || matrixBufferCount < 0 || categoryCount < 1 || scaleBufferCount < 0)
        return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleCreateInstance: bad dimensions");
    if (stateCount > 64) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleCreateInstance: more than 64 states");
    if (requirementFlags & (BEAGLE_FLAG_PRECISION_DOUBLE | BEAGLE_FLAG_EIGEN_COMPLEX | BEAGLE_FLAG_PROCESSOR_CPU |
                            BEAGLE_FLAG_FRAMEWORK_CUDA | BEAGLE_FLAG_FRAMEWORK_OPENCL | BEAGLE_FLAG_SCALERS_RAW))
        return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleCreateInstance: unsupported requirement flags");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(BEAGLE_ERROR_NO_RESOURCE, "beagleCreateInstance: no HIP device (this engine has no CPU path)");
    int dev = 0;
    if (resourceList && resourceCount > 0) {
        dev = -1;
        for (int i = 0; i < resourceCount; ++i)
            if (resourceList[i] >= 0 && resourceList[i] < ndev) { dev = resourceList[i]; break; }
        if (dev < 0) return fail(BEAGLE_ERROR_NO_RESOURCE, "beagleCreateInstance: requested resource not available");
    }
    Instance* in = new Instance();
    in->flags = kSupport | (requirementFlags & (BEAGLE_FLAG_SCALING_ALWAYS | BEAGLE_FLAG_SCALING_DYNAMIC));
    int rc = in->create(tipCount, partialsBufferCount, compactBufferCount, stateCount, patternCount, eigenBufferCount,
                        matrixBufferCount, categoryCount, scaleBufferCount, dev);
    if (rc != BEAGLE_SUCCESS) {
        in->destroy();
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
        returnInfo->resourceNumber = dev;
        returnInfo->resourceName = (dev < g_resources.length) ? g_resources.list[dev].name : const_cast<char*>("HIP device");
        returnInfo->implName = const_cast<char*>(in->s4 ? "mbamd HIP gfx950: 4-state tree-walk kernels"
                                                        : "mbamd HIP gfx950: general-state kernels");
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
    in->destroy();
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
        if (in) { in->destroy(); delete in; }
    return BEAGLE_SUCCESS;
}

int beagleSetTipStates(int instance, int tipIndex, const int* inStates)
{
    GET_INSTANCE(instance);
    return in->setTipStates(tipIndex, inStates);
}
int beagleSetTipPartials(int instance, int tipIndex, const double* inPartials)
{
    GET_INSTANCE(instance);
    return in->importPartials(tipIndex, inPartials, false);
}
int beagleSetPartials(int instance, int bufferIndex, const double* inPartials)
{
    GET_INSTANCE(instance);
    return in->importPartials(bufferIndex, inPartials, true);
}
int beagleGetPartials(int instance, int bufferIndex, int scaleIndex, double* outPartials)
{
    GET_INSTANCE(instance);
    if (scaleIndex != BEAGLE_OP_NONE) return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleGetPartials: scaleIndex must be BEAGLE_OP_NONE");
    return in->getPartials(bufferIndex, outPartials);
}
int beagleSetEigenDecomposition(int instance, int eigenIndex, const double* inEigenVectors,
                                const double* inInverseEigenVectors, const double* inEigenValues)
{
    GET_INSTANCE(instance);
    return in->setEigen(eigenIndex, inEigenVectors, inInverseEigenVectors, inEigenValues);
}
int beagleSetStateFrequencies(int instance, int idx, const double* f)
{
    GET_INSTANCE(instance);
    if (idx < 0 || idx >= in->nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetStateFrequencies: index");
    return in->upload(in->d_freqs + (size_t) idx * in->S, f, sizeof(double) * in->S);
}
int beagleSetCategoryWeights(int instance, int idx, const double* w)
{
    GET_INSTANCE(instance);
    if (idx < 0 || idx >= in->nEigen) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleSetCategoryWeights: index");
    return in->upload(in->d_weights + (size_t) idx * in->K, w, sizeof(double) * in->K);
}
int beagleSetCategoryRates(int instance, const double* r)
{
    GET_INSTANCE(instance);
    return in->upload(in->d_rates, r, sizeof(double) * in->K);
}
int beagleSetPatternWeights(int instance, const double* w)
{
    GET_INSTANCE(instance);
    return in->upload(in->d_pweights, w, sizeof(double) * in->P);
}
int beagleUpdateTransitionMatrices(int instance, int eigenIndex, const int* probabilityIndices,
                                   const int* firstDerivativeIndices, const int* secondDerivativeIndices,
                                   const double* edgeLengths, int count)
{
    GET_INSTANCE(instance);
    if (firstDerivativeIndices || secondDerivativeIndices)
        return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleUpdateTransitionMatrices: derivatives");
    return in->updateMatrices(eigenIndex, probabilityIndices, edgeLengths, count);
}
int beagleSetTransitionMatrix(int instance, int matrixIndex, const double* inMatrix, double paddedValue)
{
    (void) paddedValue;
    GET_INSTANCE(instance);
    return in->setMatrix(matrixIndex, inMatrix);
}
int beagleGetTransitionMatrix(int instance, int matrixIndex, double* outMatrix)
{
    GET_INSTANCE(instance);
    return in->getMatrix(matrixIndex, outMatrix);
}
int beagleUpdatePartials(int instance, const BeagleOperation* operations, int operationCount, int cumulativeScaleIndex)
{
    GET_INSTANCE(instance);
    return in->updatePartials(operations, operationCount, cumulativeScaleIndex);
}
int beagleWaitForPartials(int instance, const int* destinationPartials, int destinationPartialsCount)
{
    (void) destinationPartials; (void) destinationPartialsCount;
    GET_INSTANCE(instance);
    HIP_TRY(hipStreamSynchronize(in->stream));
    return BEAGLE_SUCCESS;
}
int beagleAccumulateScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex)
{
    GET_INSTANCE(instance);
    return in->accumulate(scaleIndices, count, cumulativeScaleIndex, +1);
}
int beagleRemoveScaleFactors(int instance, const int* scaleIndices, int count, int cumulativeScaleIndex)
{
    GET_INSTANCE(instance);
    return in->accumulate(scaleIndices, count, cumulativeScaleIndex, -1);
}
int beagleResetScaleFactors(int instance, int cumulativeScaleIndex)
{
    GET_INSTANCE(instance);
    if (cumulativeScaleIndex < 0 || cumulativeScaleIndex >= in->nScale)
        return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleResetScaleFactors: index");
    if (!in->scale[cumulativeScaleIndex]) return in->ensureScale(cumulativeScaleIndex);   // allocated zeroed
    HIP_TRY(hipMemsetAsync(in->scale[cumulativeScaleIndex], 0, (size_t) in->Ppad * sizeof(int32_t), in->stream));
    return BEAGLE_SUCCESS;
}
int beagleCopyScaleFactors(int instance, int destScalingIndex, int srcScalingIndex)
{
    GET_INSTANCE(instance);
    if (destScalingIndex < 0 || destScalingIndex >= in->nScale || srcScalingIndex < 0 || srcScalingIndex >= in->nScale)
        return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleCopyScaleFactors: index");
    int rc = in->ensureScale(destScalingIndex);
    if (rc) return rc;
    rc = in->ensureScale(srcScalingIndex);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(in->scale[destScalingIndex], in->scale[srcScalingIndex], (size_t) in->Ppad * sizeof(int32_t),
                           hipMemcpyDeviceToDevice, in->stream));
    return BEAGLE_SUCCESS;
}
int beagleGetScaleFactors(int instance, int srcScalingIndex, double* outScaleFactors)
{
    GET_INSTANCE(instance);
    if (srcScalingIndex < 0 || srcScalingIndex >= in->nScale) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "beagleGetScaleFactors: index");
    int rc = in->ensureScale(srcScalingIndex);
    if (rc) return rc;
    std::vector<int32_t> h(in->Ppad);
    HIP_TRY(hipStreamSynchronize(in->stream));
    HIP_TRY(hipMemcpy(h.data(), in->scale[srcScalingIndex], (size_t) in->Ppad * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int c = 0; c < in->P; ++c) outScaleFactors[c] = (double) h[c] * 0.69314718055994530942;
    return BEAGLE_SUCCESS;
}
int beagleCalculateRootLogLikelihoods(int instance, const int* bufferIndices, const int* categoryWeightsIndices,
                                      const int* stateFrequenciesIndices, const int* cumulativeScaleIndices, int count,
                                      double* outSumLogLikelihood)
{
    GET_INSTANCE(instance);
    return in->integrate(bufferIndices, nullptr, nullptr, categoryWeightsIndices, stateFrequenciesIndices,
                         cumulativeScaleIndices, count, outSumLogLikelihood);
}
int beagleCalculateEdgeLogLikelihoods(int instance, const int* parentBufferIndices, const int* childBufferIndices,
                                      const int* probabilityIndices, const int* firstDerivativeIndices,
                                      const int* secondDerivativeIndices, const int* categoryWeightsIndices,
                                      const int* stateFrequenciesIndices, const int* cumulativeScaleIndices, int count,
                                      double* outSumLogLikelihood, double* outSumFirstDerivative,
                                      double* outSumSecondDerivative)
{
    GET_INSTANCE(instance);
    if (firstDerivativeIndices || secondDerivativeIndices || outSumFirstDerivative || outSumSecondDerivative)
        return fail(BEAGLE_ERROR_NO_IMPLEMENTATION, "beagleCalculateEdgeLogLikelihoods: derivatives");
    return in->integrate(parentBufferIndices, childBufferIndices, probabilityIndices, categoryWeightsIndices,
                         stateFrequenciesIndices, cumulativeScaleIndices, count, outSumLogLikelihood);
}
int beagleGetSiteLogLikelihoods(int instance, double* outLogLikelihoods)
{
    GET_INSTANCE(instance);
    if (!in->haveSite) return fail(BEAGLE_ERROR_GENERAL, "beagleGetSiteLogLikelihoods: no likelihood computed yet");
    HIP_TRY(hipStreamSynchronize(in->stream));
    HIP_TRY(hipMemcpy(outLogLikelihoods, in->d_site, (size_t) in->P * sizeof(double), hipMemcpyDeviceToHost));
    return BEAGLE_SUCCESS;
}

// ---- engine extensions ---------------------------------------------------------------------
int mbamdSynchronize(int instance)
{
    GET_INSTANCE(instance);
    HIP_TRY(hipStreamSynchronize(in->stream));
    return BEAGLE_SUCCESS;
}
int mbamdKernelTiming(int instance, int enable)
{
    GET_INSTANCE(instance);
    in->timing = enable != 0;
    return BEAGLE_SUCCESS;
}
int mbamdGetKernelTiming(int instance, double* outMilliseconds, long* outLaunches, int reset)
{
    GET_INSTANCE(instance);
    HIP_TRY(hipStreamSynchronize(in->stream));
    for (auto& ev : in->events) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        in->timedMs += ms;
        (void) hipEventDestroy(ev.first);
        (void) hipEventDestroy(ev.second);
    }
    in->events.clear();
    in->timedLaunches += in->pendingLaunches;
    in->pendingLaunches = 0;
    if (outMilliseconds) *outMilliseconds = in->timedMs;
    if (outLaunches) *outLaunches = in->timedLaunches;
    if (reset) { in->timedMs = 0.0; in->timedLaunches = 0; }
    return BEAGLE_SUCCESS;
}
int mbamdSetKernelPath(int instance, int path)
{
    GET_INSTANCE(instance);
    if (path < 0 || path > 3) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdSetKernelPath");
    in->path = path;
    return BEAGLE_SUCCESS;
}
int mbamdSetDeferredResult(int instance, int enable)
{
    GET_INSTANCE(instance);
    in->deferred = enable != 0;
    return BEAGLE_SUCCESS;
}
int mbamdFetchLogLikelihood(int instance, double* outSumLogLikelihood)
{
    GET_INSTANCE(instance);
    return in->fetchResult(outSumLogLikelihood);
}

}  // extern "C"
