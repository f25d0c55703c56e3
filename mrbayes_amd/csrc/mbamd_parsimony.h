// mbamd_parsimony.h -- Fitch parsimony on the device (include/libhmsbeagle/mbamd_parsimony.h).
// Included at the end of mbamd_engine.cpp: kernels, the host object behind a parsimony handle, and the C ABI.
//
// HBM layout: sets[setIndex][P_pad] of T, T the narrowest unsigned type holding the division's state bits
// (u8 DNA/RNA/doublet halves... , u16, u32 amino acids, u64 codons, 2 x u64 beyond): one byte per node and pattern for
// DNA where the reference moves eight (BitsLong).  Site patterns are independent through every pass, so a thread owns
// one pattern and walks the whole operation list: no inter-thread dependency, no barrier, one launch per pass.
// Pure integer/byte work bound by the latency of its own dependency chain (each operation reads sets the same thread
// wrote a few operations earlier, an L2 round trip), not by HBM: at 500 taxa x 20 000 patterns a pass moves 30 MB.
#ifndef MBAMD_PARSIMONY_H_
#define MBAMD_PARSIMONY_H_

namespace mbamd {

struct u128 {
    uint64_t lo, hi;
};
__device__ __host__ __forceinline__ u128 operator&(u128 a, u128 b) { return {a.lo & b.lo, a.hi & b.hi}; }
__device__ __host__ __forceinline__ u128 operator|(u128 a, u128 b) { return {a.lo | b.lo, a.hi | b.hi}; }
__device__ __host__ __forceinline__ bool pars_empty(u128 a) { return (a.lo | a.hi) == 0; }
__device__ __host__ __forceinline__ bool pars_same(u128 a, u128 b) { return a.lo == b.lo && a.hi == b.hi; }
template <class T> __device__ __host__ __forceinline__ bool pars_empty(T a) { return a == 0; }
template <class T> __device__ __host__ __forceinline__ bool pars_same(T a, T b) { return a == b; }
template <class T> __device__ __host__ __forceinline__ T pars_none() { return T(0); }
template <> __device__ __host__ __forceinline__ u128 pars_none<u128>() { return {0, 0}; }

struct ParsOp {
    int a, b, c, d;
};

// GetFitchPartials (reference src/mcmc.c:4794-4846) for every operation of a down-pass (GetParsDP's post-order,
// :4849-4876).  One thread = one pattern; `partial` (may be null): one length per 64-pattern block, added up by the host.
template <class T>
__global__ void __launch_bounds__(64)
k_pars_down(const ParsOp* __restrict__ ops, int n, T* sets, size_t stride, const float* __restrict__ w, double* partial)
{
    const size_t c = (size_t) blockIdx.x * 64 + threadIdx.x;
    const float wc = w[c];
    double len = 0.0;
    for (int i = 0; i < n; ++i) {
        const ParsOp o = ops[i];
        const T l = sets[(size_t) o.b * stride + c], r = sets[(size_t) o.c * stride + c];
        T x = l & r;
        if (pars_empty(x)) {
            x = l | r;
            len += wc;
        }
        sets[(size_t) o.a * stride + c] = x;
    }
    if (!partial) return;
#if defined(MBAMD_HOST_EMU)
    if (threadIdx.x == 0) partial[blockIdx.x] = 0.0;
    partial[blockIdx.x] += len;
#else
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) len += __shfl_down(len, off);
    if (threadIdx.x == 0) partial[blockIdx.x] = len;
#endif
}

// GetParsFP (reference src/mcmc.c:4881-4954) for the nodes of a pre-order list: {node, left, right, ancestor}.
template <class T>
__global__ void __launch_bounds__(64)
k_pars_final(const ParsOp* __restrict__ ops, int n, T* sets, size_t stride)
{
    const size_t c = (size_t) blockIdx.x * 64 + threadIdx.x;
    for (int i = 0; i < n; ++i) {
        const ParsOp o = ops[i];
        const T p = sets[(size_t) o.a * stride + c], a = sets[(size_t) o.d * stride + c];
        const T l = sets[(size_t) o.b * stride + c], r = sets[(size_t) o.c * stride + c];
        T x = p & a;
        if (!pars_same(x, a)) {                      // a change of state between the node and its ancestor is allowed
            if (!pars_empty(l & r))
                x = ((l | r) & a) | p;               // one change through the node: ancestor states a child also has
            else
                x = p | a;                           // two changes: any ancestor state
        }
        sets[(size_t) o.a * stride + c] = x;
    }
}

// candidate lengths (reference src/proposal.c:10783-10876 and the like): block (i, y) sums its share of the patterns
// of tuple i; out[i * gridDim.y + y], added up by the host in a fixed order.
template <class T>
__global__ void __launch_bounds__(64)
k_pars_score(const ParsOp* __restrict__ tuples, const T* __restrict__ sets, size_t stride, int Ppad,
             const float* __restrict__ w, double* __restrict__ out)
{
    const ParsOp o = tuples[blockIdx.x];
    const T* A = o.a >= 0 ? sets + (size_t) o.a * stride : nullptr;
    const T* B = o.b >= 0 ? sets + (size_t) o.b * stride : nullptr;
    const T* C = o.c >= 0 ? sets + (size_t) o.c * stride : nullptr;
    const T* D = o.d >= 0 ? sets + (size_t) o.d * stride : nullptr;
    double len = 0.0;
    for (int c = (int) blockIdx.y * 64 + (int) threadIdx.x; c < Ppad; c += 64 * (int) gridDim.y) {
        T x = A ? A[c] : pars_none<T>(), y = C ? C[c] : pars_none<T>();
        if (B) x = x | B[c];
        if (D) y = y | D[c];
        if (pars_empty(x & y)) len += w[c];
    }
    const size_t slot = (size_t) blockIdx.x * gridDim.y + blockIdx.y;
#if defined(MBAMD_HOST_EMU)
    if (threadIdx.x == 0) out[slot] = 0.0;
    out[slot] += len;
#else
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) len += __shfl_down(len, off);
    if (threadIdx.x == 0) out[slot] = len;
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// host object behind a parsimony handle
class ParsInstance {
public:
    int device = 0;
    int nSets = 0, P = 0, Ppad = 0, words = 1, bits = 0;
    int width = 1;                                  // bytes per set on the device: 1, 2, 4, 8, 16
    hipStream_t stream{};
    bool live = false;
    void* d_sets = nullptr;
    float* d_w = nullptr;
    std::vector<float> h_w;                          // what d_w holds
    static constexpr int RING = 8;
    struct Slot {
        ParsOp* h = nullptr;                         // pinned
        ParsOp* d = nullptr;
        int cap = 0;
        hipEvent_t done = nullptr;
        bool busy = false;
    } ring[RING];
    int next = 0;
    double* d_out = nullptr;                         // per-block partial sums
    double* h_out = nullptr;                         // pinned
    size_t outCap = 0;
    void* h_stage = nullptr;                         // pinned: one set in the device type
    static constexpr int SCORE_Y = 4;

    ~ParsInstance() { destroy(); }

    int create(int setCount, int patterns, int wordsPerSet, int setBits, int dev)
    {
        device = dev;
        nSets = setCount;
        P = patterns;
        Ppad = round_up(patterns, 64);
        words = wordsPerSet;
        bits = setBits;
        width = words == 2 ? 16 : bits <= 8 ? 1 : bits <= 16 ? 2 : bits <= 32 ? 4 : 8;
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        live = true;
        const size_t bytes = (size_t) nSets * Ppad * width;
        HIP_TRY(hipMalloc(&d_sets, bytes));
        HIP_TRY(hipMemsetAsync(d_sets, 0, bytes, stream));                 // SafeCalloc'ed in the reference (src/mcmc.c:6892)
        HIP_TRY(hipMalloc(&d_w, (size_t) Ppad * sizeof(float)));
        HIP_TRY(hipMemsetAsync(d_w, 0, (size_t) Ppad * sizeof(float), stream));
        HIP_TRY(hipHostMalloc(&h_stage, (size_t) Ppad * 16, hipHostMallocDefault));
        HIP_TRY(hipStreamSynchronize(stream));
        return BEAGLE_SUCCESS;
    }

    void destroy()
    {
        if (!live) return;
        (void) hipSetDevice(device);
        (void) hipStreamSynchronize(stream);
        for (Slot& s : ring) {
            if (s.h) (void) hipHostFree(s.h);
            if (s.d) (void) hipFree(s.d);
            if (s.done) (void) hipEventDestroy(s.done);
            s = Slot();
        }
        if (d_sets) (void) hipFree(d_sets);
        if (d_w) (void) hipFree(d_w);
        if (d_out) (void) hipFree(d_out);
        if (h_out) (void) hipHostFree(h_out);
        if (h_stage) (void) hipHostFree(h_stage);
        (void) hipStreamDestroy(stream);
        d_sets = nullptr; d_w = nullptr; d_out = nullptr; h_out = nullptr; h_stage = nullptr; live = false;
    }

    int checkIndex(int idx, bool allowNone, const char* what) const
    {
        if (idx >= 0 && idx < nSets) return BEAGLE_SUCCESS;
        if (allowNone && idx == -1) return BEAGLE_SUCCESS;
        return fail(BEAGLE_ERROR_OUT_OF_RANGE, what, "set index");
    }

    int setSets(int idx, const unsigned long long* src)
    {
        int rc = checkIndex(idx, false, "mbamdParsSetSets");
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(stream));                              // (h_stage may still be in flight)
        const unsigned long long limit = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
        unsigned long long over = 0;
        if (words == 2) {
            u128* o = static_cast<u128*>(h_stage);
            for (int c = 0; c < P; ++c) o[c] = {src[2 * c], src[2 * c + 1]};
            for (int c = P; c < Ppad; ++c) o[c] = {0, 0};
        } else {
            for (int c = 0; c < P; ++c) over |= src[c] & ~limit;
            if (over) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdParsSetSets", "a set has bits beyond setBits");
            switch (width) {
                case 1: fill<uint8_t>(src); break;
                case 2: fill<uint16_t>(src); break;
                case 4: fill<uint32_t>(src); break;
                default: fill<uint64_t>(src); break;
            }
        }
        HIP_TRY(hipMemcpyAsync(static_cast<char*>(d_sets) + (size_t) idx * Ppad * width, h_stage, (size_t) Ppad * width,
                               hipMemcpyHostToDevice, stream));
        return BEAGLE_SUCCESS;
    }
    template <class T> void fill(const unsigned long long* src)
    {
        T* o = static_cast<T*>(h_stage);
        for (int c = 0; c < P; ++c) o[c] = (T) src[c];
        for (int c = P; c < Ppad; ++c) o[c] = 0;
    }

    int getSets(int idx, unsigned long long* out)
    {
        int rc = checkIndex(idx, false, "mbamdParsGetSets");
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpyAsync(h_stage, static_cast<char*>(d_sets) + (size_t) idx * Ppad * width, (size_t) Ppad * width,
                               hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (int c = 0; c < P; ++c) {
            switch (width) {
                case 1: out[c] = static_cast<uint8_t*>(h_stage)[c]; break;
                case 2: out[c] = static_cast<uint16_t*>(h_stage)[c]; break;
                case 4: out[c] = static_cast<uint32_t*>(h_stage)[c]; break;
                case 8: out[c] = static_cast<uint64_t*>(h_stage)[c]; break;
                default: out[2 * c] = static_cast<u128*>(h_stage)[c].lo; out[2 * c + 1] = static_cast<u128*>(h_stage)[c].hi; break;
            }
        }
        return BEAGLE_SUCCESS;
    }

    int setWeights(const float* w)
    {
        if (h_w.size() == (size_t) P && std::memcmp(h_w.data(), w, (size_t) P * sizeof(float)) == 0) return BEAGLE_SUCCESS;
        HIP_TRY(hipStreamSynchronize(stream));                              // (kernels in flight read the old weights; h_w is the source of the copy)
        h_w.assign(w, w + P);
        HIP_TRY(hipMemcpyAsync(d_w, h_w.data(), (size_t) P * sizeof(float), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return BEAGLE_SUCCESS;
    }

    // copy an operation list into the next ring slot (pinned host -> device, stream-ordered)
    int stage(const int* ops, int n, bool allowNone, const char* what, const ParsOp** out, Slot** used)
    {
        for (int i = 0; i < 4 * n; ++i) {
            const bool none = allowNone || (i & 3) == 3;                     // (the fourth field of a down-pass operation is unused)
            int rc = checkIndex(ops[i], none, what);
            if (rc) return rc;
        }
        Slot& s = ring[next];
        next = (next + 1) % RING;
        if (s.busy) {
            HIP_TRY(hipEventSynchronize(s.done));
            s.busy = false;
        }
        if (s.cap < n) {
            if (s.h) (void) hipHostFree(s.h);
            if (s.d) (void) hipFree(s.d);
            s.h = nullptr; s.d = nullptr;
            s.cap = std::max(2 * n, 1024);
            HIP_TRY(hipHostMalloc(&s.h, (size_t) s.cap * sizeof(ParsOp), hipHostMallocDefault));
            HIP_TRY(hipMalloc(&s.d, (size_t) s.cap * sizeof(ParsOp)));
        }
        if (!s.done) HIP_TRY(hipEventCreate(&s.done));
        std::memcpy(s.h, ops, (size_t) n * sizeof(ParsOp));
        HIP_TRY(hipMemcpyAsync(s.d, s.h, (size_t) n * sizeof(ParsOp), hipMemcpyHostToDevice, stream));
        *out = s.d;
        *used = &s;
        return BEAGLE_SUCCESS;
    }
    int release(Slot* s)
    {
        HIP_TRY(hipEventRecord(s->done, stream));
        s->busy = true;
        return BEAGLE_SUCCESS;
    }
    int growOut(size_t doubles)
    {
        if (doubles <= outCap) return BEAGLE_SUCCESS;
        HIP_TRY(hipStreamSynchronize(stream));
        if (d_out) (void) hipFree(d_out);
        if (h_out) (void) hipHostFree(h_out);
        d_out = nullptr; h_out = nullptr;
        outCap = std::max(doubles * 2, (size_t) 4096);
        HIP_TRY(hipMalloc(&d_out, outCap * sizeof(double)));
        HIP_TRY(hipHostMalloc(&h_out, outCap * sizeof(double), hipHostMallocDefault));
        return BEAGLE_SUCCESS;
    }

    template <class T> void launchDown(const ParsOp* ops, int n, double* partial)
    {
        MBAMD_LAUNCH(k_pars_down<T>, (unsigned) (Ppad / 64), 64, 0, stream, ops, n, static_cast<T*>(d_sets), (size_t) Ppad, d_w, partial);
    }
    template <class T> void launchFinal(const ParsOp* ops, int n)
    {
        MBAMD_LAUNCH(k_pars_final<T>, (unsigned) (Ppad / 64), 64, 0, stream, ops, n, static_cast<T*>(d_sets), (size_t) Ppad);
    }
    template <class T> void launchScore(const ParsOp* tuples, int n)
    {
        MBAMD_LAUNCH(k_pars_score<T>, dim3((unsigned) n, SCORE_Y), 64, 0, stream, tuples, static_cast<const T*>(d_sets), (size_t) Ppad, Ppad, d_w, d_out);
    }
#define MBAMD_PARS_DISPATCH(FN, ...)                         \
    switch (width) {                                         \
        case 1: FN<uint8_t>(__VA_ARGS__); break;             \
        case 2: FN<uint16_t>(__VA_ARGS__); break;            \
        case 4: FN<uint32_t>(__VA_ARGS__); break;            \
        case 8: FN<uint64_t>(__VA_ARGS__); break;            \
        default: FN<u128>(__VA_ARGS__); break;               \
    }

    int downPass(const int* ops, int n, double* outLength)
    {
        if (n <= 0) {
            if (outLength) *outLength = 0.0;
            return BEAGLE_SUCCESS;
        }
        const int blocks = Ppad / 64;
        if (outLength) {
            int rc = growOut((size_t) blocks);
            if (rc) return rc;
        }
        const ParsOp* d = nullptr;
        Slot* s = nullptr;
        int rc = stage(ops, n, false, "mbamdParsDownPass", &d, &s);
        if (rc) return rc;
        MBAMD_PARS_DISPATCH(launchDown, d, n, outLength ? d_out : nullptr);
        HIP_TRY(hipGetLastError());
        rc = release(s);
        if (rc) return rc;
        if (outLength) {
            HIP_TRY(hipMemcpyAsync(h_out, d_out, (size_t) blocks * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            double sum = 0.0;
            for (int b = 0; b < blocks; ++b) sum += h_out[b];
            *outLength = sum;
        }
        return BEAGLE_SUCCESS;
    }

    int finalPass(const int* ops, int n)
    {
        if (n <= 0) return BEAGLE_SUCCESS;
        for (int i = 0; i < n; ++i)
            if (ops[4 * i + 3] < 0) return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdParsFinalPass", "a node without ancestor");
        const ParsOp* d = nullptr;
        Slot* s = nullptr;
        int rc = stage(ops, n, false, "mbamdParsFinalPass", &d, &s);
        if (rc) return rc;
        MBAMD_PARS_DISPATCH(launchFinal, d, n);
        HIP_TRY(hipGetLastError());
        return release(s);
    }

    int score(const int* tuples, int n, double* out)
    {
        if (n <= 0) return BEAGLE_SUCCESS;
        int rc = growOut((size_t) n * SCORE_Y);
        if (rc) return rc;
        const ParsOp* d = nullptr;
        Slot* s = nullptr;
        rc = stage(tuples, n, true, "mbamdParsScore", &d, &s);
        if (rc) return rc;
        MBAMD_PARS_DISPATCH(launchScore, d, n);
        HIP_TRY(hipGetLastError());
        rc = release(s);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(h_out, d_out, (size_t) n * SCORE_Y * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (int i = 0; i < n; ++i) {
            double sum = 0.0;
            for (int y = 0; y < SCORE_Y; ++y) sum += h_out[(size_t) i * SCORE_Y + y];
            out[i] = sum;
        }
        return BEAGLE_SUCCESS;
    }
};

static std::mutex g_parsMutex;
static std::vector<ParsInstance*> g_pars;

static ParsInstance* pars_lookup(int id)
{
    std::lock_guard<std::mutex> lock(g_parsMutex);
    return id >= 0 && id < (int) g_pars.size() ? g_pars[id] : nullptr;
}

}  // namespace mbamd

#endif
