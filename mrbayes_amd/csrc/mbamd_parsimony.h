// mbamd_parsimony.h -- Fitch parsimony on the device (include/libhmsbeagle/mbamd_parsimony.h).
// Included at the end of mbamd_engine.cpp: kernels, the host object behind a parsimony handle, and the C ABI.
//
// HBM layout: sets[setIndex][P_pad] of T, T the narrowest unsigned type holding the division's state bits (u8 DNA / RNA,
// u16, u32 amino acids, u64 codons, 2 x u64 beyond): one byte per node and pattern for DNA where the reference moves eight
// (BitsLong).  Two more rows: a junk destination and an all-ones source for no-op steps.  Site patterns are independent
// through every pass, so a thread owns one pattern and walks a whole compiled program (k_pars_walk): no inter-thread
// dependency, no barrier; queued passes run as one launch.  Pure integer/byte work; with P/64 waves on 1024 SIMDs the bound
// is one wave's instruction stream and dependency chain, not HBM (a pass at 500 taxa x 20 000 patterns moves 30 MB).
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

// One step of a compiled pass program (ParsInstance::flush): a Fitch down-pass operation (GetFitchPartials, reference
// src/mcmc.c:4794-4846, in GetParsDP's post-order :4849-4876) or a final-pass node (GetParsFP, :4881-4954).  All steps
// of a chunk are of one kind (the host pads with no-ops where the kind changes: sources = the all-ones row, destination =
// the junk row).  Every operand names BOTH a set in HBM (prefetched a chunk ahead; the all-ones row when unused) and a
// ring code: 0xFF = take the prefetched value, else the LDS ring slot where a step of this or the previous chunk left it.
struct ParsStep {
    int kind;            // 0 down-pass, 1 final pass (the kind of step 0 of a chunk is the chunk's)
    int dest;
    int g[4];            // down: {source1, source2, -, -};  final: {node, left, right, ancestor}
    unsigned ring;       // four 8-bit ring codes
    int pad_;
};
enum { PARS_DOWN = 0, PARS_FINAL = 1 };

// The descriptors of one chunk (CH steps x 8 dwords <= 64 dwords): ONE vector load, lane l holding dword l, fetched two
// chunks ahead; the uniform fields come back out with v_readlane.  (Scalar loads would do, but a wave that is alone on its
// CU misses the scalar cache on every one of them.)
}  // namespace mbamd
#include <mbamd_dev_pars.h>      // ParsDesc, pars_desc_load (csrc/device/)
namespace mbamd {

// the type a set is computed in: 32 bits for the narrow storage types (no sub-dword packing in registers or in LDS)
template <class T> struct ParsWide { typedef uint32_t type; };
template <> struct ParsWide<uint64_t> { typedef uint64_t type; };
template <> struct ParsWide<u128> { typedef u128 type; };
template <class T, class X> __device__ __host__ __forceinline__ T pars_narrow(X x) { return (T) x; }
template <> __device__ __host__ __forceinline__ u128 pars_narrow<u128, u128>(u128 x) { return x; }

// The walk: one thread = one pattern runs a whole program, so there is no inter-thread dependency inside a program.  With
// one wave per SIMD at best (P / 64 waves for 1024 SIMDs) the bound is that wave's own instruction stream and the latency
// of its dependency chain, so the kernel is built to keep both short.  Three stages a chunk (CH steps) apart: descriptors
// (one vector load), operands (loaded into a second register set through scalar row addresses), execution.  A step's
// operands come from the LDS ring when a step of the last two chunks produced them (results are forwarded through a
// wave-private ring of 2 CH slots: ds_write -> ds_read instead of an L2 round trip of the store and the load) and from
// HBM otherwise.  Hardware keeps a wave's loads behind its own earlier stores to the same address, so "produced before the
// previous chunk began" is all the operand prefetch needs.  Within a kind every step issues the same loads, one store and
// one ring write, so the compiler's vmcnt waits are exact.  The host pads every program with two chunks of no-ops that are
// fetched but never run.
// Round 5 -- the TREE in parallel: a workgroup is W waves on the SAME 64 patterns; the host cuts the queued passes into
// PHASES of W independent programs (ParsInstance::flush: a pass's post-order is striped over the waves -- a stripe is a set of
// subtrees --, a step whose inputs come from two waves opens the next phase); the waves meet at a workgroup barrier between
// phases, and what crosses waves crosses through memory (written before the barrier, read after it).  The chain a wave walks
// is a stripe plus its share of the top of the tree instead of the whole tree.  The buffer `prog` starts with the table
// int [MBAMD_PARS_MAXPHASES][MBAMD_PARS_MAXW][2] = {first chunk, number of chunks (even; 0: nothing to do)}.
#define MBAMD_PARS_MAXW 8
#define MBAMD_PARS_MAXPHASES 16
#define MBAMD_PARS_HEADER_INTS (MBAMD_PARS_MAXPHASES * MBAMD_PARS_MAXW * 2)
template <class T, int CH>
__global__ void __launch_bounds__(64 * MBAMD_PARS_MAXW)
k_pars_walk(const int* __restrict__ prog, int nphases, T* sets, unsigned stride, const float* __restrict__ w, double* partial)
{
    typedef typename ParsWide<T>::type X;
    const int wave = mbd_wave_index(), W = (int) (blockDim.x >> 6);
    X* ring = mbd_dyn_lds<X>() + (size_t) wave * (2 * CH * 64);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned c = blockIdx.x * 64u + lane;                  // (nSets + 2) * P_pad < 2^32, checked at create
    const float wc = w[c];
    double len = 0.0;
    const ParsStep* const steps0 = reinterpret_cast<const ParsStep*>(prog + MBAMD_PARS_HEADER_INTS);

    auto row = [&](int set) { return sets + (size_t) ((unsigned) set * stride); };      // uniform: a scalar address
    auto segment = [&](const ParsStep* steps, int nchunks) {
        X A[CH][4], B[CH][4];
        auto prefetch = [&](const ParsDesc& d, X (&buf)[CH][4]) {
            const bool fin = d.get(0) == PARS_FINAL;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                buf[k][0] = (X) row(d.get(8 * k + 2))[c];
                buf[k][1] = (X) row(d.get(8 * k + 3))[c];
            }
            if (fin) {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    buf[k][2] = (X) row(d.get(8 * k + 4))[c];
                    buf[k][3] = (X) row(d.get(8 * k + 5))[c];
                }
            }
        };
        auto operand = [&](unsigned codes, int j, X fetched) {
            const unsigned code = (codes >> (8 * j)) & 0xFFu;
            return code != 0xFFu ? ring[code * 64 + lane] : fetched;
        };
        auto run = [&](int half, const ParsDesc& d, X (&buf)[CH][4]) {
            if (d.get(0) == PARS_DOWN) {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const unsigned codes = (unsigned) d.get(8 * k + 6);
                    const X a = operand(codes, 0, buf[k][0]), b = operand(codes, 1, buf[k][1]);
                    X x = a & b;
                    if (pars_empty(x)) {                      // no common state: the union, and one more step of length
                        x = a | b;
                        len += wc;
                    }
                    row(d.get(8 * k + 1))[c] = pars_narrow<T, X>(x);
                    ring[(half * CH + k) * 64 + lane] = x;
                }
            } else {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const unsigned codes = (unsigned) d.get(8 * k + 6);
                    const X p = operand(codes, 0, buf[k][0]), l = operand(codes, 1, buf[k][1]);
                    const X r = operand(codes, 2, buf[k][2]), a = operand(codes, 3, buf[k][3]);
                    X x = p & a;
                    if (!pars_same(x, a))                     // a change between the node and its ancestor is allowed:
                        x = !pars_empty(l & r) ? (((l | r) & a) | p)       // one change through the node, or
                                               : (p | a);                   // two (any ancestor state)
                    row(d.get(8 * k + 1))[c] = pars_narrow<T, X>(x);
                    ring[(half * CH + k) * 64 + lane] = x;
                }
            }
        };
        ParsDesc d0 = pars_desc_load<CH>(steps, 0, lane), d1 = pars_desc_load<CH>(steps, 1, lane);
        prefetch(d0, A);
        for (int ch = 0; ch < nchunks; ch += 2) {          // (nchunks is even: the ring slot of a step is a compile-time constant)
            ParsDesc d2 = pars_desc_load<CH>(steps, ch + 2, lane);
            prefetch(d1, B);
            run(0, d0, A);
            ParsDesc d3 = pars_desc_load<CH>(steps, ch + 3, lane);
            prefetch(d2, A);
            run(1, d1, B);
            d0 = d2;
            d1 = d3;
        }
    };
    for (int ph = 0; ph < nphases; ++ph) {
        const int first = prog[(ph * MBAMD_PARS_MAXW + wave) * 2], n = prog[(ph * MBAMD_PARS_MAXW + wave) * 2 + 1];
        if (n > 0) segment(steps0 + (size_t) first * CH, n);
        if (W > 1 && ph + 1 < nphases) MBAMD_SYNC();        // what this phase stored is in memory for every wave of the workgroup
    }
    if (!partial) return;
    if (W > 1) {                                             // the waves' lengths of one pattern meet in LDS (integer-valued weights: exact in any order)
        MBAMD_SYNC();
        double* red = mbd_dyn_lds<double>();
        red[threadIdx.x] = len;
        MBAMD_SYNC();
        if (wave != 0) return;
        for (int v = 1; v < W; ++v) len += red[v * 64 + lane];
    }
    mbd_wave_sum_store(len, partial + blockIdx.x);
}

// candidate lengths (reference src/proposal.c:10783-10876 and the like): block (i, y) sums its share of the patterns
// of tuple i; out[i * gridDim.y + y], added up by the host in a fixed order.  Four waves per block (round 5: a wave's loop over
// its patterns is a chain of dependent round trips -- 78 of them at 20 000 patterns with one wave per block, 20 with four).
#define MBAMD_PARS_SCORE_THREADS 256
template <class T>
__global__ void __launch_bounds__(MBAMD_PARS_SCORE_THREADS)
k_pars_score(const ParsOp* __restrict__ tuples, const T* __restrict__ sets, size_t stride, int Ppad,
             const float* __restrict__ w, double* __restrict__ out)
{
    const ParsOp o = tuples[blockIdx.x];
    const T* A = o.a >= 0 ? sets + (size_t) o.a * stride : nullptr;
    const T* B = o.b >= 0 ? sets + (size_t) o.b * stride : nullptr;
    const T* C = o.c >= 0 ? sets + (size_t) o.c * stride : nullptr;
    const T* D = o.d >= 0 ? sets + (size_t) o.d * stride : nullptr;
    double len = 0.0;
    for (int c = (int) blockIdx.y * MBAMD_PARS_SCORE_THREADS + (int) threadIdx.x; c < Ppad; c += MBAMD_PARS_SCORE_THREADS * (int) gridDim.y) {
        T x = A ? A[c] : pars_none<T>(), y = C ? C[c] : pars_none<T>();
        if (B) x = x | B[c];
        if (D) y = y | D[c];
        if (pars_empty(x & y)) len += w[c];
    }
    // (integer-valued weights: the sum is exact in any order)
    double* red = mbd_dyn_lds<double>();
    red[threadIdx.x] = len;
    MBAMD_SYNC();
    if (threadIdx.x >= 64) return;
    len = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
    const size_t slot = (size_t) blockIdx.x * gridDim.y + blockIdx.y;
    mbd_wave_sum_store(len, out + slot);
}

// ------------------------------------------------------------------------------------------------------------------
// host object behind a parsimony handle.  Down-pass and final-pass calls only append to a program; it is compiled (ring
// codes, padding) and run as ONE launch when a value is asked for -- a ParsSPR1 move (two down-passes, two final passes,
// one candidate matrix: reference src/proposal.c:10700-10743) is two launches and one wait.
class ParsInstance {
public:
    int device = 0;
    int nSets = 0, P = 0, Ppad = 0, words = 1, bits = 0;
    int width = 1;                                  // bytes per set on the device: 1, 2, 4, 8, 16
    hipStream_t stream{};
    bool live = false;
    void* d_sets = nullptr;                          // nSets + 2 rows: [nSets] takes what no-op steps store, [nSets + 1] is all ones (their sources)
    float* d_w = nullptr;                            // the pattern weights the next launch reads: one of d_wbuf
    float* d_wbuf[2] = {nullptr, nullptr};           // double-buffered: an upload never touches what kernels in flight read
    float* h_wpin[2] = {nullptr, nullptr};           // pinned sources of the asynchronous copies
    hipEvent_t wDone[2] = {};
    bool wBusy[2] = {false, false};
    int wNext = 0;
    std::vector<float> h_w;                          // what d_w holds
    static constexpr int RING = 4;
    struct Slot {
        void* h = nullptr;                           // pinned
        void* hdev = nullptr;                        // ... as the device sees it
        void* d = nullptr;
        size_t cap = 0;
        hipEvent_t done{};
        bool haveEvent = false, busy = false;
    } ring[RING];
    int next = 0;
    double* d_out = nullptr;                         // per-block partial sums: where the kernels write them -- the device address of h_out
    double* h_out = nullptr;                         // pinned, mapped: the host reads the sums where the kernels put them (no copy)
    size_t outCap = 0;
    // a result is waited for by polling a word the stream writes behind the kernel (as Instance::fetchResult does: the runtime's
    // wait on a stream costs ~25 us); MBAMD_NO_POLL=1: hipStreamSynchronize
    uint32_t* h_flag = nullptr;                      // pinned
    uint32_t* h_flag_dev = nullptr;
    bool pollSums = false;                           // the sums as their own completion signal (armSums)
    uint32_t flagSeq = 0;
    bool poll = false;
    void* h_stage = nullptr;                         // pinned: one set in the device type
    static constexpr int SCORE_Y = 4;
    struct Pending {
        int kind, a, b, c, d;
        int passPos, passLen, passId;                // position in / length of / number of the pass (one mbamdPars*Pass call) the step came with
    };
    std::vector<Pending> pending;
    struct SetState {                                // per set, valid while `stamp` is the current one (flush)
        int stamp = 0;
        int wPhase = -1, wWave = -1;                 // the step that wrote it last: its phase and wave
        int rPhase = -1, rWave = -1;                 // the latest phase it was read in since, and that reader's wave (-1: several)
        int wPos = 0;                                // (second pass) position of its writer in the program being emitted
        int binStamp = 0, bin = -1;                  // the wave its subtree of the latest down pass was dealt to (-1: it is in the top of the tree)
    };
    std::vector<SetState> state;
    int stamp = 0, passCounter = 0;
    std::vector<int> phaseOf, waveOf, bucketStart, bucketFill, order, header, binWave, sizeOf, kid1, kid2;
    std::vector<char> isRoot, below;
    std::vector<std::pair<int, int>> heap;           // (subtree size, its root step)
    std::vector<ParsStep> compiled;
    std::vector<unsigned char> image;                // header + programs as they go to the device
    bool verbose = false;                            // MBAMD_VERBOSE
    int phaseLimit = MBAMD_PARS_MAXPHASES;           // phases per launch (MBAMD_PARS_PHASE_LIMIT: tests force programs over several launches)
    int waves = 0;                                   // MBAMD_PARS_WAVES: waves per workgroup (0: by the length of the program; 1: one serial program)

    ~ParsInstance() { destroy(); }
    int chunk() const { return width == 16 ? 2 : width == 8 ? 4 : 8; }

    int create(int setCount, int patterns, int wordsPerSet, int setBits, int dev)
    {
        device = dev;
        nSets = setCount;
        P = patterns;
        Ppad = round_up(patterns, 64);
        words = wordsPerSet;
        bits = setBits;
        width = words == 2 ? 16 : bits <= 8 ? 1 : bits <= 16 ? 2 : bits <= 32 ? 4 : 8;
        if (((size_t) nSets + 2) * (size_t) Ppad >= ((size_t) 1 << 32))
            return fail(BEAGLE_ERROR_OUT_OF_RANGE, "mbamdParsCreateInstance", "more than 2^32 (set, pattern) pairs");
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        live = true;
        const size_t rowBytes = (size_t) Ppad * width, bytes = (size_t) (nSets + 2) * rowBytes;
        HIP_TRY(hipMalloc(&d_sets, bytes));
        HIP_TRY(hipMemsetAsync(d_sets, 0, bytes - rowBytes, stream));      // SafeCalloc'ed in the reference (src/mcmc.c:6892)
        HIP_TRY(hipMemsetAsync(static_cast<char*>(d_sets) + bytes - rowBytes, 0xFF, rowBytes, stream));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipMalloc(&d_wbuf[i], (size_t) Ppad * sizeof(float)));
            HIP_TRY(hipMemsetAsync(d_wbuf[i], 0, (size_t) Ppad * sizeof(float), stream));
            HIP_TRY(hipHostMalloc((void**) &h_wpin[i], (size_t) Ppad * sizeof(float), hipHostMallocDefault));
            HIP_TRY(hipEventCreate(&wDone[i]));
        }
        d_w = d_wbuf[0];
        HIP_TRY(hipHostMalloc(&h_stage, (size_t) Ppad * 16, hipHostMallocDefault));
        HIP_TRY(hipStreamSynchronize(stream));
        state.assign((size_t) nSets + 2, SetState());
        verbose = std::getenv("MBAMD_VERBOSE") != nullptr;
        if (std::getenv("MBAMD_NO_POLL") == nullptr && hipHostMalloc((void**) &h_flag, 64, hipHostMallocDefault) == hipSuccess &&
            hipHostGetDevicePointer((void**) &h_flag_dev, h_flag, 0) == hipSuccess) {
            *h_flag = 0;
            pollSums = std::getenv("MBAMD_NO_SUM_POLL") == nullptr;
            poll = true;
        } else {
            (void) hipGetLastError();
        }
        if (const char* e = std::getenv("MBAMD_PARS_PHASE_LIMIT")) phaseLimit = std::max(1, std::min(MBAMD_PARS_MAXPHASES, std::atoi(e)));
        if (const char* e = std::getenv("MBAMD_PARS_WAVES")) waves = std::max(1, std::min(MBAMD_PARS_MAXW, std::atoi(e)));
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
            if (s.haveEvent) (void) hipEventDestroy(s.done);
            s = Slot();
        }
        if (d_sets) (void) hipFree(d_sets);
        for (int i = 0; i < 2; ++i) {
            if (d_wbuf[i]) { (void) hipFree(d_wbuf[i]); (void) hipEventDestroy(wDone[i]); }
            if (h_wpin[i]) (void) hipHostFree(h_wpin[i]);
            d_wbuf[i] = nullptr; h_wpin[i] = nullptr; wBusy[i] = false;
        }
        if (h_out) (void) hipHostFree(h_out);
        if (h_flag) (void) hipHostFree(h_flag);
        if (h_stage) (void) hipHostFree(h_stage);
        (void) hipStreamDestroy(stream);
        d_sets = nullptr; d_w = nullptr; d_out = nullptr; h_out = nullptr; h_stage = nullptr; h_flag = nullptr; h_flag_dev = nullptr; poll = false; outCap = 0;
        live = false;
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
        rc = flush(nullptr);
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
        rc = flush(nullptr);
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
        // Stream-ordered, no wait: the parsimony moves re-randomise the weights for every proposal (reference
        // src/proposal.c:10655-10667).  Passes queued so far do not read the weights (only the down pass's length sum and the
        // score kernel do, at launch time), so they need not run first: the copy goes into the buffer no launch in flight reads.
        const int i = wNext;
        wNext ^= 1;
        if (wBusy[i]) { HIP_TRY(hipEventSynchronize(wDone[i])); wBusy[i] = false; }   // (the copy from two calls ago: long done)
        h_w.assign(w, w + P);
        std::memcpy(h_wpin[i], w, (size_t) P * sizeof(float));
        int rc = flush(nullptr);                     // (queued down passes sum with the weights they were queued under)
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(d_wbuf[i], h_wpin[i], (size_t) P * sizeof(float), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(wDone[i], stream));
        wBusy[i] = true;
        d_w = d_wbuf[i];
        return BEAGLE_SUCCESS;
    }

    // copy `bytes` of host data into the next ring slot (pinned host -> device, stream-ordered); `direct`: no device copy -- the
    // kernel reads the pinned slot over the host link (the few bytes per workgroup of a candidate list: one stream operation less)
    int stage(const void* src, size_t bytes, const void** out, Slot** used, bool direct = false)
    {
        Slot& s = ring[next];
        next = (next + 1) % RING;
        if (s.busy) {
            HIP_TRY(hipEventSynchronize(s.done));
            s.busy = false;
        }
        if (s.cap < bytes) {
            if (s.h) (void) hipHostFree(s.h);
            if (s.d) (void) hipFree(s.d);
            s.h = nullptr; s.d = nullptr;
            s.cap = std::max(2 * bytes, (size_t) 65536);
            HIP_TRY(hipHostMalloc(&s.h, s.cap, hipHostMallocDefault));
            HIP_TRY(hipMalloc(&s.d, s.cap));
            HIP_TRY(hipHostGetDevicePointer(&s.hdev, s.h, 0));
        }
        if (!s.haveEvent) {
            HIP_TRY(hipEventCreate(&s.done));
            s.haveEvent = true;
        }
        std::memcpy(s.h, src, bytes);
        if (!direct) {
            // (round 6: a copy kernel of ours instead of hipMemcpyAsync -- ~10 us of host time per call, profiles/r06_ring_copy.txt)
            static const bool noRingCopy = std::getenv("MBAMD_NO_RING_COPY") != nullptr;
            if (noRingCopy) HIP_TRY(hipMemcpyAsync(s.d, s.h, bytes, hipMemcpyHostToDevice, stream));
            else {
                const unsigned n4 = (unsigned) ((bytes + 3) / 4);
                MBAMD_LAUNCH(k_copy_from_ring4, (n4 + 255u) / 256u, 256, 0, stream, static_cast<const unsigned*>(s.hdev), static_cast<unsigned*>(s.d), n4);
                HIP_TRY(hipGetLastError());
            }
        }
        *out = direct ? s.hdev : s.d;
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
        if (h_out) (void) hipHostFree(h_out);
        d_out = nullptr; h_out = nullptr;
        outCap = std::max(doubles * 2, (size_t) 4096);
        HIP_TRY(hipHostMalloc((void**) &h_out, outCap * sizeof(double), hipHostMallocDefault));
        HIP_TRY(hipHostGetDevicePointer((void**) &d_out, h_out, 0));
        return BEAGLE_SUCCESS;
    }
    // Round 6: the sums are their own completion signal (as the likelihood engine's block sums, mbamd_engine.cpp armSums): the host
    // fills the `count` sums the next launch writes with a bit pattern no sum has, and waits until all of them have changed -- no
    // stream operation behind the kernel.  MBAMD_NO_SUM_POLL=1: the stream's flag (A/B).
    static constexpr uint64_t kSumSentinel = 0x7FF4DEADBEEF0001ull;
    size_t armed = 0;
    void armSums(size_t count)
    {
        armed = (poll && pollSums) ? count : 0;
        uint64_t* p = reinterpret_cast<uint64_t*>(h_out);
        for (size_t i = 0; i < armed; ++i) p[i] = kSumSentinel;
        __atomic_thread_fence(__ATOMIC_RELEASE);
    }
    // the kernels queued so far have written their sums into h_out when this returns
    int waitForSums()
    {
        StatTimer st_(ST_PARS_WAIT);
        bool landed = false;
        if (armed) {
            const volatile uint64_t* p = reinterpret_cast<const volatile uint64_t*>(h_out);
            const auto t0 = std::chrono::steady_clock::now();
            size_t i = 0;
            for (long spins = 0; i < armed; ++spins) {
                if (p[i] != kSumSentinel) { ++i; continue; }
                if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
            }
            landed = i == armed;
            armed = 0;
            if (landed) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return BEAGLE_SUCCESS; }
        }
        if (poll) {
            if (hipStreamWriteValue32(stream, h_flag_dev, ++flagSeq, 0) != hipSuccess) {
                (void) hipGetLastError();
                poll = false;
            } else {
                volatile uint32_t* f = h_flag;
                const auto t0 = std::chrono::steady_clock::now();
                for (long spins = 0; !(landed = (*f == flagSeq)); ++spins) {
                    if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
#if defined(__x86_64__) || defined(__i386__)
                    __builtin_ia32_pause();
#endif
                }
                if (landed) __atomic_thread_fence(__ATOMIC_ACQUIRE);
            }
        }
        if (!landed) HIP_TRY(hipStreamSynchronize(stream));
        return BEAGLE_SUCCESS;
    }

    template <class T, int CH> void launchWalkT(const int* prog, int nphases, int W, double* partial)
    {
        auto kernel = k_pars_walk<T, CH>;
        MBAMD_LAUNCH_BARRIER(kernel, (unsigned) (Ppad / 64), 64 * W, (size_t) W * 2 * CH * 64 * sizeof(typename ParsWide<T>::type), stream, prog, nphases,
                             static_cast<T*>(d_sets), (unsigned) Ppad, d_w, partial);
    }
    template <class T> void launchScore(const ParsOp* tuples, int n)
    {
        MBAMD_LAUNCH_BARRIER(k_pars_score<T>, dim3((unsigned) n, SCORE_Y), MBAMD_PARS_SCORE_THREADS, MBAMD_PARS_SCORE_THREADS * sizeof(double), stream, tuples,
                             static_cast<const T*>(d_sets), (size_t) Ppad, Ppad, d_w, d_out);
    }
    void launchWalk(const int* prog, int nphases, int W, double* partial)
    {
        switch (width) {
            case 1: launchWalkT<uint8_t, 8>(prog, nphases, W, partial); break;
            case 2: launchWalkT<uint16_t, 8>(prog, nphases, W, partial); break;
            case 4: launchWalkT<uint32_t, 8>(prog, nphases, W, partial); break;
            case 8: launchWalkT<uint64_t, 4>(prog, nphases, W, partial); break;
            default: launchWalkT<u128, 2>(prog, nphases, W, partial); break;
        }
    }

    // validate and queue
    int enqueue(int kind, const int* ops, int n, const char* what)
    {
        for (int i = 0; i < n; ++i) {
            const int* o = ops + 4 * i;
            for (int j = 0; j < 4; ++j) {
                int rc = checkIndex(o[j], kind == PARS_DOWN && j == 3, what);      // (the fourth field of a down-pass operation is unused)
                if (rc) return rc;
            }
        }
        ++passCounter;
        for (int i = 0; i < n; ++i) pending.push_back({kind, ops[4 * i], ops[4 * i + 1], ops[4 * i + 2], ops[4 * i + 3], i, n, passCounter});
        return BEAGLE_SUCCESS;
    }

    // Compile the queued steps and run them as one launch.
    // (1) Every step gets a (phase, wave): steps of one (phase, wave) run in program order on that wave; a step may only depend on
    //     steps of earlier phases or on earlier steps of its own (phase, wave).  Dependencies: the writers of its sources, the
    //     previous writer of its destination, and every earlier reader of its destination's old value.  A step without
    //     dependencies starts in phase 0 on the wave of its stripe (a pass's post-order / pre-order cut into W contiguous pieces:
    //     whole subtrees, mostly); a step whose latest dependencies sit on ONE wave stays there -- unless it is a final-pass step
    //     whose own subtree lives elsewhere: that one goes home, one phase later, and takes its descendants with it --; a step
    //     whose latest dependencies sit on several waves opens the next phase.
    // (2) Ring codes per (phase, wave) program: an operand written by a step of this or the previous chunk of the SAME program is
    //     taken from the LDS ring, anything older (or from another phase) was stored before the chunk's prefetch is issued.
    int flush(double* outLength)
    {
        if (pending.empty()) {
            if (outLength) *outLength = 0.0;
            return BEAGLE_SUCCESS;
        }
        const int CH = chunk(), NSLOT = 2 * CH;
        const ParsStep nop{PARS_DOWN, nSets, {nSets + 1, nSets + 1, nSets + 1, nSets + 1}, 0xFFFFFFFFu, 0};
        const int blocks = Ppad / 64;
        if (outLength) {
            int rc = growOut((size_t) blocks);
            if (rc) return rc;
        }
        size_t done = 0;
        double lengthSum = 0.0;
        while (done < pending.size()) {
            StatTimer compileTimer(ST_PARS_COMPILE);
            // ---- (1) phases and waves of as many steps as fit into MBAMD_PARS_MAXPHASES.  Per set: who wrote it last (step, phase,
            // wave) and the latest phase in which it was read since (and by which wave, or by several) -- stamped per launch, so
            // nothing is cleared.
            const int n = (int) (pending.size() - done);
            const int W = waves > 0 ? waves : (n >= 96 ? 8 : n >= 32 ? 4 : 1);
            // Where a step WITHOUT dependencies starts: down passes that are here from their first step are cut into subtrees of about
            // 1 / W of the pass (the largest subtree is split at its root until none is larger), the subtrees dealt to the waves
            // largest first; everything else falls back on stripes of the pass's order.
            binWave.assign((size_t) n, -1);
            const int launchStamp = ++stamp;
            if (W > 1)
                for (int s0 = 0; s0 < n;) {
                    const Pending& q0 = pending[done + (size_t) s0];
                    const int L = q0.passLen;
                    if (q0.kind != PARS_DOWN || q0.passPos != 0 || L < 4 * W || s0 + L > n) { ++s0; continue; }
                    ++stamp;
                    sizeOf.assign((size_t) L, 1);
                    kid1.assign((size_t) L, -1);
                    kid2.assign((size_t) L, -1);
                    isRoot.assign((size_t) L, 1);
                    for (int i = 0; i < L; ++i) {
                        const Pending& q = pending[done + (size_t) (s0 + i)];
                        const int so[2] = {q.b, q.c};
                        for (int jx = 0; jx < 2; ++jx) {
                            const SetState& t = state[(size_t) so[jx]];
                            if (t.stamp != stamp) continue;
                            (jx ? kid2 : kid1)[(size_t) i] = t.wPos;
                            sizeOf[(size_t) i] += sizeOf[(size_t) t.wPos];
                            isRoot[(size_t) t.wPos] = 0;
                        }
                        SetState& t = state[(size_t) q.a];
                        t.stamp = stamp;
                        t.wPos = i;
                    }
                    const int target = (L + W - 1) / W;
                    heap.clear();
                    for (int i = 0; i < L; ++i) if (isRoot[(size_t) i]) heap.push_back({sizeOf[(size_t) i], i});
                    std::make_heap(heap.begin(), heap.end());
                    while (!heap.empty() && heap.front().first > target) {       // split the largest subtree at its root (the root joins the top)
                        std::pop_heap(heap.begin(), heap.end());
                        const int r = heap.back().second;
                        heap.pop_back();
                        for (int k : {kid1[(size_t) r], kid2[(size_t) r]})
                            if (k >= 0) { heap.push_back({sizeOf[(size_t) k], k}); std::push_heap(heap.begin(), heap.end()); }
                    }
                    std::sort(heap.begin(), heap.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
                    int load[MBAMD_PARS_MAXW] = {0};
                    for (const auto& bin : heap) {
                        int best = 0;
                        for (int wq = 1; wq < W; ++wq) if (load[wq] < load[best]) best = wq;
                        load[best] += bin.first;
                        binWave[(size_t) (s0 + bin.second)] = best;
                    }
                    for (int i = L - 1; i >= 0; --i) {                           // a subtree's steps inherit its root's wave
                        const int bwv = binWave[(size_t) (s0 + i)];
                        SetState& t = state[(size_t) pending[done + (size_t) (s0 + i)].a];
                        t.binStamp = launchStamp;
                        t.bin = bwv;
                        if (bwv < 0) continue;
                        if (kid1[(size_t) i] >= 0) binWave[(size_t) (s0 + kid1[(size_t) i])] = bwv;
                        if (kid2[(size_t) i] >= 0) binWave[(size_t) (s0 + kid2[(size_t) i])] = bwv;
                    }
                    s0 += L;
                }
            // A final pass over a tree whose down pass was just cut: its steps for the top of the tree first, then the subtrees'
            // (a valid order: a final step depends on its ancestors' only) -- all subtrees then start in ONE phase behind the top.
            if (W > 1)
                for (int s0 = 0; s0 < n;) {
                    const Pending& q0 = pending[done + (size_t) s0];
                    const int L = q0.passLen;
                    if (q0.kind != PARS_FINAL || q0.passPos != 0 || s0 + L > n) { ++s0; continue; }
                    auto inTop = [&](const Pending& q) { const SetState& t = state[(size_t) q.a]; return !(t.binStamp == launchStamp && t.bin >= 0); };
                    // ... valid only if nothing of "the top" hangs below a subtree: a step outside the bins whose ANCESTOR's step is in
                    // one (or below one) would be moved in front of the step that makes the final set it reads.  MrBayes' passes cover
                    // whole trees (never so); through the public API a partial down pass followed by a full final pass can: keep the
                    // caller's order then.
                    bool movable = true;
                    below.assign(state.size(), 0);
                    for (int i = 0; i < L && movable; ++i) {
                        const Pending& q = pending[done + (size_t) (s0 + i)];
                        const bool under = q.d >= 0 && below[(size_t) q.d];
                        if (inTop(q) && under) movable = false;
                        below[(size_t) q.a] = (!inTop(q) || under) ? 1 : 0;
                    }
                    if (movable)
                        std::stable_partition(pending.begin() + (long) (done + (size_t) s0), pending.begin() + (long) (done + (size_t) (s0 + L)), inTop);
                    s0 += L;
                }
            ++stamp;
            phaseOf.resize((size_t) n);
            waveOf.resize((size_t) n);
            int taken = 0, nphases = 0, topPass = -1, topMax = -1;
            for (; taken < n; ++taken) {
                const Pending& q = pending[done + (size_t) taken];
                const int src[4] = {q.kind == PARS_DOWN ? q.b : q.a, q.kind == PARS_DOWN ? q.c : q.b, q.kind == PARS_DOWN ? -1 : q.c,
                                    q.kind == PARS_DOWN ? -1 : q.d};
                int P = -1, wv = -1, home = -1;
                bool several = false;
                auto dep = [&](int ph, int wq) {
                    if (ph > P) { P = ph; wv = wq; several = wq < 0; }
                    else if (ph == P && wq != wv) several = true;
                };
                for (int jx = 0; jx < 4; ++jx) {
                    if (src[jx] < 0) continue;
                    const SetState& t = state[(size_t) src[jx]];
                    if (t.stamp != stamp || t.wPhase < 0) continue;
                    dep(t.wPhase, t.wWave);
                    if (jx == 0) home = t.wWave;
                }
                {
                    const SetState& t = state[(size_t) q.a];
                    if (t.stamp == stamp) {
                        if (t.wPhase >= 0) dep(t.wPhase, t.wWave);
                        if (t.rPhase >= 0) dep(t.rPhase, t.rWave);
                    }
                }
                const int passLen = q.passLen > 0 ? q.passLen : 1;
                if (home < 0) home = binWave[(size_t) taken] >= 0 ? binWave[(size_t) taken] : (passLen >= 4 * W) ? (int) ((long) q.passPos * W / passLen) : 0;
                int ph, wq;
                if (P < 0) { ph = 0; wq = home; }
                else if (several) { ph = P + 1; wq = home; }
                else if (q.kind == PARS_FINAL && wv != home) { ph = P + 1; wq = home; }
                else { ph = P; wq = wv; }
                if (q.kind == PARS_FINAL && W > 1) {
                    const SetState& t = state[(size_t) q.a];
                    const bool inBin = t.binStamp == launchStamp && t.bin >= 0;
                    if (q.passId != topPass) { topPass = q.passId; topMax = -1; }
                    if (!inBin) topMax = std::max(topMax, ph);
                    else if (ph <= topMax) { ph = topMax + 1; wq = home; }       // (later than it must: with the other subtrees)
                }
                if (ph >= phaseLimit) break;                                 // the rest in another launch
                phaseOf[(size_t) taken] = ph;
                waveOf[(size_t) taken] = wq;
                nphases = std::max(nphases, ph + 1);
                for (int jx = 0; jx < 4; ++jx) {
                    if (src[jx] < 0) continue;
                    SetState& t = state[(size_t) src[jx]];
                    if (t.stamp != stamp) { t.stamp = stamp; t.wPhase = t.rPhase = -1; t.wWave = t.rWave = -1; }
                    if (ph > t.rPhase) { t.rPhase = ph; t.rWave = wq; }
                    else if (ph == t.rPhase && t.rWave != wq) t.rWave = -1;             // (-1: several waves)
                }
                SetState& t = state[(size_t) q.a];
                t.stamp = stamp;
                t.wPhase = ph; t.wWave = wq;
                t.rPhase = -1; t.rWave = -1;
            }
            // ---- (2) one program per (phase, wave): the steps bucketed in program order
            const int nbuckets = nphases * W;
            bucketStart.assign((size_t) nbuckets + 1, 0);
            for (int i = 0; i < taken; ++i) bucketStart[(size_t) (phaseOf[(size_t) i] * W + waveOf[(size_t) i]) + 1]++;
            for (int b = 0; b < nbuckets; ++b) bucketStart[(size_t) b + 1] += bucketStart[(size_t) b];
            bucketFill.assign(bucketStart.begin(), bucketStart.end() - 1);
            order.resize((size_t) taken);
            for (int i = 0; i < taken; ++i) order[(size_t) bucketFill[(size_t) (phaseOf[(size_t) i] * W + waveOf[(size_t) i])]++] = i;
            header.assign((size_t) MBAMD_PARS_HEADER_INTS, 0);
            compiled.clear();
            for (int ph = 0; ph < nphases; ++ph)
                for (int wq = 0; wq < W; ++wq) {
                    const int b = ph * W + wq;
                    if (bucketStart[(size_t) b] == bucketStart[(size_t) b + 1]) continue;
                    const size_t first = compiled.size();
                    ++stamp;                                                    // (positions of this program's writers)
                    auto padTo = [&](int kind) {                                // no-ops up to the next chunk boundary
                        while ((compiled.size() - first) % (size_t) CH) {
                            compiled.push_back(nop);
                            compiled.back().kind = kind;
                        }
                    };
                    for (int k = bucketStart[(size_t) b]; k < bucketStart[(size_t) b + 1]; ++k) {
                        const Pending& q = pending[done + (size_t) order[(size_t) k]];
                        if (compiled.size() > first && compiled.back().kind != q.kind) padTo(compiled.back().kind);
                        const int pos = (int) (compiled.size() - first);
                        ParsStep st = nop;
                        const int src[4] = {q.kind == PARS_DOWN ? q.b : q.a, q.kind == PARS_DOWN ? q.c : q.b, q.kind == PARS_DOWN ? -1 : q.c,
                                            q.kind == PARS_DOWN ? -1 : q.d};
                        const int windowStart = (pos / CH) * CH - CH;
                        st.kind = q.kind;
                        st.dest = q.a;
                        st.ring = 0;
                        for (int jx = 0; jx < 4; ++jx) {
                            unsigned code = 0xFFu;
                            if (src[jx] >= 0) {
                                const SetState& t = state[(size_t) src[jx]];
                                if (t.stamp == stamp && t.wPos >= windowStart) code = (unsigned) (t.wPos % NSLOT);
                                else st.g[jx] = src[jx];
                            }
                            st.ring |= code << (8 * jx);
                        }
                        compiled.push_back(st);
                        SetState& t = state[(size_t) q.a];
                        t.stamp = stamp;
                        t.wPos = pos;
                    }
                    padTo(compiled.back().kind);
                    int nchunks = (int) ((compiled.size() - first) / (size_t) CH);
                    if (nchunks & 1) {
                        for (int k = 0; k < CH; ++k) compiled.push_back(nop);
                        ++nchunks;
                    }
                    for (int k = 0; k < 2 * CH; ++k) compiled.push_back(nop);    // two more chunks of no-ops: fetched ahead, never run
                    header[(size_t) (ph * MBAMD_PARS_MAXW + wq) * 2] = (int) (first / (size_t) CH);
                    header[(size_t) (ph * MBAMD_PARS_MAXW + wq) * 2 + 1] = nchunks;
                }
            if (verbose) {
                std::fprintf(stderr, "[mbamd] parsimony program: %d steps, %d waves, %d phases; chunks per wave:", taken, W, nphases);
                for (int ph = 0; ph < nphases; ++ph) {
                    std::fprintf(stderr, " |");
                    for (int wq = 0; wq < W; ++wq) std::fprintf(stderr, " %d", header[(size_t) (ph * MBAMD_PARS_MAXW + wq) * 2 + 1]);
                }
                std::fprintf(stderr, "\n");
            }
            done += (size_t) taken;
            const bool last = done >= pending.size();
            const size_t bytes = header.size() * sizeof(int) + compiled.size() * sizeof(ParsStep);
            image.resize(bytes);
            std::memcpy(image.data(), header.data(), header.size() * sizeof(int));
            std::memcpy(image.data() + header.size() * sizeof(int), compiled.data(), compiled.size() * sizeof(ParsStep));
            compileTimer.stop();
            const void* d = nullptr;
            Slot* sl = nullptr;
            int rc = stage(image.data(), bytes, &d, &sl);
            if (rc) return rc;
            // (a length is asked for a pass queued alone -- downPass() flushes what came before it; should the pass need more
            //  phases than one launch holds, every launch's sums are fetched and added)
            if (outLength) armSums((size_t) blocks);
            launchWalk(static_cast<const int*>(d), nphases, W, outLength ? d_out : nullptr);
            HIP_TRY(hipGetLastError());
            rc = release(sl);
            if (rc) return rc;
            if (outLength) {
                rc = waitForSums();
                if (rc) return rc;
                for (int b = 0; b < blocks; ++b) lengthSum += h_out[b];
            }
            (void) last;
        }
        pending.clear();
        if (outLength) *outLength = lengthSum;
        return BEAGLE_SUCCESS;
    }

    int downPass(const int* ops, int n, double* outLength)
    {
        if (!outLength) return enqueue(PARS_DOWN, ops, n, "mbamdParsDownPass");
        int rc = flush(nullptr);                     // the length asked for is this pass's alone
        if (rc) return rc;
        rc = enqueue(PARS_DOWN, ops, n, "mbamdParsDownPass");
        if (rc) return rc;
        return flush(outLength);
    }

    int finalPass(const int* ops, int n) { return enqueue(PARS_FINAL, ops, n, "mbamdParsFinalPass"); }

    int score(const int* tuples, int n, double* out)
    {
        for (int i = 0; i < 4 * n; ++i) {
            int rc = checkIndex(tuples[i], true, "mbamdParsScore");
            if (rc) return rc;
        }
        int rc = flush(nullptr);
        if (rc) return rc;
        if (n <= 0) return BEAGLE_SUCCESS;
        rc = growOut((size_t) n * SCORE_Y);
        if (rc) return rc;
        const void* d = nullptr;
        Slot* s = nullptr;
        rc = stage(tuples, (size_t) n * sizeof(ParsOp), &d, &s, true);
        if (rc) return rc;
        armSums((size_t) n * SCORE_Y);
        switch (width) {
            case 1: launchScore<uint8_t>(static_cast<const ParsOp*>(d), n); break;
            case 2: launchScore<uint16_t>(static_cast<const ParsOp*>(d), n); break;
            case 4: launchScore<uint32_t>(static_cast<const ParsOp*>(d), n); break;
            case 8: launchScore<uint64_t>(static_cast<const ParsOp*>(d), n); break;
            default: launchScore<u128>(static_cast<const ParsOp*>(d), n); break;
        }
        HIP_TRY(hipGetLastError());
        rc = release(s);
        if (rc) return rc;
        rc = waitForSums();
        if (rc) return rc;
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
