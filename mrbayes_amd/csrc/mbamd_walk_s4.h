// mbamd_walk_s4.h -- the 4-state tree-walk kernel (included by mbamd_kernels.h).
//
// Replaces CondLikeDown_NUC4* / CondLikeRoot_NUC4* + CondLikeScaler_NUC4* + RemoveNodeScalers
// (reference src/likelihood.c:786-1570, 2953-4000, 5137-5410, 7981-8070) for a whole operation list
// in ONE launch.
//
// One workgroup owns 64 site patterns (lane = pattern, all K categories in registers) for the whole
// list: site patterns are independent through the entire pruning recursion, so there is no
// inter-workgroup dependency.  The host compiles the list into a schedule of steps: step s holds up
// to W mutually independent operations, one per COMPUTE wave, whose inputs were produced in earlier
// steps; a workgroup barrier (LDS-only: s_waitcnt lgkmcnt(0) + s_barrier) separates steps.
//
//   * A result that is consumed later in the list is kept in an LDS slot chosen by the host (Belady
//     eviction), so a parent reads its children back from LDS: HBM never serves a re-read of a
//     value produced in the same launch, and sees every node as one 4K KiB contiguous write
//     (block-major arena).
//   * Compute waves issue NO vector loads.  The vector-memory counter (vmcnt) retires in order, so
//     a wave that both stores and loads must drain its stores before it can consume a younger
//     load.  One extra LOADER wave per workgroup therefore fetches everything the next step needs
//     -- the W descriptors, their 2K transposed 4x4 matrices, tip state codes, stored scale
//     exponents, children that live in global memory -- one step ahead and puts it into LDS; the
//     compute waves only read LDS, do the arithmetic and fire their stores without ever waiting.
//   * Matrix elements reach the FMAs through DPP quad broadcasts (see Mat4): 4 VGPRs per matrix.
//   * The per-pattern max-rescale (the reference's separate CondLikeScaler pass) is fused in; the
//     factor is the power of two 2^-e (top of mbamd_kernels.h) and the cumulative buffer gets
//     the per-lane sum of e with one integer atomic per compute wave at the end.
//
// LDS map (16-byte units): [2 parities][W] step inputs (descriptor 4 | matrices 8K | tips 8 |
// exponents 16), [slots][64K] values, 28 units of descriptor staging for the loader.  A child that lives in global memory (a buffer not
// produced by this list, or a value evicted from LDS, CHILD_RELOAD) gets a slot for one step: the
// loader copies it in during the step before its consumer runs.  A re-read of an evicted value is
// legal two steps after it was produced; the host then flags MBAMD_OP_DRAIN so that the compute waves
// wait for their stores before the barrier that precedes the loader's copy.
#ifndef MBAMD_WALK_S4_H_
#define MBAMD_WALK_S4_H_

namespace mbamd {

#define MBAMD_OP_DRAIN 1      // PartialsOp::flags: compute waves drain their stores before this step's barrier
#define MBAMD_OP_HAS_READ 2   // (replicated over the row) some entry of this step divides by stored scale factors
#define MBAMD_OP_HAS_GLOBAL 4 // (replicated over the row) some entry of this step has a child in global memory
#define MBAMD_OP_LOAD 16      // (this entry only) not an operation: copy buffer `dst` (global) into LDS slot dst_slot

__host__ __device__ inline int walk_input_units(int K) { return 4 + 8 * K + 8 + 16; }   // f4 per step-input entry
__host__ __device__ inline int walk_slot_units(int K) { return 64 * K; }               // f4 per value slot
__host__ __device__ inline int walk_lds_units(int K, int W, int slots)
{
    return 2 * W * walk_input_units(K) + slots * walk_slot_units(K) + 7 * 4      // + the loader's descriptor staging
           + W * 2 * 16;                                                          // + per-entry max exchange (category split)
}
#define MBAMD_WALK_MAXW 7

__device__ __forceinline__ void walk_step_barrier()
{
#if !defined(MBAMD_HOST_EMU)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
}

__device__ __forceinline__ f4 tip_vector(unsigned s)
{
    f4 one;
    one.x = (s == 0u || s >= 4u) ? 1.0f : 0.0f;
    one.y = (s == 1u || s >= 4u) ? 1.0f : 0.0f;
    one.z = (s == 2u || s >= 4u) ? 1.0f : 0.0f;
    one.w = (s == 3u || s >= 4u) ? 1.0f : 0.0f;
    return one;
}

// the fields of a PartialsOp a compute wave needs, in (scalar) registers
struct WalkFields {
    float* dst;
    int32_t* scale;
    int c1_kind, c2_kind, c1_slot, c2_slot, dst_slot, scale_mode, flags;
};

// One operation (or the KN categories [k0, k0+KN) of it), all inputs already in LDS / registers:
//   M1, M2   the lane's rows of the transposed matrices (Mat4)
//   a, b     the children's values per category (tips: the 0/1 vector)
// walk_products: product of the two child factors and its per-pattern maximum;
// walk_store:    rescale by 2^-e, result -> HBM and -> LDS slot (if a parent will read it).
template <int KN>
__device__ __forceinline__ float walk_products(const Mat4 (&M1)[KN], const Mat4 (&M2)[KN], const f4 (&a)[KN], const f4 (&b)[KN],
                                               f4 (&out)[KN])
{
    float mx = 0.0f;
#pragma unroll
    for (int k = 0; k < KN; ++k) {
        const f4 f1 = mat4_mul(M1[k], a[k]);
        const f4 f2 = mat4_mul(M2[k], b[k]);
        out[k].x = f1.x * f2.x;
        out[k].y = f1.y * f2.y;
        out[k].z = f1.z * f2.z;
        out[k].w = f1.w * f2.w;
        mx = fmaxf(mx, max4(out[k]));
    }
    return mx;
}
template <int K, int KN>
__device__ __forceinline__ void walk_store(const WalkFields& op, f4 (&out)[KN], int e, int k0, bool owner, size_t poff, size_t soff,
                                           int lane, f4* slots)
{
    MBAMD_AS_GLOBAL f4* __restrict__ dst = as_global(reinterpret_cast<f4*>(op.dst)) + poff + k0 * 64;
#pragma unroll
    for (int k = 0; k < KN; ++k) {          // multiplying by 2^0 is exact: no branch needed
        out[k].x = scale_pow2(out[k].x, -e);
        out[k].y = scale_pow2(out[k].y, -e);
        out[k].z = scale_pow2(out[k].z, -e);
        out[k].w = scale_pow2(out[k].w, -e);
        dst[k * 64] = out[k];               // 4K KiB contiguous per node update; never waited for
    }
    if (owner && op.scale_mode == SCALE_WRITE) as_global(op.scale)[soff] = e;
    if (op.dst_slot != MBAMD_NO_SLOT) {
        f4* slot = slots + op.dst_slot * walk_slot_units(K) + k0 * 64 + lane;
#pragma unroll
        for (int k = 0; k < KN; ++k) slot[k * 64] = out[k];
    }
}
template <int K>
__device__ __forceinline__ void walk_math(const WalkFields& op, const Mat4 (&M1)[K], const Mat4 (&M2)[K], const f4 (&a)[K],
                                          const f4 (&b)[K], int e_read, size_t poff, size_t soff, int lane, f4* slots,
                                          int& cum_e)
{
    f4 out[K];
    const float mx = walk_products<K>(M1, M2, a, b, out);
    int e = 0;
    if (op.scale_mode == SCALE_WRITE) { e = scale_exponent(mx); cum_e += e; }
    else if (op.scale_mode == SCALE_READ) e = e_read;
    walk_store<K, K>(op, out, e, 0, true, poff, soff, lane, slots);
}

#if !defined(MBAMD_HOST_EMU)
// ---- loader wave ----------------------------------------------------------------------------------
// One row of the schedule = the W descriptors of a step = 16*W dwords, fetched with two coalesced
// dword loads (lane l holds dwords l and l+64); single fields are picked out with v_readlane.
struct WalkRow { unsigned lo, hi; };
__device__ __forceinline__ WalkRow walk_row_request(const PartialsOp* ops, int step, int W, int lane)
{
    const MBAMD_AS_GLOBAL unsigned* p = as_global(reinterpret_cast<const unsigned*>(ops)) + (size_t) step * W * 16;
    WalkRow r;
    r.lo = p[lane < W * 16 ? lane : 0];
    r.hi = p[W > 4 ? 64 + lane : lane & 15];
    return r;
}
__device__ __forceinline__ unsigned walk_row_dword(const WalkRow& r, int w, int i)
{
    return (unsigned) __builtin_amdgcn_readlane((int) ((w >= 4) ? r.hi : r.lo), (w & 3) * 16 + i);
}
__device__ __forceinline__ const void* walk_row_ptr(const WalkRow& r, int w, int i)
{
    return reinterpret_cast<const void*>(((unsigned long) walk_row_dword(r, w, i + 1) << 32) | walk_row_dword(r, w, i));
}

// What the loader fetches for a whole step, spread over its 64 lanes (20 VGPRs while in flight):
//   matrices   W*8K rows of 16 bytes: row index i = lane + 64 j  ->  entry i / 8K, row i % 8K (first 4K: m1)
//   tips       2W children x 64 state codes = 2W*16 dwords: index i = lane + 64 j -> child i / 16, dword i % 16
// Per-lane source addresses come from the descriptor row, staged in a small LDS area first.
#define MBAMD_WALK_FETCH 4          // ceil(7*32/64) row loads, ceil(14*16/64) tip loads
struct WalkFetch {
    f4 mrow[MBAMD_WALK_FETCH];
    unsigned tips[MBAMD_WALK_FETCH];
};
// fdesc: LDS staging of one descriptor row ([W][16] dwords), private to the loader wave
template <int K>
__device__ __forceinline__ void walk_fetch_issue(const WalkRow& row, int W, unsigned* fdesc, size_t tblock, int lane,
                                                 WalkFetch& f)
{
    if (lane < W * 16 && lane < 64) fdesc[lane] = row.lo;
    if (W > 4 && lane < (W - 4) * 16) fdesc[64 + lane] = row.hi;
    // (LDS operations of one wave execute in order: the reads below see the row)
#pragma unroll
    for (int j = 0; j < MBAMD_WALK_FETCH; ++j) {
        int i = lane + 64 * j;
        if (i >= W * 8 * K) i = 0;                                   // surplus lanes repeat row 0 (straight-line loads)
        const int w = i / (8 * K), r = i % (8 * K);
        const bool first = r < 4 * K;
        const unsigned long base = *reinterpret_cast<const unsigned long*>(fdesc + w * 16 + (first ? 6 : 8));
        f.mrow[j] = as_global(reinterpret_cast<const f4*>(base))[first ? r : r - 4 * K];
    }
#pragma unroll
    for (int j = 0; j < MBAMD_WALK_FETCH; ++j) {
        int i = lane + 64 * j;
        if (i >= 2 * W * 16) i = 0;
        const int q = i >> 4, d = i & 15;                            // child q = 2*entry + (0: c1, 1: c2)
        const unsigned long base = *reinterpret_cast<const unsigned long*>(fdesc + (q >> 1) * 16 + 2 + 2 * (q & 1));
        f.tips[j] = as_global(reinterpret_cast<const unsigned*>(base + tblock))[d];
    }
}
// fetched data + the descriptor row -> the step-input area of LDS
template <int K>
__device__ __forceinline__ void walk_fetch_commit(const WalkRow& row, int W, const WalkFetch& f, int lane, f4* inputs)
{
    const int IU = walk_input_units(K);
    unsigned* in32 = reinterpret_cast<unsigned*>(inputs);
    if (lane < W * 16 && lane < 64) in32[(lane >> 4) * IU * 4 + (lane & 15)] = row.lo;      // descriptors
    if (W > 4 && lane < (W - 4) * 16) in32[(4 + (lane >> 4)) * IU * 4 + (lane & 15)] = row.hi;
#pragma unroll
    for (int j = 0; j < MBAMD_WALK_FETCH; ++j) {
        const int i = lane + 64 * j;
        if (i < W * 8 * K) inputs[(i / (8 * K)) * IU + 4 + (i % (8 * K))] = f.mrow[j];
    }
#pragma unroll
    for (int j = 0; j < MBAMD_WALK_FETCH; ++j) {
        const int i = lane + 64 * j;
        if (i < 2 * W * 16) {
            const int q = i >> 4, d = i & 15;
            in32[((q >> 1) * IU + 4 + 8 * K + 4 * (q & 1)) * 4 + d] = f.tips[j];
        }
    }
}
// stored scale exponents of the entries that divide by them (SCALE_READ; dynamic scaling's no-rescale pass)
template <int K>
__device__ __forceinline__ void walk_load_exponents(const WalkRow& row, int W, size_t soff, int lane, f4* inputs)
{
#pragma unroll 1
    for (int w = 0; w < W; ++w) {
        const unsigned hi = walk_row_dword(row, w, 13);
        if (((hi >> 8) & 0xFF) != SCALE_READ) continue;
        const int e = as_global(reinterpret_cast<const int32_t*>(walk_row_ptr(row, w, 10)))[soff];
        reinterpret_cast<int*>(inputs + w * walk_input_units(K) + 4 + 8 * K + 8)[lane] = e;
    }
}
// children that live in global memory: copied into their one-step slots (issue, wait, write)
template <int K>
__device__ __forceinline__ void walk_load_global_children(const WalkRow& row, int W, size_t poff, int lane, f4* slots)
{
#pragma unroll 1
    for (int w = 0; w < W; ++w) {
        const unsigned bits = walk_row_dword(row, w, 12);
        if (walk_row_dword(row, w, 0) == 0u && walk_row_dword(row, w, 1) == 0u) continue;   // empty entry
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kind = (bits >> (8 * t)) & 0xFF, slot = (bits >> (16 + 8 * t)) & 0xFF;
            if (kind != CHILD_PARTIALS && kind != CHILD_RELOAD) continue;
            const MBAMD_AS_GLOBAL f4* p = as_global(reinterpret_cast<const f4*>(walk_row_ptr(row, w, 2 + 2 * t))) + poff;
            f4 v[K];
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = p[k * 64];
            f4* sl = slots + slot * walk_slot_units(K) + lane;
#pragma unroll
            for (int k = 0; k < K; ++k) sl[k * 64] = v[k];
        }
    }
}
#endif

// ops: [nsteps + 4][W] (an empty entry has dst == nullptr and valid dummy pointers; flags are
// replicated over a step's entries; four empty rows pad the end for the loader's read-ahead).
// blockDim.x == 64*(W+1): W compute waves + the loader wave.
// (Host emulation: one 64-thread block; each thread runs the W entries of a step one after the
//  other -- lanes never exchange data -- and plays the loader where the GPU's loader would act.)
template <int K>
__global__ void __launch_bounds__(512, (K <= 4 ? 4 : 2))
k_walk_s4(const PartialsOp* __restrict__ ops, int nsteps, int W, int nslots, int ksplit, BlockGeom g,
          int32_t* __restrict__ cumulative, long long* __restrict__ trace)
{
    const int lane = threadIdx.x & 63;
    const size_t poff = (size_t) blockIdx.x * g.pstride + lane;     // this workgroup's block in every buffer
    const size_t soff = (size_t) blockIdx.x * g.sstride + lane;
    int cum_e = 0;
#if defined(MBAMD_HOST_EMU)
    (void) trace; (void) nslots; (void) ksplit;
    const size_t toff = (size_t) blockIdx.x * g.tstride + lane;
    const bool reversed = W < 0;                 // test hook: run a step's entries in the opposite order
    if (reversed) W = -W;
    f4* slots = reinterpret_cast<f4*>(mbamd_emu_dyn_lds()) + 2 * W * walk_input_units(K);
    auto load_children = [&](int s) {            // the loader's copy of global children for step s
        for (int i = 0; i < W; ++i) {
            const PartialsOp* op = ops + (size_t) s * W + i;
            if (op->dst == nullptr) continue;
            const void* cp[2] = {op->c1, op->c2};
            const int ck[2] = {op->c1_kind, op->c2_kind}, cs[2] = {op->c1_slot, op->c2_slot};
            for (int t = 0; t < 2; ++t)
                if (ck[t] == CHILD_PARTIALS || ck[t] == CHILD_RELOAD)
                    for (int k = 0; k < K; ++k)
                        slots[cs[t] * walk_slot_units(K) + k * 64 + lane] = reinterpret_cast<const f4*>(cp[t])[poff + k * 64];
        }
    };
    load_children(0);
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) load_children(s + 1);     // happens while step s computes: before its results land
        for (int i = 0; i < W; ++i) {
            const PartialsOp* op = ops + (size_t) s * W + (reversed ? W - 1 - i : i);
            if (op->dst == nullptr) continue;
            if (op->flags & MBAMD_OP_LOAD) {
                for (int k = 0; k < K; ++k)
                    slots[op->dst_slot * walk_slot_units(K) + k * 64 + lane] = reinterpret_cast<const f4*>(op->dst)[poff + k * 64];
                continue;
            }
            WalkFields f;
            f.dst = op->dst; f.scale = op->scale; f.c1_kind = op->c1_kind; f.c2_kind = op->c2_kind;
            f.c1_slot = op->c1_slot; f.c2_slot = op->c2_slot; f.dst_slot = op->dst_slot;
            f.scale_mode = op->scale_mode; f.flags = op->flags;
            Mat4 M1[K], M2[K];
            f4 a[K], b[K];
            const f4 one1 = tip_vector(f.c1_kind == CHILD_STATES ? reinterpret_cast<const uint8_t*>(op->c1)[toff] : 0u);
            const f4 one2 = tip_vector(f.c2_kind == CHILD_STATES ? reinterpret_cast<const uint8_t*>(op->c2)[toff] : 0u);
            for (int k = 0; k < K; ++k) {
                M1[k] = mat4_load(op->m1 + 16 * k, lane);
                M2[k] = mat4_load(op->m2 + 16 * k, lane);
                a[k] = f.c1_kind == CHILD_STATES ? one1 : slots[f.c1_slot * walk_slot_units(K) + k * 64 + lane];
                b[k] = f.c2_kind == CHILD_STATES ? one2 : slots[f.c2_slot * walk_slot_units(K) + k * 64 + lane];
            }
            const int e_read = f.scale_mode == SCALE_READ ? op->scale[soff] : 0;
            walk_math<K>(f, M1, M2, a, b, e_read, poff, soff, lane, slots, cum_e);
        }
    }
#else
    extern __shared__ f4 lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int IU = walk_input_units(K);
    f4* slots = lds_all + 2 * W * IU;
    // (timing experiments: MBAMD_WALK_TRACE makes block 0 record s_memtime stamps [step][wave][3])
    const bool tracing = trace != nullptr && blockIdx.x == 0 && lane == 0;
#define MBAMD_STAMP(STEP, I) if (tracing) trace[((size_t) (STEP) * 8 + wave) * 3 + (I)] = (long long) __builtin_amdgcn_s_memtime();
    const int CW = W * ksplit;                                       // compute waves: `ksplit` per table entry
    if (wave == CW) {
        // ---- loader wave: inputs of step s+1 go to LDS while the compute waves work on step s ------------
        unsigned* fdesc = reinterpret_cast<unsigned*>(slots + nslots * walk_slot_units(K));   // [7][16] dwords
        const size_t tblock = (size_t) blockIdx.x * g.tstride;
        WalkFetch f;
        WalkRow rowA = walk_row_request(ops, 0, W, lane);            // row of the step being committed next
        WalkRow rowB = walk_row_request(ops, 1, W, lane);            // the one after
        walk_fetch_issue<K>(rowA, W, fdesc, tblock, lane, f);
        walk_fetch_commit<K>(rowA, W, f, lane, lds_all);             // step 0 -> parity 0
        walk_load_exponents<K>(rowA, W, soff, lane, lds_all);
        walk_load_global_children<K>(rowA, W, poff, lane, slots);
        walk_fetch_issue<K>(rowB, W, fdesc, tblock, lane, f);
        WalkRow rowC = walk_row_request(ops, 2, W, lane);
        walk_step_barrier();                                         // barrier(-1): step 0 may start
        for (int s = 0; s < nsteps; ++s) {
            MBAMD_STAMP(s, 0)
            // rowB = row of step s+1 (its fetches were issued one iteration ago), rowC = row of step s+2
            f4* in_next = lds_all + ((s + 1) & 1) * W * IU;
            walk_fetch_commit<K>(rowB, W, f, lane, in_next);
            walk_fetch_issue<K>(rowC, W, fdesc, tblock, lane, f);
            if (walk_row_dword(rowB, 0, 13) & (MBAMD_OP_HAS_READ << 16)) walk_load_exponents<K>(rowB, W, soff, lane, in_next);
            if (walk_row_dword(rowB, 0, 13) & (MBAMD_OP_HAS_GLOBAL << 16)) walk_load_global_children<K>(rowB, W, poff, lane, slots);
            rowB = rowC;
            rowC = walk_row_request(ops, s + 3, W, lane);
            if (ksplit == 2) walk_step_barrier();                    // (the compute waves' max exchange)
            MBAMD_STAMP(s, 1)
            walk_step_barrier();
            MBAMD_STAMP(s, 2)
        }
        return;
    }
    // ---- compute waves: LDS in, arithmetic, stores out; no vector loads -------------------------------
    // ksplit == 2 (even K): two waves share a table entry, each takes half of its categories; the
    // per-pattern maximum is exchanged through LDS at a mid-step barrier.  Halves the serial
    // instruction stream of an operation without needing more LDS slots.
    const int entry = ksplit == 2 ? wave >> 1 : wave;
    const int khalf = ksplit == 2 ? wave & 1 : 0;
    float* mxbuf = reinterpret_cast<float*>(slots + nslots * walk_slot_units(K) + 7 * 4) + (entry * 2) * 64;
    walk_step_barrier();                                             // barrier(-1)
    for (int s = 0; s < nsteps; ++s) {
        MBAMD_STAMP(s, 0)
        const f4* in = lds_all + ((s & 1) * W + entry) * IU;
        const f4 d0 = in[0], d2 = in[2], d3 = in[3];                 // descriptor dwords 0-3, 8-11, 12-15 (broadcast reads)
        // (readfirstlane returns int: go through unsigned before widening, or bit 31 smears into the high half)
        auto sgpr = [](float v) { return (unsigned long) (unsigned) __builtin_amdgcn_readfirstlane((int) __float_as_uint(v)); };
        WalkFields op;
        op.dst = reinterpret_cast<float*>((sgpr(d0.y) << 32) | sgpr(d0.x));
        op.scale = reinterpret_cast<int32_t*>((sgpr(d2.w) << 32) | sgpr(d2.z));
        const unsigned lo = (unsigned) sgpr(d3.x);
        const unsigned hi = (unsigned) sgpr(d3.y);
        op.c1_kind = lo & 0xFF; op.c2_kind = (lo >> 8) & 0xFF; op.c1_slot = (lo >> 16) & 0xFF; op.c2_slot = lo >> 24;
        op.dst_slot = hi & 0xFF; op.scale_mode = (hi >> 8) & 0xFF; op.flags = (hi >> 16) & 0xFF;
        const uint8_t* tips = reinterpret_cast<const uint8_t*>(in + 4 + 8 * K);
        const bool tip1 = op.c1_kind == CHILD_STATES, tip2 = op.c2_kind == CHILD_STATES;
        const f4* l1 = slots + (tip1 ? 0 : op.c1_slot) * walk_slot_units(K) + lane;       // tips read slot 0 and discard it
        const f4* l2 = slots + (tip2 ? 0 : op.c2_slot) * walk_slot_units(K) + lane;
        if (op.flags & MBAMD_OP_LOAD) {
            // an idle compute wave plays prefetcher: a child that lives in global memory (the sibling on a
            // path-to-root update) is copied into its LDS slot one or more steps before its consumer runs
            if (khalf == 0) {
                const MBAMD_AS_GLOBAL f4* p = as_global(reinterpret_cast<const f4*>(op.dst)) + poff;
                f4 v[K];
#pragma unroll
                for (int k = 0; k < K; ++k) v[k] = p[k * 64];
                f4* sl = slots + op.dst_slot * walk_slot_units(K) + lane;
#pragma unroll
                for (int k = 0; k < K; ++k) sl[k * 64] = v[k];
            }
            if (ksplit == 2) walk_step_barrier();
        } else if (ksplit == 2) {
            if constexpr (K % 2 == 0) {
                constexpr int KN = K / 2;
                const int k0 = khalf * KN;
                f4 out[KN];
                float mx = 0.0f;
                if (op.dst != nullptr) {
                    Mat4 M1[KN], M2[KN];
                    f4 a[KN], b[KN];
                    const f4 one1 = tip_vector(tips[lane]), one2 = tip_vector(tips[64 + lane]);
#pragma unroll
                    for (int k = 0; k < KN; ++k) {
                        M1[k].col = in[4 + 4 * (k0 + k) + (lane & 3)];
                        M2[k].col = in[4 + 4 * K + 4 * (k0 + k) + (lane & 3)];
                        const f4 va = l1[(k0 + k) * 64], vb = l2[(k0 + k) * 64];
                        a[k] = tip1 ? one1 : va;
                        b[k] = tip2 ? one2 : vb;
                    }
                    mx = walk_products<KN>(M1, M2, a, b, out);
                    mxbuf[khalf * 64 + lane] = mx;
                }
                walk_step_barrier();                                 // both halves' maxima are in LDS
                if (op.dst != nullptr) {
                    int e = 0;
                    if (op.scale_mode == SCALE_WRITE) {
                        e = scale_exponent(fmaxf(mx, mxbuf[(1 - khalf) * 64 + lane]));
                        if (khalf == 0) cum_e += e;
                    } else if (op.scale_mode == SCALE_READ) {
                        e = reinterpret_cast<const int*>(in + 4 + 8 * K + 8)[lane];
                    }
                    walk_store<K, KN>(op, out, e, k0, khalf == 0, poff, soff, lane, slots);
                }
            }
        } else if (op.dst != nullptr) {
            Mat4 M1[K], M2[K];
            f4 a[K], b[K];
            const f4 one1 = tip_vector(tips[lane]), one2 = tip_vector(tips[64 + lane]);
            const int e_read = reinterpret_cast<const int*>(in + 4 + 8 * K + 8)[lane];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                M1[k].col = in[4 + 4 * k + (lane & 3)];
                M2[k].col = in[4 + 4 * K + 4 * k + (lane & 3)];
                const f4 va = l1[k * 64], vb = l2[k * 64];
                a[k] = tip1 ? one1 : va;
                b[k] = tip2 ? one2 : vb;
            }
            walk_math<K>(op, M1, M2, a, b, e_read, poff, soff, lane, slots, cum_e);
        }
        if (op.flags & MBAMD_OP_DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MBAMD_STAMP(s, 1)
        walk_step_barrier();
        MBAMD_STAMP(s, 2)
    }
#undef MBAMD_STAMP
#endif
    if (cumulative != nullptr && cum_e != 0) atomicAdd(cumulative + soff, cum_e);
}

}  // namespace mbamd
#endif
