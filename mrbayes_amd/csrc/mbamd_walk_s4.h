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
//   * Every result goes into an LDS slot chosen by the host (Belady eviction), so a parent reads its
//     children back from LDS: HBM never serves a re-read of a value produced in the same launch.
//   * Compute waves issue NO global stores.  One extra WRITER wave per workgroup copies the slots
//     produced in step s-1 to HBM while the compute waves work on step s (4K KiB contiguous per
//     node in the block-major arena) and stores the node's scale exponents.  The vector-memory
//     counter (vmcnt) retires in order, so a wave that both stores and prefetches has to drain
//     its stores before it can consume a younger load; with the roles split, the compute waves' vmcnt
//     queue holds only prefetches and the writer never waits for anything but LDS.
//   * What an operation needs from memory -- its descriptor, its 2K transposed 4x4 matrices, the
//     state codes of compact tips -- is requested one or more steps ahead (software pipeline,
//     ping-pong registers) and costs a handful of VGPRs while in flight.
//   * Matrix elements reach the FMAs through DPP quad broadcasts (see Mat4): 4 VGPRs per matrix.
//   * The per-pattern max-rescale (the reference's separate CondLikeScaler pass) is fused in; the
//     factor is the power of two 2^-e (top of mbamd_kernels.h) and the cumulative buffer gets
//     the per-lane sum of e with one integer atomic per compute wave at the end.
//
// LDS map (16-byte units): [W][8K] matrix staging | [2][W][2] writer mailboxes | [slots][64K + 16]
// values + exponents.  Only when a value was evicted from LDS and must be re-read from memory
// (CHILD_RELOAD, at least two steps after it was produced) does the host flag MBAMD_OP_DRAIN so that the
// writer waits for its stores before the barrier that precedes the re-read.
#ifndef MBAMD_WALK_S4_H_
#define MBAMD_WALK_S4_H_

namespace mbamd {

#define MBAMD_OP_DRAIN 1      // PartialsOp::flags: the writer drains its stores after handling this step's results
#define MBAMD_OP_SLOW  2      // the operation reads a child from global memory or divides by stored scale factors

__host__ __device__ inline int walk_slot_units(int K) { return 64 * K + 16; }          // f4 per value slot
__host__ __device__ inline int walk_lds_units(int K, int W, int slots)
{
    return W * 8 * K + 2 * W * 2 + slots * walk_slot_units(K);
}

__device__ __forceinline__ void walk_step_barrier()
{
#if !defined(MBAMD_HOST_EMU)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
}

// the fields of a PartialsOp as the kernel holds them in (scalar) registers
struct WalkFields {
    float* dst;
    const void *c1, *c2;
    const float *m1, *m2;
    int32_t* scale;
    int c1_kind, c2_kind, c1_slot, c2_slot, dst_slot, scale_mode, flags;
};
__device__ __forceinline__ WalkFields walk_fields(const PartialsOp* p)
{
    WalkFields f;
    f.dst = p->dst; f.c1 = p->c1; f.c2 = p->c2; f.m1 = p->m1; f.m2 = p->m2; f.scale = p->scale;
    f.c1_kind = p->c1_kind; f.c2_kind = p->c2_kind; f.c1_slot = p->c1_slot; f.c2_slot = p->c2_slot;
    f.dst_slot = p->dst_slot; f.scale_mode = p->scale_mode; f.flags = p->flags;
    return f;
}

// What one operation needs from memory besides LDS-resident children: its 2K transposed 4x4
// matrices (8K rows of 16 bytes, one dwordx4 load by lanes 0..8K-1) and the state codes of compact
// tip children.  Straight-line on purpose (no branch around a load): lanes >= 8K repeat the last
// row, and both "tip" bytes are always loaded (garbage when the child is not a compact tip -- the
// pointers of every table entry, empty ones included, address readable memory).
struct WalkInputs {
    f4 mrow;
    unsigned s1, s2;
};
template <int K>
__device__ __forceinline__ void walk_request(const WalkFields& op, size_t toff, int lane, WalkInputs& in)
{
#if defined(MBAMD_HOST_EMU)
    in.mrow = f4{0.0f, 0.0f, 0.0f, 0.0f};
    in.s1 = (op.c1_kind == CHILD_STATES) ? (unsigned) reinterpret_cast<const uint8_t*>(op.c1)[toff] : 0u;
    in.s2 = (op.c2_kind == CHILD_STATES) ? (unsigned) reinterpret_cast<const uint8_t*>(op.c2)[toff] : 0u;
    (void) lane;
#else
    const int r = lane < 8 * K ? lane : 8 * K - 1;
    const bool first = r < 4 * K;
    const float* base = first ? op.m1 : op.m2;
    in.mrow = as_global(reinterpret_cast<const f4*>(base))[first ? r : r - 4 * K];
    in.s1 = (unsigned) as_global(reinterpret_cast<const uint8_t*>(op.c1))[toff];
    in.s2 = (unsigned) as_global(reinterpret_cast<const uint8_t*>(op.c2))[toff];
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

// matrices of one operation: spread the prefetched rows over the wave through the staging area
template <int K>
__device__ __forceinline__ void walk_matrices(const WalkFields& op, const WalkInputs& in, f4* stage, int lane,
                                              Mat4 (&M1)[K], Mat4 (&M2)[K])
{
#if defined(MBAMD_HOST_EMU)
    (void) in; (void) stage;
#pragma unroll
    for (int k = 0; k < K; ++k) { M1[k] = mat4_load(op.m1 + 16 * k, lane); M2[k] = mat4_load(op.m2 + 16 * k, lane); }
#else
    (void) op;
    stage[lane < 8 * K ? lane : 8 * K - 1] = in.mrow;       // lanes >= 8K rewrite the last row with the same value
#pragma unroll
    for (int k = 0; k < K; ++k) {
        M1[k].col = stage[4 * k + (lane & 3)];
        M2[k].col = stage[4 * K + 4 * k + (lane & 3)];
    }
#endif
}

// product of the two child factors, rescale, result -> LDS slot (values + exponent row)
template <int K>
__device__ __forceinline__ void walk_finish(const WalkFields& op, f4 (&out)[K], float mx, int e_read, int lane, f4* slots,
                                            int& cum_e)
{
    int e = 0;
    if (op.scale_mode == SCALE_WRITE) { e = scale_exponent(mx); cum_e += e; }
    else if (op.scale_mode == SCALE_READ) e = e_read;
    f4* slot = slots + op.dst_slot * walk_slot_units(K);
#pragma unroll
    for (int k = 0; k < K; ++k) {           // multiplying by 2^0 is exact: no branch needed
        out[k].x = scale_pow2(out[k].x, -e);
        out[k].y = scale_pow2(out[k].y, -e);
        out[k].z = scale_pow2(out[k].z, -e);
        out[k].w = scale_pow2(out[k].w, -e);
        slot[k * 64 + lane] = out[k];
    }
    reinterpret_cast<int*>(slot + 64 * K)[lane] = e;
}

// The common case -- every child is an LDS-resident value or a compact tip, scaling is "write" or
// "none" -- as straight-line code without a single vector-memory instruction.
template <int K>
__device__ __forceinline__ void walk_op_fast(const WalkFields& op, const WalkInputs& in, f4* stage, int lane, f4* slots,
                                             int& cum_e)
{
    Mat4 M1[K], M2[K];
    walk_matrices<K>(op, in, stage, lane, M1, M2);
    const bool tip1 = op.c1_kind == CHILD_STATES, tip2 = op.c2_kind == CHILD_STATES;
    const f4* l1 = slots + (tip1 ? 0 : op.c1_slot) * walk_slot_units(K) + lane;   // tips read slot 0 and discard it
    const f4* l2 = slots + (tip2 ? 0 : op.c2_slot) * walk_slot_units(K) + lane;
    f4 a[K], b[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { a[k] = l1[k * 64]; b[k] = l2[k * 64]; }
    const f4 one1 = tip_vector(in.s1), one2 = tip_vector(in.s2);
    f4 out[K];
    float mx = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const f4 f1 = mat4_mul(M1[k], tip1 ? one1 : a[k]);
        const f4 f2 = mat4_mul(M2[k], tip2 ? one2 : b[k]);
        out[k].x = f1.x * f2.x;
        out[k].y = f1.y * f2.y;
        out[k].z = f1.z * f2.z;
        out[k].w = f1.w * f2.w;
        mx = fmaxf(mx, max4(out[k]));
    }
    walk_finish<K>(op, out, mx, 0, lane, slots, cum_e);
}

// Everything else: children in global memory (buffers not produced in this launch, or values
// evicted from LDS -- their writer drained its stores before the preceding barrier and all waves
// of a workgroup share one vector L1, so a plain load observes them), stored-factor scaling.
template <int K>
__device__ __forceinline__ void walk_op_slow(const WalkFields& op, const WalkInputs& in, f4* stage, size_t poff,
                                             size_t soff, int lane, f4* slots, int& cum_e)
{
    Mat4 M1[K], M2[K];
    walk_matrices<K>(op, in, stage, lane, M1, M2);
    f4 a[K], b[K];
    const void* cp[2] = {op.c1, op.c2};
    const int ck[2] = {op.c1_kind, op.c2_kind}, cs[2] = {op.c1_slot, op.c2_slot};
    const unsigned st[2] = {in.s1, in.s2};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f4 (&v)[K] = t == 0 ? a : b;
        if (ck[t] == CHILD_LDS) {
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = slots[cs[t] * walk_slot_units(K) + k * 64 + lane];
        } else if (ck[t] == CHILD_STATES) {
            const f4 one = tip_vector(st[t]);
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = one;
        } else {
            const MBAMD_AS_GLOBAL f4* p = as_global(reinterpret_cast<const f4*>(cp[t])) + poff;
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = p[k * 64];
        }
    }
    const int e_read = (op.scale_mode == SCALE_READ) ? as_global(op.scale)[soff] : 0;
    f4 out[K];
    float mx = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const f4 f1 = mat4_mul(M1[k], a[k]);
        const f4 f2 = mat4_mul(M2[k], b[k]);
        out[k].x = f1.x * f2.x;
        out[k].y = f1.y * f2.y;
        out[k].z = f1.z * f2.z;
        out[k].w = f1.w * f2.w;
        mx = fmaxf(mx, max4(out[k]));
    }
    walk_finish<K>(op, out, mx, e_read, lane, slots, cum_e);
}

// writer: one finished value, LDS slot -> its buffer in HBM (+ its scale exponents)
template <int K>
__device__ __forceinline__ void walk_write_out(float* dst, int32_t* scale, int slot, size_t poff, size_t soff, int lane,
                                               const f4* slots)
{
    const f4* sl = slots + slot * walk_slot_units(K);
    f4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = sl[k * 64 + lane];
    const int e = reinterpret_cast<const int*>(sl + 64 * K)[lane];
    MBAMD_AS_GLOBAL f4* __restrict__ d = as_global(reinterpret_cast<f4*>(dst)) + poff;
#pragma unroll
    for (int k = 0; k < K; ++k) d[k * 64] = v[k];          // 4K KiB contiguous per node update
    as_global(scale)[soff] = e;                            // (non-rescaling operations point at a scratch buffer)
}

#if !defined(MBAMD_HOST_EMU)
// One row of the schedule = the W descriptors of a step = 16*W dwords.  A wave fetches a row with
// two coalesced dword loads (lane l holds dwords l and l+64; the vector L1 serves the other waves
// of the workgroup) and later picks its own 16 dwords out of the lanes with v_readlane: the
// descriptors travel through the vmcnt queue like every other prefetch and end up in SGPRs.
struct WalkRow { unsigned lo, hi; };
__device__ __forceinline__ WalkRow walk_row_request(const PartialsOp* ops, int step, int W, int lane)
{
    const MBAMD_AS_GLOBAL unsigned* p = as_global(reinterpret_cast<const unsigned*>(ops)) + (size_t) step * W * 16;
    WalkRow r;
    r.lo = p[lane < W * 16 ? lane : 0];
    r.hi = p[W > 4 ? 64 + lane : lane & 15];               // always issued: the load count must not depend on W
    return r;
}
__device__ __forceinline__ WalkFields walk_row_fields(const WalkRow& r, int wave)
{
    const unsigned v = (wave >= 4) ? r.hi : r.lo;
    const int base = (wave & 3) * 16;
    unsigned d[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) d[i] = (unsigned) __builtin_amdgcn_readlane((int) v, base + i);
    WalkFields f;
    f.dst = reinterpret_cast<float*>(((unsigned long) d[1] << 32) | d[0]);
    f.c1 = reinterpret_cast<const void*>(((unsigned long) d[3] << 32) | d[2]);
    f.c2 = reinterpret_cast<const void*>(((unsigned long) d[5] << 32) | d[4]);
    f.m1 = reinterpret_cast<const float*>(((unsigned long) d[7] << 32) | d[6]);
    f.m2 = reinterpret_cast<const float*>(((unsigned long) d[9] << 32) | d[8]);
    f.scale = reinterpret_cast<int32_t*>(((unsigned long) d[11] << 32) | d[10]);
    const unsigned lo = d[12], hi = d[13];
    f.c1_kind = lo & 0xFF; f.c2_kind = (lo >> 8) & 0xFF; f.c1_slot = (lo >> 16) & 0xFF; f.c2_slot = lo >> 24;
    f.dst_slot = hi & 0xFF; f.scale_mode = (hi >> 8) & 0xFF; f.flags = (hi >> 16) & 0xFF;
    return f;
}

// writer: the results of one finished step, as announced in the compute waves' mailboxes.  All
// mailboxes are read in one batch, then the slots in groups of four (17 VGPRs per value in flight),
// so a step costs the writer three LDS round trips, not 2W.
#define MBAMD_WALK_MAXW 7
template <int K>
__device__ __forceinline__ bool walk_writer_step(const f4* mailbox, int W, size_t poff, size_t soff, int lane,
                                                 const f4* slots)
{
    f4 m0[MBAMD_WALK_MAXW], m1[MBAMD_WALK_MAXW];
#pragma unroll
    for (int w = 0; w < MBAMD_WALK_MAXW; ++w) {
        const int ww = w < W ? w : 0;                                    // same address in every lane: broadcast
        m0[w] = mailbox[2 * ww];
        m1[w] = mailbox[2 * ww + 1];
    }
    float* dst[MBAMD_WALK_MAXW];
    int32_t* scale[MBAMD_WALK_MAXW];
    int slot[MBAMD_WALK_MAXW];
    bool drain = false;
#pragma unroll
    for (int w = 0; w < MBAMD_WALK_MAXW; ++w) {
        const unsigned d0 = __builtin_amdgcn_readfirstlane(__float_as_uint(m0[w].x));
        const unsigned d1 = __builtin_amdgcn_readfirstlane(__float_as_uint(m0[w].y));
        const unsigned s0 = __builtin_amdgcn_readfirstlane(__float_as_uint(m0[w].z));
        const unsigned s1 = __builtin_amdgcn_readfirstlane(__float_as_uint(m0[w].w));
        slot[w] = (int) __builtin_amdgcn_readfirstlane(__float_as_uint(m1[w].x));
        const int flags = (int) __builtin_amdgcn_readfirstlane(__float_as_uint(m1[w].y));
        dst[w] = (w < W) ? reinterpret_cast<float*>(((unsigned long) d1 << 32) | d0) : nullptr;
        scale[w] = reinterpret_cast<int32_t*>(((unsigned long) s1 << 32) | s0);
        drain |= (w < W) && (flags & MBAMD_OP_DRAIN) != 0;
        if (dst[w] == nullptr) slot[w] = 0;                               // (read slot 0, store nothing)
    }
#pragma unroll
    for (int g0 = 0; g0 < MBAMD_WALK_MAXW; g0 += 4) {
        if (g0 >= W) break;
        f4 v[4][K];
        int e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int w = g0 + i < MBAMD_WALK_MAXW ? g0 + i : MBAMD_WALK_MAXW - 1;
            const f4* sl = slots + slot[w] * walk_slot_units(K);
#pragma unroll
            for (int k = 0; k < K; ++k) v[i][k] = sl[k * 64 + lane];
            e[i] = reinterpret_cast<const int*>(sl + 64 * K)[lane];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (g0 + i >= MBAMD_WALK_MAXW) continue;
            const int w = g0 + i;
            if (dst[w] == nullptr) continue;
            MBAMD_AS_GLOBAL f4* __restrict__ d = as_global(reinterpret_cast<f4*>(dst[w])) + poff;
#pragma unroll
            for (int k = 0; k < K; ++k) d[k * 64] = v[i][k];           // 4K KiB contiguous per node update
            as_global(scale[w])[soff] = e[i];                          // (non-rescaling operations point at a scratch buffer)
        }
    }
    return drain;
}
#endif

// ops: [nsteps + 5][W] (an empty entry has dst == nullptr and valid dummy pointers; flags are
// replicated over a step's entries; five empty rows pad the end for the prefetch pipeline).
// blockDim.x == 64*(W+1): W compute waves + the writer wave.
// (Host emulation: one 64-thread block; each thread runs the W entries of a step one after the
//  other -- lanes never exchange data -- and plays the writer one step behind, like the GPU.)
template <int K>
__global__ void __launch_bounds__(512, (K <= 4 ? 4 : 2))
k_walk_s4(const PartialsOp* __restrict__ ops, int nsteps, int W, BlockGeom g, int32_t* __restrict__ cumulative,
          long long* __restrict__ trace)
{
    const int lane = threadIdx.x & 63;
    const size_t poff = (size_t) blockIdx.x * g.pstride + lane;     // this workgroup's block in every buffer
    const size_t toff = (size_t) blockIdx.x * g.tstride + lane;
    const size_t soff = (size_t) blockIdx.x * g.sstride + lane;
    int cum_e = 0;
#if defined(MBAMD_HOST_EMU)
    (void) trace;
    const bool reversed = W < 0;                 // test hook: run a step's entries in the opposite order
    if (reversed) W = -W;
    f4* slots = reinterpret_cast<f4*>(mbamd_emu_dyn_lds()) + W * 8 * K + 2 * W * 2;
    for (int s = 0; s <= nsteps; ++s) {
        if (s < nsteps)
            for (int i = 0; i < W; ++i) {
                const PartialsOp* op = ops + (size_t) s * W + (reversed ? W - 1 - i : i);
                if (op->dst == nullptr) continue;
                const WalkFields f = walk_fields(op);
                WalkInputs in;
                walk_request<K>(f, toff, lane, in);
                walk_op_slow<K>(f, in, nullptr, poff, soff, lane, slots, cum_e);
            }
        if (s >= 1)                               // the writer is one step behind the compute waves
            for (int i = 0; i < W; ++i) {
                const PartialsOp* op = ops + (size_t) (s - 1) * W + i;
                if (op->dst != nullptr) walk_write_out<K>(op->dst, op->scale, op->dst_slot, poff, soff, lane, slots);
            }
    }
#else
    extern __shared__ f4 lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    f4* mailboxes = lds_all + W * (8 * K);                // [2][W][2]
    f4* slots = mailboxes + 2 * W * 2;
    // (timing experiments: MBAMD_WALK_TRACE makes block 0 record s_memtime stamps [step][wave][3])
    const bool tracing = trace != nullptr && blockIdx.x == 0 && lane == 0;
#define MBAMD_STAMP(STEP, I) if (tracing) trace[((size_t) (STEP) * 8 + wave) * 3 + (I)] = (long long) __builtin_amdgcn_s_memtime();
    if (wave == W) {
        // ---- writer wave ---------------------------------------------------------------------------
        for (int s = 0; s < nsteps; ++s) {
            MBAMD_STAMP(s, 0)
            if (s >= 1) {
                const bool drain = walk_writer_step<K>(mailboxes + ((s - 1) & 1) * W * 2, W, poff, soff, lane, slots);
                if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            MBAMD_STAMP(s, 1)
            walk_step_barrier();
            MBAMD_STAMP(s, 2)
        }
        (void) walk_writer_step<K>(mailboxes + ((nsteps - 1) & 1) * W * 2, W, poff, soff, lane, slots);
        return;
    }
    // ---- compute waves -----------------------------------------------------------------------------
    f4* stage = lds_all + wave * (8 * K);                 // per-wave matrix staging: 8K rows of 16 bytes
    // Software pipeline: while step s computes, the inputs of step s+1 and the descriptor rows of
    // steps s+2, s+3 are in flight (ping-pong registers: copying the destination of an in-flight
    // load would force a wait for it).
    WalkFields cur = walk_row_fields(walk_row_request(ops, 0, W, lane), wave);
    WalkFields nxt = walk_row_fields(walk_row_request(ops, 1, W, lane), wave);
    WalkRow rowX = walk_row_request(ops, 2, W, lane), rowY = rowX;
    WalkInputs inA, inB;
    walk_request<K>(cur, toff, lane, inA);
    inB = inA;
#define MBAMD_WALK_STEP(IN_CUR, IN_NXT, ROW_LOAD, ROW_USE, STEP)                                              \
    {                                                                                                          \
        MBAMD_STAMP(STEP, 0)                                                                                   \
        walk_request<K>(nxt, toff, lane, IN_NXT);                                                              \
        ROW_LOAD = walk_row_request(ops, (STEP) + 3, W, lane);                                                 \
        if (cur.dst != nullptr) {                                                                              \
            if (cur.flags & MBAMD_OP_SLOW) walk_op_slow<K>(cur, IN_CUR, stage, poff, soff, lane, slots, cum_e); \
            else                           walk_op_fast<K>(cur, IN_CUR, stage, lane, slots, cum_e);             \
        }                                                                                                      \
        if (lane == 0) {                                  /* tell the writer what this wave produced */        \
            f4* mb = mailboxes + ((((STEP) & 1) * W) + wave) * 2;                                              \
            const unsigned long dp = reinterpret_cast<unsigned long>(cur.dst);                                 \
            const unsigned long sp = reinterpret_cast<unsigned long>(cur.scale);                               \
            f4 m0, m1;                                                                                         \
            m0.x = __uint_as_float((unsigned) dp); m0.y = __uint_as_float((unsigned) (dp >> 32));              \
            m0.z = __uint_as_float((unsigned) sp); m0.w = __uint_as_float((unsigned) (sp >> 32));              \
            m1.x = __uint_as_float((unsigned) cur.dst_slot); m1.y = __uint_as_float((unsigned) cur.flags);     \
            m1.z = 0.0f; m1.w = 0.0f;                                                                          \
            mb[0] = m0; mb[1] = m1;                                                                            \
        }                                                                                                      \
        MBAMD_STAMP(STEP, 1)                                                                                   \
        walk_step_barrier();                                                                                   \
        MBAMD_STAMP(STEP, 2)                                                                                   \
        cur = nxt;                                                                                             \
        nxt = walk_row_fields(ROW_USE, wave);                                                                  \
    }
    for (int s = 0; s < nsteps; s += 2) {
        MBAMD_WALK_STEP(inA, inB, rowY, rowX, s)
        if (s + 1 >= nsteps) break;
        MBAMD_WALK_STEP(inB, inA, rowX, rowY, s + 1)
    }
#undef MBAMD_WALK_STEP
#undef MBAMD_STAMP
#endif
    if (cumulative != nullptr && cum_e != 0) atomicAdd(cumulative + soff, cum_e);
}

}  // namespace mbamd
#endif
