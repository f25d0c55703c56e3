// mbamd_walk4.h -- the 4-state tree-walk kernel (included by mbamd_kernels.h).
//
// Replaces CondLikeDown_NUC4* / CondLikeRoot_NUC4* + CondLikeScaler_NUC4* + RemoveNodeScalers
// (reference src/likelihood.c:786-1570, 2953-4000, 5137-5410, 7981-8070) for a whole operation list
// in ONE launch.
//
// Work decomposition.  A (site pattern, rate category) column is independent of every other column through
// the whole pruning recursion -- also in its rescaling, because the engine keeps one binary exponent per
// (pattern, category) instead of the reference's one scaler per pattern (multiplying by 2^-e is exact, so the
// mantissas are those of a per-pattern scaler; the root integration recombines the categories exactly,
// k_integrate_lnl_s4).  So:
//     lane  = one pattern of a 64-pattern block, ONE category          (no data crosses lanes)
//     wave  = a self-contained interpreter of a host-compiled PROGRAM: a straight list of operations on its
//             64 columns; children produced earlier by the same wave are read back from LDS slots private
//             to the wave (1 KiB each), never from HBM; HBM sees one 1 KiB contiguous store per operation
//     workgroup = W waves (tree parallelism: the host cuts the operation forest into subtrees, packs them into
//             W bins, and separates the few dependent phases with a workgroup barrier; values crossing waves
//             travel through HBM/L2 and an LDS-DMA prefetch), grid = (pattern blocks, categories)
//
// Everything wave-uniform comes through the SCALAR memory path (s_load_* through the scalar cache, counted by
// lgkmcnt): the program entries, the 4x4 transition matrices -- which then feed v_pk_fma_f32 directly as scalar
// operands: no matrix VGPRs, no LDS staging, no cross-lane broadcast -- and the compact tips, which are stored as
// four 64-bit STATE BITPLANES per (tip, pattern block): plane i, bit l = "state i is compatible with pattern l"
// (ambiguity codes included).  A plane in a scalar register pair is a lane mask: the tip's 0/1 vector is four
// v_cndmask_b32.  Stored exponents (dynamic rescaling's "divide by the existing factors" pass) are 64 bytes per
// wave: a scalar load, sixteen v_writelane and one ds_bpermute spread them over the lanes.
//
// Why: the vector-memory counter (vmcnt) retires in order and is shared by loads and stores, so a wave that waits for
// ANY vector load also waits for all its older stores -- microseconds under a saturated write stream.  With the above
// a wave in a full-tree evaluation issues vector STORES only and never waits for one.  The only vector loads left are
// children that live in HBM (results of earlier launches on a partial update, values that crossed waves or were
// evicted): they are LDS-DMA prefetches (global_load_lds_dwordx4, no destination register) issued through inline asm as
// early as the host can place them -- all of a root-ward path's siblings before the first store when the slots allow
// -- and their consumer waits with the exact s_waitcnt vmcnt(N) the host computed by replaying the instruction
// sequence (Walk4Entry vmwait): per iteration [0-2 prefetches] [wait] [2 stores unless NOP].
#ifndef MBAMD_WALK4_H_
#define MBAMD_WALK4_H_

namespace mbamd {

#define MBAMD_W4_NOP      1u     // Walk4Entry flags: no operation (padding so that all waves meet at the barriers)
#define MBAMD_W4_BARRIER  2u     // drain this wave's stores and meet the other waves before reading this entry's children
#define MBAMD_W4_TIP      1u     // child kind: compact tip (state bitplanes); 0 = LDS slot
#define MBAMD_W4_MAXW     8
#define MBAMD_W4_NOWAIT   63u    // vmwait value: this entry reads no prefetched child

// One step of a wave's program (wave-uniform; fetched with one s_load_dwordx8).
struct alignas(32) Walk4Entry {
    uint32_t dst;      // [15:0] destination partials buffer   [23:16] LDS slot that keeps the result (0xFF: none)   [31:24] flags
    uint32_t c1;       // [15:0] tip buffer (kind TIP)          [23:16] LDS slot (kind SLOT)                         [31:24] kind
    uint32_t c2;
    uint32_t mats;     // [15:0] transition matrix of child 1   [31:16] of child 2
    uint32_t scale;    // [15:0] exponent buffer written (scratch unless SCALE_WRITE)   [23:16] ScaleMode   [31:24] vmwait
    uint32_t pf0;      // LDS-DMA prefetch: [15:0] partials buffer   [23:16] LDS slot   [24] valid (pf1 valid only if pf0 is)
    uint32_t pf1;
    uint32_t sread;    // [15:0] exponent buffer read (SCALE_READ)
};
static_assert(sizeof(Walk4Entry) == 32, "Walk4Entry is 8 dwords");

struct Walk4Args {
    const Walk4Entry* prog;      // [W][entries]
    int entries;                 // per wave, including two trailing NOP entries (read-ahead)
    int nslots;                  // LDS slots per wave
    f4* partials;                // arena f4 [block][buffer][K][64]
    unsigned long pstride;       // f4 elements between blocks
    const uint64_t* tips;        // arena uint64 [block][buffer][4] state bitplanes
    unsigned tstride;            // uint64 elements between blocks
    int8_t* exps;                // arena int8 [block][scale buffer][K][64]
    unsigned estride;            // bytes between blocks
    const float* matrices;       // [matrix][K][4][4] transposed
    int32_t* cum;                // wide cumulative buffer int32 [K][Ppad], or nullptr
    int K, Ppad;
};

__host__ __device__ inline size_t walk4_lds_bytes(int W, int nslots) { return (size_t) W * nslots * 1024; }

struct Walk4Planes { uint64_t p[4]; };
struct Walk4Exps { uint32_t d[16]; };

#if defined(MBAMD_HOST_EMU)
struct Walk4Mat { float m[16]; };
__device__ inline Walk4Mat walk4_load_matrix(const float* p) { Walk4Mat r; for (int i = 0; i < 16; ++i) r.m[i] = p[i]; return r; }
__device__ inline Walk4Planes walk4_load_planes(const uint64_t* p) { Walk4Planes r; for (int i = 0; i < 4; ++i) r.p[i] = p[i]; return r; }
__device__ inline Walk4Exps walk4_load_exps(const int8_t* p) { Walk4Exps r; std::memcpy(r.d, p, 64); return r; }
__device__ inline f4 walk4_tip_vector(const Walk4Planes& t, unsigned lane)
{
    f4 v;
    v.x = (float) (t.p[0] >> lane & 1u); v.y = (float) (t.p[1] >> lane & 1u);
    v.z = (float) (t.p[2] >> lane & 1u); v.w = (float) (t.p[3] >> lane & 1u);
    return v;
}
__device__ inline int walk4_lane_exponent(const Walk4Exps& x, unsigned lane) { return (int) (int8_t) (x.d[lane >> 2] >> (8 * (lane & 3))); }
__device__ inline void walk4_dma(const f4* base, unsigned lane, f4* slot) { slot[lane] = base[lane]; }
__device__ inline void walk4_wait_vm(unsigned) {}
__device__ inline void walk4_barrier() { mbamd_emu_barrier(); }
__device__ inline Walk4Entry walk4_load_entry(const Walk4Entry* p) { return *p; }
#else
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u8v __attribute__((ext_vector_type(8)));
typedef unsigned u16v __attribute__((ext_vector_type(16)));
typedef unsigned long ul4v __attribute__((ext_vector_type(4)));
struct Walk4Mat { f16v m; };
// wave-uniform, read-only: constant address space -> s_load_dwordx16 / s_load_dwordx8
__device__ __forceinline__ Walk4Mat walk4_load_matrix(const float* p)
{
    Walk4Mat r;
    r.m = *reinterpret_cast<const MBAMD_AS_CONST f16v*>((uintptr_t) p);
    return r;
}
__device__ __forceinline__ Walk4Entry walk4_load_entry(const Walk4Entry* p)
{
    const u8v v = *reinterpret_cast<const MBAMD_AS_CONST u8v*>((uintptr_t) p);
    Walk4Entry e;
    e.dst = v[0]; e.c1 = v[1]; e.c2 = v[2]; e.mats = v[3]; e.scale = v[4]; e.pf0 = v[5]; e.pf1 = v[6]; e.sread = v[7];
    return e;
}
__device__ __forceinline__ Walk4Planes walk4_load_planes(const uint64_t* p)
{
    const ul4v v = *reinterpret_cast<const MBAMD_AS_CONST ul4v*>((uintptr_t) p);
    Walk4Planes r;
    r.p[0] = v[0]; r.p[1] = v[1]; r.p[2] = v[2]; r.p[3] = v[3];
    return r;
}
__device__ __forceinline__ Walk4Exps walk4_load_exps(const int8_t* p)
{
    const u16v v = *reinterpret_cast<const MBAMD_AS_CONST u16v*>((uintptr_t) p);
    Walk4Exps r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r.d[i] = v[i];
    return r;
}
// a bitplane in a scalar register pair IS a lane mask: one v_cndmask_b32 per state
__device__ __forceinline__ f4 walk4_tip_vector(const Walk4Planes& t, unsigned)
{
    f4 v;
    asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(v.x) : "s"(t.p[0]));
    asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(v.y) : "s"(t.p[1]));
    asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(v.z) : "s"(t.p[2]));
    asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(v.w) : "s"(t.p[3]));
    return v;
}
// 64 exponent bytes in 16 scalar registers -> lane l gets byte l: the dwords go to lanes 0-15 of one VGPR
// (v_writelane), every lane fetches dword l/4 through the LDS crossbar (ds_bpermute) and extracts its byte
__device__ __forceinline__ int walk4_lane_exponent(const Walk4Exps& x, unsigned lane)
{
    int t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) asm("v_writelane_b32 %0, %1, %2" : "+v"(t) : "s"(x.d[i]), "i"(i));
    const int dw = __builtin_amdgcn_ds_bpermute((int) (lane & ~3u), t);          // byte address of lane l/4
    return (int) (int8_t) ((unsigned) dw >> (8 * (lane & 3u)));
}
// LDS-DMA: 64 lanes x 16 bytes (base + lane16 each) straight into the 1 KiB LDS slot at byte address lds_dst
// (lane-linear).  `base` is wave-uniform (scalar registers).  M0 is compiler-reserved: saved and restored inside.
__device__ __forceinline__ void walk4_dma(const f4* base, unsigned lane16, unsigned lds_dst)
{
    // sc0 sc1: served by L2, never by this CU's vector L1 (the line is read once, and it may have been written by
    // another wave of this workgroup a moment ago)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc0 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(base), "s"(lds_dst) : "memory");
}
// wait until at most n vector-memory instructions of this wave are outstanding (s_waitcnt takes an immediate: the
// host rounds n down to one of these values; only entries that read a prefetched child come here)
__device__ __forceinline__ void walk4_wait_vm(unsigned n)
{
#define MBAMD_W4_WAIT(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        MBAMD_W4_WAIT(1) MBAMD_W4_WAIT(2) MBAMD_W4_WAIT(3) MBAMD_W4_WAIT(4) MBAMD_W4_WAIT(5) MBAMD_W4_WAIT(6)
        MBAMD_W4_WAIT(8) MBAMD_W4_WAIT(10) MBAMD_W4_WAIT(12) MBAMD_W4_WAIT(16) MBAMD_W4_WAIT(20) MBAMD_W4_WAIT(24)
        MBAMD_W4_WAIT(32) MBAMD_W4_WAIT(40) MBAMD_W4_WAIT(48) MBAMD_W4_WAIT(56)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef MBAMD_W4_WAIT
}
__device__ __forceinline__ void walk4_barrier()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#endif
// the values walk4_wait_vm implements, for the host: the largest supported count <= n
__host__ __device__ inline unsigned walk4_round_wait(long n)
{
    const unsigned ok[] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 56};
    unsigned r = 0;
    for (unsigned v : ok) if ((long) v <= n) r = v;
    return r;
}

// f_i = sum_j P(i->j) v_j with the transposed matrix mT[j][i] in scalar registers; the same fma chain
// (j = 0..3, first term a plain product) as the reference's scalar loop order, two rows per v_pk_fma_f32.
__device__ __forceinline__ f4 walk4_matvec(const Walk4Mat& M, f4 v)
{
    f4 r;
#if defined(MBAMD_HOST_EMU)
    const float* m = M.m;
    r.x = fmaf(m[12], v.w, fmaf(m[8], v.z, fmaf(m[4], v.y, m[0] * v.x)));
    r.y = fmaf(m[13], v.w, fmaf(m[9], v.z, fmaf(m[5], v.y, m[1] * v.x)));
    r.z = fmaf(m[14], v.w, fmaf(m[10], v.z, fmaf(m[6], v.y, m[2] * v.x)));
    r.w = fmaf(m[15], v.w, fmaf(m[11], v.z, fmaf(m[7], v.y, m[3] * v.x)));
#else
    const f16v m = M.m;
    f2v lo = f2v{m[0], m[1]} * f2v{v.x, v.x};
    f2v hi = f2v{m[2], m[3]} * f2v{v.x, v.x};
    lo = __builtin_elementwise_fma(f2v{m[4], m[5]}, f2v{v.y, v.y}, lo);
    hi = __builtin_elementwise_fma(f2v{m[6], m[7]}, f2v{v.y, v.y}, hi);
    lo = __builtin_elementwise_fma(f2v{m[8], m[9]}, f2v{v.z, v.z}, lo);
    hi = __builtin_elementwise_fma(f2v{m[10], m[11]}, f2v{v.z, v.z}, hi);
    lo = __builtin_elementwise_fma(f2v{m[12], m[13]}, f2v{v.w, v.w}, lo);
    hi = __builtin_elementwise_fma(f2v{m[14], m[15]}, f2v{v.w, v.w}, hi);
    r.x = lo[0]; r.y = lo[1]; r.z = hi[0]; r.w = hi[1];
#endif
    return r;
}

// blockDim.x = 64 * W; grid = (pattern blocks, K).  Dynamic LDS: walk4_lds_bytes(W, nslots).
__global__ void __launch_bounds__(64 * MBAMD_W4_MAXW)
k_walk4(Walk4Args A)
{
    const unsigned lane = threadIdx.x & 63;
#if defined(MBAMD_HOST_EMU)
    const int wave = (int) (threadIdx.x >> 6);
    f4* lds = reinterpret_cast<f4*>(mbamd_emu_dyn_lds());
#else
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    extern __shared__ f4 lds_walk4[];
    f4* lds = lds_walk4;
#endif
    const unsigned blk = blockIdx.x, k = blockIdx.y;
    const unsigned K = (unsigned) A.K;
    f4* const slots = lds + (size_t) wave * A.nslots * 64;                     // this wave's private slots
    // this wave's columns (wave-uniform bases; lane offsets are added by the memory instructions):
    // buffer i of each kind is `i * stride` further
    f4* const P0 = A.partials + (size_t) blk * A.pstride + (size_t) k * 64;
    const unsigned pbuf = K * 64u;                                             // f4 per partials buffer within a block
    const uint64_t* const T0 = A.tips + (size_t) blk * A.tstride;              // 4 planes per tip buffer
    int8_t* const E0 = A.exps + (size_t) blk * A.estride + (size_t) k * 64;
    const unsigned ebuf = K * 64u;
    const float* const M0 = A.matrices + (size_t) k * 16;
    const unsigned mbuf = K * 16u;
#if defined(MBAMD_HOST_EMU)
#define MBAMD_W4_PREFETCH(WORD) walk4_dma(P0 + ((WORD) & 0xFFFFu) * pbuf, lane, slots + (((WORD) >> 16) & 0xFFu) * 64)
#else
    const unsigned lane16 = lane * 16u;
    const unsigned slots_lds = (unsigned) (uintptr_t) (__attribute__((address_space(3))) char*) slots;
#define MBAMD_W4_PREFETCH(WORD) walk4_dma(P0 + ((WORD) & 0xFFFFu) * pbuf, lane16, slots_lds + (((WORD) >> 16) & 0xFFu) * 1024u)
#endif

    const Walk4Entry* prog = A.prog + (size_t) wave * A.entries;
    const int n = A.entries - 2;
    Walk4Entry D0 = walk4_load_entry(prog), D1 = walk4_load_entry(prog + 1);
    Walk4Mat M1 = walk4_load_matrix(M0 + (D0.mats & 0xFFFFu) * mbuf);
    Walk4Mat M2 = walk4_load_matrix(M0 + (D0.mats >> 16) * mbuf);
    // inputs of entry 0 that come through the scalar path, already in their per-lane form
    f4 tipA = walk4_tip_vector(walk4_load_planes(T0 + (D0.c1 & 0xFFFFu) * 4u), lane);
    f4 tipB = walk4_tip_vector(walk4_load_planes(T0 + (D0.c2 & 0xFFFFu) * 4u), lane);
    int eread = 0;
    if (((D0.scale >> 16) & 0xFFu) == SCALE_READ) eread = walk4_lane_exponent(walk4_load_exps(E0 + (D0.sread & 0xFFFFu) * ebuf), lane);
    int cum_e = 0;

    // Vector-memory instruction sequence of iteration j (the host's vmwait counts on exactly this):
    //     [DMA pf0] [DMA pf1]   s_waitcnt vmcnt(vmwait_j) if the entry reads a prefetched child   [2 stores, unless NOP]
    // Everything wave-uniform is a scalar branch or a scalar select (no divergent control flow).
    for (int j = 0; j < n; ++j) {
        const Walk4Entry D2 = walk4_load_entry(prog + j + 2);
        // scalar-path inputs of the NEXT entry: requested now, turned into per-lane values at the end of this iteration
        // (a child that is not a tip reads the planes of buffer 0: a valid address, the value is not used)
        const Walk4Planes np1 = walk4_load_planes(T0 + (D1.c1 & 0xFFFFu) * 4u);
        const Walk4Planes np2 = walk4_load_planes(T0 + (D1.c2 & 0xFFFFu) * 4u);
        // prefetches riding on this entry: children of this or later operations that live in HBM -> LDS slots
        if (D0.pf0 & (1u << 24)) {
            MBAMD_W4_PREFETCH(D0.pf0);
            if (D0.pf1 & (1u << 24)) MBAMD_W4_PREFETCH(D0.pf1);
        }
        const unsigned vmwait = D0.scale >> 24;
        if (vmwait != MBAMD_W4_NOWAIT) walk4_wait_vm(vmwait);                   // the prefetched children this entry reads have landed
        const unsigned flags = D0.dst >> 24;
        if (flags & MBAMD_W4_BARRIER) walk4_barrier();
        if (!(flags & MBAMD_W4_NOP)) {
            const bool tip1 = (D0.c1 >> 24) == MBAMD_W4_TIP, tip2 = (D0.c2 >> 24) == MBAMD_W4_TIP;
            f4 a = tipA, b = tipB;
            if (!tip1) a = slots[((D0.c1 >> 16) & 0xFFu) * 64 + lane];
            if (!tip2) b = slots[((D0.c2 >> 16) & 0xFFu) * 64 + lane];
            const f4 f1 = walk4_matvec(M1, a);
            const f4 f2 = walk4_matvec(M2, b);
            f4 out;
            out.x = f1.x * f2.x; out.y = f1.y * f2.y; out.z = f1.z * f2.z; out.w = f1.w * f2.w;
            const unsigned mode = (D0.scale >> 16) & 0xFFu;
            const int wm = mode == SCALE_WRITE ? -1 : 0, rm = mode == SCALE_READ ? -1 : 0;
            const int e = (scale_exponent(max4(out)) & wm) | (eread & rm);
            cum_e += e & wm;
            out.x = scale_pow2(out.x, -e); out.y = scale_pow2(out.y, -e);       // (2^0 is exact: no branch)
            out.z = scale_pow2(out.z, -e); out.w = scale_pow2(out.w, -e);
            const unsigned dslot = (D0.dst >> 16) & 0xFFu;
            if (dslot != 0xFFu) slots[dslot * 64 + lane] = out;
            as_global(P0 + (D0.dst & 0xFFFFu) * pbuf)[lane] = out;              // 1 KiB contiguous per wave; never waited for
            as_global(E0 + (D0.scale & 0xFFFFu) * ebuf)[lane] = (int8_t) e;
        }
        // the next entry's matrices (the registers of this entry's are free now), its tips and its stored exponents
        M1 = walk4_load_matrix(M0 + (D1.mats & 0xFFFFu) * mbuf);
        M2 = walk4_load_matrix(M0 + (D1.mats >> 16) * mbuf);
        tipA = walk4_tip_vector(np1, lane);
        tipB = walk4_tip_vector(np2, lane);
        if (((D1.scale >> 16) & 0xFFu) == SCALE_READ) eread = walk4_lane_exponent(walk4_load_exps(E0 + (D1.sread & 0xFFFFu) * ebuf), lane);
        D0 = D1; D1 = D2;
    }
#undef MBAMD_W4_PREFETCH
    if (A.cum != nullptr && cum_e != 0) atomicAdd(A.cum + (size_t) k * A.Ppad + (size_t) blk * 64 + lane, cum_e);
}

}  // namespace mbamd
#endif
