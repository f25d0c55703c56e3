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
// v_cndmask_b32.
//
// Why: the vector-memory counter (vmcnt) retires in order and is shared by loads and stores, so a wave that waits for
// ANY vector load also waits for all its older stores -- microseconds under a saturated write stream.  With the above
// a wave in a full-tree evaluation issues vector STORES only and never waits for one.  The only vector loads left are
// children that live in HBM (results of earlier launches on a partial update, values that crossed waves or were
// evicted): they are LDS-DMA prefetches (global_load_lds_dwordx4, no destination register) issued through inline asm as
// early as the host can place them -- all of a root-ward path's siblings before the first store when the slots allow
// -- and the stored exponents of dynamic rescaling's "divide by the existing factors" pass (one byte per lane: an LDS-DMA
// of the next entry's 64 bytes into a two-deep staging area).  Their consumer waits with the exact s_waitcnt vmcnt(N) the
// host computed by replaying the instruction sequence (Walk4Entry vmwait): per iteration
// [0-2 prefetches (PF entries only)] [wait] [exponent DMA for the next entry if it is SCALE_READ] [2 stores if an operation].
// The stores are non-temporal: a result is never read again in the launch that wrote it (parents read the LDS copy), and
// letting 0.7 GB of write-allocated lines stream through L2 evicted the matrices and programs every wave keeps
// re-reading -- every scalar load then paid an HBM round trip (measured: 1.6x on the whole kernel).
#ifndef MBAMD_WALK4_H_
#define MBAMD_WALK4_H_

namespace mbamd {

// Walk4Entry::ctl flag bits
#define MBAMD_W4_NOP      0x01u  // no operation (padding so that all waves meet at the barriers; PF entries)
#define MBAMD_W4_BARRIER  0x02u  // drain this wave's stores and meet the other waves before reading this entry's children
#define MBAMD_W4_PF0      0x04u  // PF entry: LDS-DMA prefetch 0 (and, with PF1, prefetch 1)
#define MBAMD_W4_PF1      0x08u
#define MBAMD_W4_VMWAIT   0x10u  // this entry reads something an LDS-DMA brought: s_waitcnt vmcnt(vmwait) first
#define MBAMD_W4_TIP1     0x20u  // child 1 / 2 is a compact tip (state bitplanes); otherwise an LDS slot
#define MBAMD_W4_TIP2     0x40u
#define MBAMD_W4_KEEP     0x80u  // the result is also written to LDS slot `keep`
#define MBAMD_W4_FWD1     0x01000000u  // child 1 / 2 is the result of the operation this wave executed last: still in registers
#define MBAMD_W4_FWD2     0x02000000u
#define MBAMD_W4_RARE     (MBAMD_W4_NOP | MBAMD_W4_BARRIER | MBAMD_W4_PF0 | MBAMD_W4_VMWAIT)
#define MBAMD_W4_MAXW     8

// One step of a wave's program (wave-uniform; fetched with one s_load_dwordx8).  Addresses are ready-made byte
// offsets from a base the wave computes once (scalar adds only, no multiplications in the loop).
struct alignas(32) Walk4Entry {
    uint32_t ctl;      // [7:0] flags   [9:8] ScaleMode   [15:10] vmwait   [23:16] slot that keeps the result (flag KEEP)   [25:24] FWD1 / FWD2
    uint32_t dst;      // destination partials buffer: byte offset inside this wave's (block, category) column set
    uint32_t c1;       // child 1: tip -> byte offset of its 4 bitplanes inside the block's tip area; else LDS byte offset of its slot
    uint32_t c2;
    uint32_t m1;       // transition matrices: byte offsets (this wave's category)
    uint32_t m2;
    uint32_t ewrite;   // exponent buffer written (the scratch buffer unless SCALE_WRITE): byte offset
    uint32_t eread;    // exponent buffer read (SCALE_READ): byte offset
    // PF entry (flags NOP | PF0 [| PF1]): dst / c1 = partials byte offset -> LDS slot byte offset of prefetch 0, c2 / m1 of prefetch 1
};
static_assert(sizeof(Walk4Entry) == 32, "Walk4Entry is 8 dwords");

struct Walk4Args {
    const Walk4Entry* prog;      // [W][entries]
    int entries;                 // per wave: even, including two trailing NOP entries (read-ahead)
    int nslots;                  // LDS slots per wave
    f4* partials;                // arena f4 [block][buffer][K][64]
    unsigned long pstride;       // f4 elements between blocks
    const uint64_t* tips;        // arena uint64 [block][buffer][4] state bitplanes
    unsigned tstride;            // uint64 elements between blocks
    int8_t* exps;                // arena int8 [block][scale buffer][K][64]
    unsigned estride;            // bytes between blocks
    const float* matrices;       // [matrix][K][4][4] transposed
    int32_t* cum;                // wide cumulative buffer int32 [K][Ppad], or nullptr
    int cumFresh;                // the cumulative buffer holds nothing yet: store the sums instead of adding them
    int K, Ppad, nblocks;
    int tail;                    // trailing NOP entries of every program (read-ahead): 2, or tipAhead + 1
    int tipAhead;                // > 0: touch the tip bitplanes of the entry this far ahead (walk4_touch_planes), else 0
};

#define MBAMD_W4_STAGE 768       // bytes per wave in front of its slots: two 64-dword landing areas for stored exponents + one nobody reads
__host__ __device__ inline size_t walk4_lds_bytes(int W, int nslots) { return (size_t) W * (MBAMD_W4_STAGE + (size_t) nslots * 1024); }
// The grid is one-dimensional and XCD-aware: workgroup id -> XCD id % 8 (observed dispatch rule), and the K category
// workgroups of one pattern block get consecutive positions on ONE XCD, so that the tip bitplanes and the matrix
// lines they all read are fetched into that XCD's L2 once.
__host__ __device__ inline unsigned walk4_grid(int nblocks, int K) { return 8u * (unsigned) K * (unsigned) ((nblocks + 7) / 8); }

struct Walk4Planes { uint64_t p[4]; };

#if defined(MBAMD_HOST_EMU)
struct Walk4Mat { float m[16]; };
__device__ inline Walk4Mat walk4_load_matrix(const float* p) { Walk4Mat r; for (int i = 0; i < 16; ++i) r.m[i] = p[i]; return r; }
__device__ inline Walk4Planes walk4_load_planes(const uint64_t* p) { Walk4Planes r; for (int i = 0; i < 4; ++i) r.p[i] = p[i]; return r; }
__device__ inline f4 walk4_tip_vector(const Walk4Planes& t, unsigned lane)
{
    f4 v;
    v.x = (float) (t.p[0] >> lane & 1u); v.y = (float) (t.p[1] >> lane & 1u);
    v.z = (float) (t.p[2] >> lane & 1u); v.w = (float) (t.p[3] >> lane & 1u);
    return v;
}
__device__ inline void walk4_dma(const f4* base, unsigned lane, f4* slot) { slot[lane] = base[lane]; }
__device__ inline void walk4_dma_exps(const int8_t* base, unsigned lane, int* stage) { stage[lane] = base[lane]; }
__device__ inline void walk4_touch_planes(const uint64_t*, const uint64_t*, unsigned, int*) {}
__device__ inline void walk4_wait_vm(unsigned) {}
__device__ inline void walk4_barrier() { mbamd_emu_barrier(); }
__device__ inline Walk4Entry walk4_load_entry(const Walk4Entry* p) { return *p; }
struct Walk4Half { unsigned ctl, dst, c1, c2; };
__device__ inline Walk4Half walk4_load_half(const Walk4Entry* p) { Walk4Half h; h.ctl = p->ctl; h.dst = p->dst; h.c1 = p->c1; h.c2 = p->c2; return h; }
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
    e.ctl = v[0]; e.dst = v[1]; e.c1 = v[2]; e.c2 = v[3]; e.m1 = v[4]; e.m2 = v[5]; e.ewrite = v[6]; e.eread = v[7];
    return e;
}
// the first half of an entry (ctl, dst, c1, c2): all the far-ahead tip touch needs
struct Walk4Half { unsigned ctl, dst, c1, c2; };
__device__ __forceinline__ Walk4Half walk4_load_half(const Walk4Entry* p)
{
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const u4v v = *reinterpret_cast<const MBAMD_AS_CONST u4v*>((uintptr_t) p);
    Walk4Half h;
    h.ctl = v[0]; h.dst = v[1]; h.c1 = v[2]; h.c2 = v[3];
    return h;
}
__device__ __forceinline__ Walk4Planes walk4_load_planes(const uint64_t* p)
{
    const ul4v v = *reinterpret_cast<const MBAMD_AS_CONST ul4v*>((uintptr_t) p);
    Walk4Planes r;
    r.p[0] = v[0]; r.p[1] = v[1]; r.p[2] = v[2]; r.p[3] = v[3];
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
// one signed byte per lane (base + lane) -> a dword per lane at LDS byte address lds_dst + 4 * lane
__device__ __forceinline__ void walk4_dma_exps(const int8_t* base, unsigned lane, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_sbyte %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane), "s"(base), "s"(lds_dst) : "memory");
}
// Pull the 64-byte lines that hold two sets of tip bitplanes into this XCD's L2 ahead of the scalar loads that will want
// them: every lane asks for the same dword (one request), the LDS-DMA form has no destination register to keep alive, and
// what lands (256 bytes at lds_dst, twice) is never read.  Two vector-memory instructions, counted by the host like the others.
__device__ __forceinline__ void walk4_touch_planes(const uint64_t* p1, const uint64_t* p2, unsigned zero, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(zero), "s"(p1), "s"(p2), "s"(lds_dst) : "memory");
}
// wait until at most n vector-memory instructions of this wave are outstanding (s_waitcnt takes an immediate: the
// host rounds n down to one of these values; only entries that read a prefetched child come here)
__device__ __forceinline__ void walk4_wait_vm(unsigned n)
{
#define MBAMD_W4_WAIT(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        MBAMD_W4_WAIT(1) MBAMD_W4_WAIT(2) MBAMD_W4_WAIT(3) MBAMD_W4_WAIT(4) MBAMD_W4_WAIT(5) MBAMD_W4_WAIT(6)
        MBAMD_W4_WAIT(8) MBAMD_W4_WAIT(10) MBAMD_W4_WAIT(12) MBAMD_W4_WAIT(16) MBAMD_W4_WAIT(20) MBAMD_W4_WAIT(24)
        MBAMD_W4_WAIT(32) MBAMD_W4_WAIT(40) MBAMD_W4_WAIT(48)
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
    const unsigned ok[] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48};
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

// byte-offset addressing helpers (wave-uniform base + 32-bit offset: two scalar adds)
template <class T> __device__ __forceinline__ T* walk4_at(T* base, unsigned byteOffset)
{
    return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(base) + byteOffset);
}

// A short program (a root-ward path: the partial update of an MCMC generation) travels in the kernel arguments instead of
// a device buffer: no copy kernel in front of the walk, nothing to keep alive until the launch has run.
#define MBAMD_W4_INLINE 96       // entries (3 KiB of the 4 KiB argument block)
struct Walk4ArgsInline {
    Walk4Args a;
    Walk4Entry inl[MBAMD_W4_INLINE];
};
__device__ __forceinline__ const Walk4Args& walk4_args(const Walk4Args& a) { return a; }
__device__ __forceinline__ const Walk4Args& walk4_args(const Walk4ArgsInline& a) { return a.a; }
__device__ __forceinline__ const Walk4Entry* walk4_program(const Walk4Args& a) { return a.prog; }
#if defined(MBAMD_HOST_EMU)
__device__ __forceinline__ const Walk4Entry* walk4_program(const Walk4ArgsInline& a) { return a.inl; }
#else
// (the address of a by-value kernel parameter would be that of a private copy: read the argument block itself)
__device__ __forceinline__ const Walk4Entry* walk4_program(const Walk4ArgsInline&)
{
    return reinterpret_cast<const Walk4Entry*>((uintptr_t) __builtin_amdgcn_kernarg_segment_ptr() + offsetof(Walk4ArgsInline, inl));
}
#endif

// blockDim.x = 64 * W; grid = walk4_grid(nblocks, K) workgroups.  Dynamic LDS: walk4_lds_bytes(W, nslots).
// ARGS = Walk4Args (program in a device buffer) or Walk4ArgsInline (program in the arguments).
// TIPPF: the tip bitplanes of entry j + tipAhead are touched at the top of iteration j (full-tree evaluations: a tip's planes
// are read once per launch, i.e. from HBM -- a round trip of microseconds under a saturated write stream, and the scalar load
// that takes it is waited for one iteration after it was issued).
template <class ARGS, bool TIPPF = false>
__global__ void __launch_bounds__(64 * MBAMD_W4_MAXW)
k_walk4_t(ARGS AA)
{
    const Walk4Args& A = walk4_args(AA);
    const unsigned lane = threadIdx.x & 63;
#if defined(MBAMD_HOST_EMU)
    const int wave = (int) (threadIdx.x >> 6);
    char* lds = reinterpret_cast<char*>(mbamd_emu_dyn_lds());
#else
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    extern __shared__ f4 lds_walk4[];
    char* lds = reinterpret_cast<char*>(lds_walk4);
#endif
    const unsigned K = (unsigned) A.K;
    // workgroup id -> (pattern block, category): see walk4_grid
    const unsigned xcd = blockIdx.x & 7u, pos = blockIdx.x >> 3;
    const unsigned blk = (pos / K) * 8u + xcd, k = pos % K;
    if (blk >= (unsigned) A.nblocks) return;
    char* const mine = lds + (size_t) wave * (MBAMD_W4_STAGE + (size_t) A.nslots * 1024);
    int* const stage = reinterpret_cast<int*>(mine);                           // [2][64] landing areas of stored exponents
    char* const slots = mine + MBAMD_W4_STAGE + lane * 16;                     // this wave's private slots, this lane's f4
    // this wave's columns: wave-uniform bases, the entries hold byte offsets from them
    f4* const P0 = A.partials + (size_t) blk * A.pstride + (size_t) k * 64;
    const uint64_t* const T0 = A.tips + (size_t) blk * A.tstride;
    int8_t* const E0 = A.exps + (size_t) blk * A.estride + (size_t) k * 64;
    const float* const M0 = A.matrices + (size_t) k * 16;
#if defined(MBAMD_HOST_EMU)
#define MBAMD_W4_PREFETCH(SRC, DST) walk4_dma(walk4_at(P0, SRC), lane, reinterpret_cast<f4*>(slots - lane * 16 + (DST)))
#define MBAMD_W4_EXPS(OFF, PARITY) walk4_dma_exps(walk4_at(E0, OFF), lane, stage + 64 * (PARITY))
#else
    const unsigned lane16 = lane * 16u;
    const unsigned stage_lds = (unsigned) (uintptr_t) (__attribute__((address_space(3))) char*) mine;
    const unsigned slots_lds = stage_lds + MBAMD_W4_STAGE;
#define MBAMD_W4_PREFETCH(SRC, DST) walk4_dma(walk4_at(P0, SRC), lane16, slots_lds + (DST))
#define MBAMD_W4_EXPS(OFF, PARITY) walk4_dma_exps(walk4_at(E0, OFF), lane, stage_lds + 256u * (PARITY))
#endif

    const Walk4Entry* prog = walk4_program(AA) + (size_t) wave * A.entries;
    const int n = A.entries - A.tail;
    Walk4Entry DA = walk4_load_entry(prog), DB = walk4_load_entry(prog + 1);
    const int ahead = TIPPF ? A.tipAhead : 0;
    Walk4Half FAR = walk4_load_half(prog + ahead);               // (TIPPF) entry j + ahead, loaded during iteration j - 1
#if defined(MBAMD_HOST_EMU)
    int* const junk = stage + 128;
    const unsigned vzero = 0;
#else
    const unsigned junk = stage_lds + 512u;
    unsigned vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
#endif
    // inputs of entry 0
    Walk4Mat M1 = walk4_load_matrix(walk4_at(M0, DA.m1));
    Walk4Mat M2 = walk4_load_matrix(walk4_at(M0, DA.m2));
    Walk4Planes T1 = walk4_load_planes(walk4_at(T0, (DA.ctl & MBAMD_W4_TIP1) ? DA.c1 : 0u));
    Walk4Planes T2 = walk4_load_planes(walk4_at(T0, (DA.ctl & MBAMD_W4_TIP2) ? DA.c2 : 0u));
    if (((DA.ctl >> 8) & 3u) == SCALE_READ) MBAMD_W4_EXPS(DA.eread, 0);
    int cum_e = 0;
    f4 prev = {0.0f, 0.0f, 0.0f, 0.0f};            // the result of the operation executed last (FWD1 / FWD2 children)

    // One iteration = one entry.  Vector-memory instruction sequence (the host's vmwait counts on exactly this):
    //     [DMA pf0] [DMA pf1] (PF entries)   s_waitcnt vmcnt(vmwait) (flag VMWAIT)   [2 tip touches (TIPPF kernels)]
    //     [exponent DMA for the next entry, if that is SCALE_READ]   [2 stores, if this entry is an operation]
    // Everything wave-uniform is a scalar branch or a scalar select (no divergent control flow).  The loop is unrolled by
    // two so that the two entry descriptors in flight keep their registers (no moves): `cur` is executed, `nxt` is the
    // next one, and cur's registers receive entry j + 2.  ALL scalar loads of an iteration (next entry's matrices and tip
    // planes, the entry after next) are issued in one burst as soon as this entry's matrix products are done, and are
    // consumed after the next iteration's LDS reads: one lgkmcnt(0) per iteration covers both.
    auto step = [&](Walk4Entry& cur, const Walk4Entry& nxt, int j, int parity) {
        const unsigned ctl = cur.ctl;
        bool run = true;
        if (ctl & MBAMD_W4_RARE) {
            if (ctl & MBAMD_W4_PF0) {
                // PF entry: children of later operations that live in HBM -> LDS slots
                MBAMD_W4_PREFETCH(cur.dst, cur.c1);
                if (ctl & MBAMD_W4_PF1) MBAMD_W4_PREFETCH(cur.c2, cur.m1);
            }
            if (ctl & MBAMD_W4_VMWAIT) walk4_wait_vm((ctl >> 10) & 63u);       // what an LDS-DMA brought for this entry has landed
            if (ctl & MBAMD_W4_BARRIER) walk4_barrier();
            run = !(ctl & MBAMD_W4_NOP);
        }
        const unsigned mode = (ctl >> 8) & 3u;
        f4 out = {0.0f, 0.0f, 0.0f, 0.0f};
        int er = 0;
        if (run) {
            f4 a, b;
            if (ctl & MBAMD_W4_TIP1) a = walk4_tip_vector(T1, lane);
            else if (ctl & MBAMD_W4_FWD1) a = prev;
            else a = *reinterpret_cast<const f4*>(slots + cur.c1);
            if (ctl & MBAMD_W4_TIP2) b = walk4_tip_vector(T2, lane);
            else if (ctl & MBAMD_W4_FWD2) b = prev;
            else b = *reinterpret_cast<const f4*>(slots + cur.c2);
            if (mode == SCALE_READ) er = stage[64 * parity + lane];
            const f4 f1 = walk4_matvec(M1, a);
            const f4 f2 = walk4_matvec(M2, b);
            out.x = f1.x * f2.x; out.y = f1.y * f2.y; out.z = f1.z * f2.z; out.w = f1.w * f2.w;
        }
        // (TIPPF) touch the tip planes of the entry tipAhead further on.  Here -- behind the matrix products -- and not at the
        // top of the iteration: FAR came with the previous burst, and the first instruction that reads anything of a burst
        // waits for ALL of it (one lgkmcnt, scalar loads return out of order); the products have paid for that wait already
        if (TIPPF) {
            walk4_touch_planes(walk4_at(T0, (FAR.ctl & MBAMD_W4_TIP1) ? FAR.c1 : 0u), walk4_at(T0, (FAR.ctl & MBAMD_W4_TIP2) ? FAR.c2 : 0u), vzero, junk);
        }
        // the scalar-load burst for the next entry (the registers of this entry's matrices / planes are free now); a
        // child that is not a tip reads the planes at offset 0 -- a valid address, the value is not used
        const unsigned dst = cur.dst, ewrite = cur.ewrite;
#if defined(MBAMD_W4X_MAT0)      // (timing experiments: every matrix / every tip from one hot line -- wrong values)
        M1 = walk4_load_matrix(walk4_at(M0, 0u));
        M2 = walk4_load_matrix(walk4_at(M0, 0u));
#else
        M1 = walk4_load_matrix(walk4_at(M0, nxt.m1));
        M2 = walk4_load_matrix(walk4_at(M0, nxt.m2));
#endif
#if defined(MBAMD_W4X_TIP0)
        T1 = walk4_load_planes(walk4_at(T0, 0u));
        T2 = walk4_load_planes(walk4_at(T0, 0u));
#else
        T1 = walk4_load_planes(walk4_at(T0, (nxt.ctl & MBAMD_W4_TIP1) ? nxt.c1 : 0u));
        T2 = walk4_load_planes(walk4_at(T0, (nxt.ctl & MBAMD_W4_TIP2) ? nxt.c2 : 0u));
#endif
        if (((nxt.ctl >> 8) & 3u) == SCALE_READ) MBAMD_W4_EXPS(nxt.eread, parity ^ 1);
        cur = walk4_load_entry(prog + j + 2);
        if (TIPPF) FAR = walk4_load_half(prog + j + 1 + ahead);
        if (run) {
            const int wm = mode == SCALE_WRITE ? -1 : 0, rm = mode == SCALE_READ ? -1 : 0;
            const int e = (scale_exponent(max4(out)) & wm) | (er & rm);
            cum_e += e & wm;
            out.x = scale_pow2(out.x, -e); out.y = scale_pow2(out.y, -e);       // (2^0 is exact: no branch)
            out.z = scale_pow2(out.z, -e); out.w = scale_pow2(out.w, -e);
            prev = out;
            if (ctl & MBAMD_W4_KEEP) *reinterpret_cast<f4*>(slots + ((ctl >> 6) & 0x3FC00u)) = out;
#if defined(MBAMD_HOST_EMU)
            walk4_at(P0, dst)[lane] = out;
            walk4_at(E0, ewrite)[lane] = (int8_t) e;
#else
            // 1 KiB contiguous per wave; never waited for.  Non-temporal: the result is not read again in this launch
            // (parents read the LDS copy), so it must not push the matrices and programs out of L2
#if defined(MBAMD_W4X_NOSTORE)
            if (e == 12345) {
#endif
            __builtin_nontemporal_store(out, as_global(walk4_at(P0, dst)) + lane);
            __builtin_nontemporal_store((int8_t) e, as_global(walk4_at(E0, ewrite)) + lane);
#if defined(MBAMD_W4X_NOSTORE)
            }
#endif
#endif
        }
    };
    for (int j = 0; j < n; j += 2) {
        step(DA, DB, j, 0);
        step(DB, DA, j + 1, 1);
    }
#undef MBAMD_W4_EXPS
#undef MBAMD_W4_PREFETCH
    // cumulative exponents of this workgroup's 64 columns: the waves' sums meet in LDS, wave 0 owns the memory update
    // (nobody else touches these 64 entries: no atomics; a fresh buffer is simply stored)
    if (A.cum != nullptr) {
        const int W = (int) (blockDim.x >> 6);
        if (W > 1) {
            stage[lane] = cum_e;
            walk4_barrier();
            if (wave != 0) return;
            cum_e = 0;
            for (int w = 0; w < W; ++w)
                cum_e += reinterpret_cast<const int*>(lds + (size_t) w * (MBAMD_W4_STAGE + (size_t) A.nslots * 1024))[lane];
        }
        int32_t* dst = A.cum + (size_t) k * A.Ppad + (size_t) blk * 64 + lane;
        if (A.cumFresh) *dst = cum_e;
        else if (cum_e != 0) *dst += cum_e;
    }
}

}  // namespace mbamd
#endif
