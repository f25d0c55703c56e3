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
#define MBAMD_W4_READS      0x00000200u  // (= ScaleMode SCALE_READ in [9:8]) this entry divides by stored exponents: they were fetched into the landing area
#define MBAMD_W4_NEXT_READS 0x04000000u  // the NEXT entry of this wave is SCALE_READ: fetch its stored exponents now (4-state walk; set by the host)
#define MBAMD_W4_RARE     (MBAMD_W4_NOP | MBAMD_W4_BARRIER | MBAMD_W4_PF0 | MBAMD_W4_VMWAIT | MBAMD_W4_READS | MBAMD_W4_NEXT_READS)
// (Round 6: a ring of eight landing areas with the exponents requested six entries ahead was built on the hypothesis that the wait for
//  the DMA is a wait for the previous entry's stores -- it was not the reason SCALE_READ evaluations were slow (the scratch row was,
//  below) and cost the SCALE_WRITE walk 1.6 % at DNA 1000 x 50 000: taken out again, profiles/r06_scale_read.txt)
#define MBAMD_W4_SCRATCH_ROWS 32   // exponent rows behind the scale buffers: where entries that record no exponents store their byte, in rotation
#define MBAMD_W4_MAXW     8

// One step of a wave's program (wave-uniform; fetched with one s_load_dwordx8).  Addresses are ready-made byte
// offsets from a base the wave computes once (scalar adds only, no multiplications in the loop).
struct alignas(32) Walk4Entry {
    uint32_t ctl;      // [7:0] flags   [9:8] ScaleMode   [15:10] vmwait   [23:16] slot that keeps the result (flag KEEP)   [25:24] FWD1 / FWD2   [26] NEXT_READS
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
    int tail;                    // trailing NOP entries of every program (read-ahead): 2
};

#define MBAMD_W4_STAGE 768       // bytes per wave in front of its slots: two 64-dword landing areas for stored exponents + one nobody reads
__host__ __device__ inline size_t walk4_lds_bytes(int W, int nslots) { return (size_t) W * (MBAMD_W4_STAGE + (size_t) nslots * 1024); }
// The grid is one-dimensional and XCD-aware: workgroup id -> XCD id % 8 (observed dispatch rule), and the K category
// workgroups of one pattern block get consecutive positions on ONE XCD, so that the tip bitplanes and the matrix
// lines they all read are fetched into that XCD's L2 once.
__host__ __device__ inline unsigned walk4_grid(int nblocks, int K) { return 8u * (unsigned) K * (unsigned) ((nblocks + 7) / 8); }

struct Walk4Planes { uint64_t p[4]; };

}  // namespace mbamd
#include <mbamd_dev_walk4.h>     // Walk4Mat, walk4_load_* / walk4_tip_vector / walk4_dma* / walk4_wait_vm / walk4_barrier / walk4_matvec / ... (csrc/device/)
namespace mbamd {

// the values walk4_wait_vm implements, for the host: the largest supported count <= n
__host__ __device__ inline unsigned walk4_round_wait(long n)
{
    const unsigned ok[] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48};
    unsigned r = 0;
    for (unsigned v : ok) if ((long) v <= n) r = v;
    return r;
}

// byte-offset addressing helpers (wave-uniform base + 32-bit offset: two scalar adds)
// (partials: the arena is BUFFER-major -- a buffer is P_pad/64 x K KiB, up to several MB -- and the entries hold its offset in KiB)
template <class T> __device__ __forceinline__ T* walk4_at_kib(T* base, unsigned kib)
{
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + ((size_t) kib << 10));
}
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
}  // namespace mbamd
#include <mbamd_dev_walk4_args.h>   // walk4_program(const Walk4ArgsInline&): the program inside the kernel arguments
namespace mbamd {

// blockDim.x = 64 * W; grid = walk4_grid(nblocks, K) workgroups.  Dynamic LDS: walk4_lds_bytes(W, nslots).
// ARGS = Walk4Args (program in a device buffer) or Walk4ArgsInline (program in the arguments).
//
// Round 4: the loop was issue-bound (profiles/r03_c4_pmc.txt: 50 VALU + 58 SALU + 6 SMEM + 4 VMEM instructions per operation and
// wave, the scalar side -- shared by the four SIMDs of a CU -- the larger half).  What left the common path:
//   * the tip-plane touch (two LDS-DMAs, a half-entry scalar load and their address arithmetic per operation: +-2 % in round 3);
//   * everything that is not "an operation that rescales or does not": prefetch entries, waits, barriers, no-ops AND the two
//     halves of dynamic rescaling's SCALE_READ pass (fetch the next entry's stored exponents / read this entry's) sit behind ONE
//     test of the entry's flags (MBAMD_W4_RARE; the host marks the entry in front of a SCALE_READ entry with NEXT_READS);
//   * the copy of a result into the forwarding registers (results alternate between two register sets with the loop's two halves);
//   * per-entry address arithmetic of the program (a running pointer, one add per two entries).
template <class ARGS>
__global__ void __launch_bounds__(64 * MBAMD_W4_MAXW)
k_walk4_t(ARGS AA)
{
    const Walk4Args& A = walk4_args(AA);
    const unsigned lane = threadIdx.x & 63;
    const int wave = mbd_wave_index();
    char* lds = mbd_dyn_lds<char>();
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
    const Walk4Lds L = walk4_lds(mine, lane);                                    // the same window for the LDS-DMA forms
#define MBAMD_W4_PREFETCH(SRC, DST) walk4_prefetch(L, walk4_at_kib(P0, SRC), (DST))
#define MBAMD_W4_EXPS(OFF, PARITY) walk4_fetch_exps(L, walk4_at(E0, OFF), lane, (PARITY))

    const Walk4Entry* pp = walk4_program(AA) + (size_t) wave * A.entries;       // entry j of this wave's program
    const int n = A.entries - A.tail;
    Walk4Entry DA = walk4_load_entry(pp), DB = walk4_load_entry(pp + 1);
    // inputs of entry 0
    Walk4Mat M1 = walk4_load_matrix(walk4_at(M0, DA.m1));
    Walk4Mat M2 = walk4_load_matrix(walk4_at(M0, DA.m2));
    Walk4Planes T1 = walk4_load_planes(walk4_at(T0, (DA.ctl & MBAMD_W4_TIP1) ? DA.c1 : 0u));
    Walk4Planes T2 = walk4_load_planes(walk4_at(T0, (DA.ctl & MBAMD_W4_TIP2) ? DA.c2 : 0u));
    if (DA.ctl & MBAMD_W4_READS) MBAMD_W4_EXPS(DA.eread, 0);
    int cum_e = 0;
    f4 RA = {0.0f, 0.0f, 0.0f, 0.0f}, RB = RA;     // results of the even / odd entries (a no-op entry passes the previous one through): a FWD child reads the other set

    // One iteration = one entry.  Vector-memory instruction sequence (the host's vmwait counts on exactly this):
    //     [DMA pf0] [DMA pf1] (PF entries)   s_waitcnt vmcnt(vmwait) (flag VMWAIT)
    //     [exponent DMA for the next entry, if that is SCALE_READ]   [2 stores, if this entry is an operation]
    // Everything wave-uniform is a scalar branch or a scalar select (no divergent control flow).  The loop is unrolled by
    // two so that the two entry descriptors in flight keep their registers (no moves): `cur` is executed, `nxt` is the
    // next one, and cur's registers receive entry j + 2.  ALL scalar loads of an iteration (next entry's matrices and tip
    // planes, the entry after next) are issued in one burst as soon as this entry's matrix products are done, and are
    // consumed after the next iteration's LDS reads: one lgkmcnt(0) per iteration covers both.
    auto step = [&](Walk4Entry& cur, const Walk4Entry& nxt, const Walk4Entry* after, int parity, f4& out, const f4& prev) {
        const unsigned ctl = cur.ctl;
        int er = 0;
        if (ctl & MBAMD_W4_RARE) {
            if (ctl & MBAMD_W4_PF0) {
                // PF entry: children of later operations that live in HBM -> LDS slots
                MBAMD_W4_PREFETCH(cur.dst, cur.c1);
                if (ctl & MBAMD_W4_PF1) MBAMD_W4_PREFETCH(cur.c2, cur.m1);
            }
            if (ctl & MBAMD_W4_VMWAIT) walk4_wait_vm((ctl >> 10) & 63u);       // what an LDS-DMA brought for this entry has landed
            if (ctl & MBAMD_W4_BARRIER) walk4_barrier();
            if (ctl & MBAMD_W4_NEXT_READS) MBAMD_W4_EXPS(nxt.eread, parity ^ 1);
            if (ctl & MBAMD_W4_READS) er = stage[64 * parity + lane];
        }
        f4 o = prev;                                   // (a no-op entry passes the result of the operation executed last through)
        if (!(ctl & MBAMD_W4_NOP)) {
            f4 a, b;
            if (ctl & MBAMD_W4_TIP1) a = walk4_tip_vector(T1, lane);
            else if (ctl & MBAMD_W4_FWD1) a = prev;
            else a = *reinterpret_cast<const f4*>(slots + cur.c1);
            if (ctl & MBAMD_W4_TIP2) b = walk4_tip_vector(T2, lane);
            else if (ctl & MBAMD_W4_FWD2) b = prev;
            else b = *reinterpret_cast<const f4*>(slots + cur.c2);
            const f4 f1 = walk4_matvec(M1, a);
            const f4 f2 = walk4_matvec(M2, b);
            o.x = f1.x * f2.x; o.y = f1.y * f2.y; o.z = f1.z * f2.z; o.w = f1.w * f2.w;
        }
        // the scalar-load burst for the next entry (the registers of this entry's matrices / planes are free now); a
        // child that is not a tip reads the planes at offset 0 -- a valid address, the value is not used
        const unsigned dst = cur.dst, ewrite = cur.ewrite;
        M1 = walk4_load_matrix(walk4_at(M0, nxt.m1));
        M2 = walk4_load_matrix(walk4_at(M0, nxt.m2));
        T1 = walk4_load_planes(walk4_at(T0, (nxt.ctl & MBAMD_W4_TIP1) ? nxt.c1 : 0u));
        T2 = walk4_load_planes(walk4_at(T0, (nxt.ctl & MBAMD_W4_TIP2) ? nxt.c2 : 0u));
        cur = walk4_load_entry(after);
        if (!(ctl & MBAMD_W4_NOP)) {
            // SCALE_WRITE: this column's own power of two; SCALE_READ (rare path): the stored one; else 2^0 (exact: no branch)
            const int wm = (int) (ctl << 23) >> 31;                                    // mode bit 8 (SCALE_WRITE) -> all ones
            const int ew = scale_exponent(max4(o)) & wm;
            const int e = ew | er;
            cum_e += ew;
            o.x = scale_pow2(o.x, -e); o.y = scale_pow2(o.y, -e);
            o.z = scale_pow2(o.z, -e); o.w = scale_pow2(o.w, -e);
            if (ctl & MBAMD_W4_KEEP) *reinterpret_cast<f4*>(slots + ((ctl >> 6) & 0x3FC00u)) = o;
            // (two stores, always: a branch around the second cost the SCALE_WRITE walk 5 % (call 31).  An entry that records no exponents writes its
            //  byte to one of MBAMD_W4_SCRATCH_ROWS scratch rows, in rotation: every entry of a SCALE_READ evaluation writing to ONE row
            //  made such an evaluation a third slower -- same-address stores, profiles/r06_scale_read.txt)
            walk4_store(walk4_at_kib(P0, dst), walk4_at(E0, ewrite), lane, o, e);
        }
        out = o;
    };
    for (int j = 0; j < n; j += 2) {
        step(DA, DB, pp + 2, 0, RA, RB);
        step(DB, DA, pp + 3, 1, RB, RA);
        pp += 2;
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


// ---------------------------------------------------------------------------------------------------------------------------------
// k_path4 -- a ROOT-WARD PATH (round 5): the list of a move that dirtied one branch.  Every operation has the previous result as one
// child and a "sibling" the list does not write (a compact tip, or a buffer of an earlier launch) as the other:
//     out_j = rescale ((M1_j out_{j-1}) o (M2_j sib_j)).
// Only the first factor depends on the chain.  k_walk4_t ran such a program entry by entry -- two matrix loads, a prefetched sibling,
// two products, a store: ~1 us = 2 200 cycles per operation (profiles/r04_mcmc_fixed_topology.txt: 17-27 operations in 20-22 us), almost
// all of it the scalar-load round trip of the entry's matrices in front of work that needs the previous result anyway.  Here a wave takes
// the path in chunks of MBAMD_P4_CHUNK operations and each chunk in two passes:
//   1. every sibling that lives in HBM and every stored exponent of the chunk is requested back to back (no store sits between them
//      in the vmcnt order), then the sibling factors F_j = M2_j sib_j are formed -- independent work, the matrices in bursts of four;
//   2. the chain: out_j = rescale ((M1_j out_{j-1}) o F_j), matrices in bursts of four -- four dependent v_pk_fma_f32 steps, a product,
//      a maximum, the exponent, two multiplications and a store that nobody waits for.
// No LDS, no slots, no barrier; F lives in registers.  Same arithmetic, operation by operation, as k_walk4_t: the same bits.
// Entries (Walk4Entry): c1 = the chain's INPUT (entry 0 only: tip planes or a buffer), c2 = the sibling, m1 / m2 their matrices;
// ctl: TIP1 (entry 0: the input is a compact tip), TIP2 (the sibling is one), [9:8] the scale mode.  blockDim.x = 64, grid = walk4_grid,
// dynamic LDS = path4_lds_bytes(entries).
//
// FORKED PATHS (round 6): the list of a topology move (NNI / SPR / TBR) is two root-ward paths that JOIN -- in the list's post-order:
// arm A, arm B, the operation whose children are the two arms' last results, the common stem.  The program is a sequence of ARMS:
// an entry flagged START begins one (its c1 is the arm's input, as entry 0's; ctl[23:16] = the arm's length) and, if it is not entry 0,
// the running result is SAVED in registers first; an entry flagged JOIN has the saved result as its sibling.  One saved result at a
// time (the host compiles anything deeper for k_walk4_t).  Chunks do not cross arms.
#define MBAMD_P4_START 0x08000000u
#define MBAMD_P4_JOIN  0x10000000u
#if !defined(MBAMD_P4_CHUNK)
#define MBAMD_P4_CHUNK 24         // (8: 17.8 us, 16: 16.6, 24: 15.9, 32: 16.3 -- median of the paths of DNA 500 x 20 000, profiles/r05_path4.txt)
#endif
#define MBAMD_P4_GROUP 4         // matrices per burst of scalar loads (4 x 16 scalar registers)
// the path itself: `prev` starts as the chain's input and ends as the last result; returns the exponents this wave's column gained
// `MS`: this wave's LDS area for the matrices of a chunk (MBAMD_P4_MATRIX_BYTES).  Round 6: the 2 x 24 matrices of a chunk came through
// the scalar cache in bursts of four -- twelve dependent round trips per chunk on a wave that is alone on its SIMD; now eight lanes per
// entry fetch them (three vector loads per chunk, in front of the sibling loads), they wait in LDS and every use is four broadcast reads.
#define MBAMD_P4_MATRIX_BYTES (MBAMD_P4_CHUNK * 2 * 64)
__device__ __forceinline__ int walk4_path_run(const Walk4Entry* pp, int n, unsigned lane, f4* P0, const uint64_t* T0, int8_t* E0, const float* M0, f4& prev, f4* MS)
{
    constexpr int C = MBAMD_P4_CHUNK;
    static_assert(C % 8 == 0, "eight entries' matrices per vector load");
    int cum_e = 0;
    f4 saved = {0.0f, 0.0f, 0.0f, 0.0f};
    prev = saved;
    for (int arm0 = 0, arm1 = 0; arm0 < n; arm0 = arm1) {
    {
        const Walk4Entry e0 = walk4_entry_from_lds(pp + arm0);
        arm1 = arm0 + (int) ((e0.ctl >> 16) & 0xFFu);
        saved = prev;                                 // (the first arm saves nothing anyone reads)
        if (e0.ctl & MBAMD_W4_TIP1) prev = walk4_tip_vector(walk4_load_planes(walk4_at(T0, e0.c1)), lane);
        else prev = walk4_at_kib(P0, e0.c1)[lane];
    }
    for (int base = arm0; base < arm1; base += C) {
        const int cnt = arm1 - base < C ? arm1 - base : C;
        const Walk4Entry* const q = pp + base;
        f4 F[C];
        int er[C];
        // 1a. what the chunk reads from HBM, back to back -- its matrices first: lane -> (entry 8 r + lane / 8, child (lane / 4) & 1, quarter lane & 3)
        f4 mq[C / 8];
#pragma unroll
        for (int r = 0; r < C / 8; ++r) {
            const int i = 8 * r + (int) (lane >> 3);
            mq[r] = f4{0.0f, 0.0f, 0.0f, 0.0f};
            if (i < cnt) {
                const Walk4Entry* const ep = q + i;
                const unsigned off = (lane & 4u) ? ep->m2 : ep->m1;
                mq[r] = walk4_load_f4(reinterpret_cast<const f4*>(walk4_at(M0, off)) + (lane & 3u));
            }
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            er[i] = 0;
            if (i < cnt) {
                const Walk4Entry e = walk4_entry_from_lds(q + i);
#if defined(MBAMD_P4_ABL_NO_LOAD)
                F[i] = f4{0.25f, 0.25f, 0.25f, 0.25f};
#else
                if (!(e.ctl & (MBAMD_W4_TIP2 | MBAMD_P4_JOIN))) F[i] = walk4_at_kib(P0, e.c2)[lane];
#endif
                if (e.ctl & MBAMD_W4_READS) er[i] = walk4_at(E0, e.eread)[lane];
            }
        }
        // (the matrices land first -- loads return in order -- and go to LDS: [entry][child][16 floats])
#pragma unroll
        for (int r = 0; r < C / 8; ++r) MS[(size_t) r * 64 + lane] = mq[r];
        walk4_wave_lds_fence();
        // 1b. the sibling factors (nothing here depends on the chain)
        constexpr int G = MBAMD_P4_GROUP;
#pragma unroll
        for (int g = 0; g < C; g += G) {
            if (g < cnt) {
                Walk4Entry e[G];
                Walk4Mat M[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    e[u] = walk4_entry_from_lds(q + (g + u < cnt ? g + u : g));
                    M[u] = walk4_matrix_from_lds(MS + (size_t) ((g + u < cnt ? g + u : g) * 2 + 1) * 4);
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    if (g + u < cnt) {
                        f4 v = F[g + u];
                        if (e[u].ctl & MBAMD_W4_TIP2) v = walk4_tip_vector(walk4_load_planes(walk4_at(T0, e[u].c2)), lane);   // (a tip sibling: once or twice per path)
                        if (e[u].ctl & MBAMD_P4_JOIN) v = saved;                     // (the other arm's last result)
                        F[g + u] = walk4_matvec(M[u], v);
                    }
                }
            }
        }
        // 2. the chain
#pragma unroll
        for (int g = 0; g < C; g += G) {
            if (g < cnt) {
                Walk4Entry e[G];
                Walk4Mat M[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    e[u] = walk4_entry_from_lds(q + (g + u < cnt ? g + u : g));
                    M[u] = walk4_matrix_from_lds(MS + (size_t) ((g + u < cnt ? g + u : g) * 2) * 4);
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    if (g + u < cnt) {
                        const f4 f1 = walk4_matvec(M[u], prev);
                        const f4 f2 = F[g + u];
                        f4 o;
                        o.x = f1.x * f2.x; o.y = f1.y * f2.y; o.z = f1.z * f2.z; o.w = f1.w * f2.w;
                        const int wm = (int) (e[u].ctl << 23) >> 31;               // mode bit 8 (SCALE_WRITE) -> all ones
                        const int ew = scale_exponent(max4(o)) & wm;
                        const int ex = ew | er[g + u];
                        cum_e += ew;
                        o.x = scale_pow2(o.x, -ex); o.y = scale_pow2(o.y, -ex);
                        o.z = scale_pow2(o.z, -ex); o.w = scale_pow2(o.w, -ex);
#if defined(MBAMD_P4_ABL_NO_STORE)
                        if (o.x == 123.456f) walk4_store_partials(walk4_at_kib(P0, e[u].dst), lane, o);
#else
                        if (wm) walk4_store(walk4_at_kib(P0, e[u].dst), walk4_at(E0, e[u].ewrite), lane, o, ex);
                        else walk4_store_partials(walk4_at_kib(P0, e[u].dst), lane, o);
#endif
                        prev = o;
                    }
                }
            }
        }
        walk4_wave_lds_fence();                      // (the next chunk's matrices overwrite these)
    }
    }
    return cum_e;
}

__host__ __device__ inline size_t path4_program_bytes(int entries) { return (size_t) ((entries * 32 + 63) / 64 * 64); }
__host__ __device__ inline size_t path4_lds_bytes(int entries) { return path4_program_bytes(entries) + MBAMD_P4_MATRIX_BYTES; }
template <class ARGS>
__global__ void __launch_bounds__(64)
k_path4(ARGS AA)
{
    const Walk4Args& A = walk4_args(AA);
    const unsigned lane = threadIdx.x & 63;
    const unsigned K = (unsigned) A.K;
    const unsigned xcd = blockIdx.x & 7u, pos = blockIdx.x >> 3;
    const unsigned blk = (pos / K) * 8u + xcd, k = pos % K;
    if (blk >= (unsigned) A.nblocks) return;
    f4* const P0 = A.partials + (size_t) blk * A.pstride + (size_t) k * 64;
    const uint64_t* const T0 = A.tips + (size_t) blk * A.tstride;
    int8_t* const E0 = A.exps + (size_t) blk * A.estride + (size_t) k * 64;
    const float* const M0 = A.matrices + (size_t) k * 16;
    const int n = A.entries;
    // the program: from the kernel arguments (host-visible memory) into LDS, one vector load per 32 entries
    Walk4Entry* const pp = mbd_dyn_lds<Walk4Entry>();
    walk4_program_to_lds(walk4_program(AA), pp, n, lane);
    f4 prev;
    const int cum_e = walk4_path_run(pp, n, lane, P0, T0, E0, M0, prev, reinterpret_cast<f4*>(reinterpret_cast<char*>(pp) + path4_program_bytes(n)));
    if (A.cum != nullptr) {
        int32_t* dst = A.cum + (size_t) k * A.Ppad + (size_t) blk * 64 + lane;
        if (A.cumFresh) *dst = cum_e;
        else if (cum_e != 0) *dst += cum_e;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_path4_lnl -- the path AND the log-likelihood behind it in one launch (round 6).  A generation of a fixed-topology chain is: one or
// two new matrices, the root-ward path of the changed branch, the integration over the root branch -- three launches of 5 + 16 + 6 us
// with two gaps of 4-5 us between them (profiles/r05_path4.txt).  The path's last result IS the integration's parent: a workgroup here
// is the K category waves of one 64-pattern block; each wave walks the path as k_path4 does, leaves its last result and its column's
// cumulative exponent in LDS, the workgroup meets once, and wave 0 integrates the block -- the arithmetic of k_integrate_lnl_s4, term
// by term: the same site values, the same block sum.  blockDim.x = 64 K (K <= 8), grid = 8 ceil(nblocks / 8) (XCD-aware as
// walk4_grid), dynamic LDS = path4_lnl_lds_bytes(entries, K).
struct PathLnl4 {
    const void*    child;          // the root tip (state bitplanes) / the child's partials / nullptr (root integration)
    const float*   matrix;         // the root branch's matrices [K][4][4] transposed
    const double*  weights;        // K category weights
    const double*  freqs;          // 4 state frequencies
    const int32_t* cum;            // the integration's cumulative exponents [K][Ppad] or nullptr (may be the buffer the path adds to)
    const double*  pattern_weights;
    double*        site;
    double*        wsite;
    int            P;
    int            child_kind;     // CHILD_STATES / CHILD_PARTIALS
};
__host__ __device__ inline size_t path4_lnl_lds_bytes(int entries, int K) { return (size_t) ((entries * 32 + 63) / 64 * 64) + (size_t) K * 64 * (sizeof(f4) + sizeof(int)) + (size_t) K * MBAMD_P4_MATRIX_BYTES; }
template <class ARGS>
__global__ void __launch_bounds__(512)
k_path4_lnl(ARGS AA, PathLnl4 t)
{
    const Walk4Args& A = walk4_args(AA);
    const unsigned lane = threadIdx.x & 63;
    const unsigned K = (unsigned) A.K;
    const unsigned k = (unsigned) mbd_wave_index();
    const unsigned blk = (blockIdx.x >> 3) * 8u + (blockIdx.x & 7u);
    if (blk >= (unsigned) A.nblocks) return;
    f4* const P0 = A.partials + (size_t) blk * A.pstride + (size_t) k * 64;
    const uint64_t* const T0 = A.tips + (size_t) blk * A.tstride;
    int8_t* const E0 = A.exps + (size_t) blk * A.estride + (size_t) k * 64;
    const float* const M0 = A.matrices + (size_t) k * 16;
    const int n = A.entries;
    // (every wave copies the program to the same place: the same bytes)
    Walk4Entry* const pp = mbd_dyn_lds<Walk4Entry>();
    walk4_program_to_lds(walk4_program(AA), pp, n, lane);
    f4 prev;
    char* const xb = reinterpret_cast<char*>(pp) + (size_t) ((n * 32 + 63) / 64 * 64);
    const int cum_e = walk4_path_run(pp, n, lane, P0, T0, E0, M0, prev,
                                     reinterpret_cast<f4*>(xb + (size_t) K * 64 * (sizeof(f4) + sizeof(int)) + (size_t) k * MBAMD_P4_MATRIX_BYTES));
    const size_t col = (size_t) k * A.Ppad + (size_t) blk * 64 + lane;
    int e_col = 0;                                   // this column's cumulative exponent as the integration reads it
    if (A.cum != nullptr) {
        int32_t* dst = A.cum + col;
        const int total = (A.cumFresh ? 0 : *dst) + cum_e;
        if (A.cumFresh || cum_e != 0) *dst = total;
        if (t.cum == A.cum) e_col = total;
    }
    if (t.cum != nullptr && t.cum != A.cum) e_col = t.cum[col];
    f4* const xp = reinterpret_cast<f4*>(xb);                                  // [K][64] last results
    int* const xe = reinterpret_cast<int*>(xb + (size_t) K * 64 * sizeof(f4));   // [K][64] exponents
    xp[k * 64 + lane] = prev;
    xe[k * 64 + lane] = e_col;
    walk4_barrier();
    if (k != 0) return;
    // ---- wave 0: k_integrate_lnl_s4 for this block (count = 1), from LDS
    const int c = (int) (blk * 64 + lane);
    double wl = 0.0;
    if (c < t.P) {
        int emax = -2147483647;
        for (unsigned q = 0; q < K; ++q) { const int e = t.cum ? xe[q * 64 + lane] : 0; emax = e > emax ? e : emax; }
        unsigned mask = 0;
        if (t.child != nullptr && t.child_kind == CHILD_STATES) {
            const uint64_t* planes = reinterpret_cast<const uint64_t*>(t.child) + (size_t) blk * A.tstride;
            for (int i = 0; i < 4; ++i) mask |= (unsigned) (planes[i] >> lane & 1u) << i;
        }
        double total = 0.0;
        for (unsigned q = 0; q < K; ++q) {
            const f4 p = xp[q * 64 + lane];
            const int e = t.cum ? xe[q * 64 + lane] : 0;
            float f[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            if (t.child != nullptr) {
                const float* __restrict__ mT = t.matrix + q * 16;
                float v[4];
                if (t.child_kind == CHILD_STATES) {
                    for (int j = 0; j < 4; ++j) v[j] = (mask >> j & 1u) ? 1.0f : 0.0f;
                } else {
                    const f4 cq = reinterpret_cast<const f4*>(t.child)[(size_t) blk * A.pstride + (size_t) q * 64 + lane];
                    v[0] = cq.x; v[1] = cq.y; v[2] = cq.z; v[3] = cq.w;
                }
                for (int i = 0; i < 4; ++i)
                    f[i] = fmaf(mT[12 + i], v[3], fmaf(mT[8 + i], v[2], fmaf(mT[4 + i], v[1], mT[i] * v[0])));
            }
            const double cat = (double) (p.x * f[0]) * t.freqs[0] + (double) (p.y * f[1]) * t.freqs[1] + (double) (p.z * f[2]) * t.freqs[2] +
                               (double) (p.w * f[3]) * t.freqs[3];
            total += ldexp(cat * t.weights[q], e - emax);
        }
        const double lnl = log(total) + (double) emax * 0.69314718055994530942;
        t.site[c] = lnl;
        wl = lnl * t.pattern_weights[c];
    } else if (c < A.Ppad) {
        t.site[c] = 0.0;
    }
    mbd_wave_sum_store(wl, t.wsite + blk);
}

}  // namespace mbamd
#endif
