// mbamd_walkg.h -- the 20/61-state tree-walk kernel on the matrix cores (included by mbamd_kernels_mfma.h).
//
// Replaces CondLikeDown_Gen[_SSE] / CondLikeDown_NY98[_SSE] + CondLikeScaler_Gen / _NY98 + RemoveNodeScalers
// (reference src/likelihood.c:204-588, 1575-1900, 4939-5070, 5413-5545, 7981-8070) for whole operation lists --
// all eigen-system parts of a codon model together -- in ONE launch.
//
// Same decomposition as the 4-state walk (mbamd_walk4.h), with the matrix-vector products on v_mfma_f32_32x32x2_f32:
//     wave      = one 32-pattern tile, ONE rate category, interpreting a host-compiled program (Walk4Entry): a
//                 straight list of operations; no data ever crosses waves inside a phase, and the rescaling exponent is
//                 per (pattern, category) (exact powers of two, recombined exactly at the root: k_integrate_lnl_wg_wide)
//     workgroup = W waves (tree parallelism: subtrees packed into W bins, dependent phases separated by a barrier)
//     grid      = (tiles, categories), XCD-aware
// A child's conditional likelihoods live as the MFMA B operand wants them: [T = ceil(S/2)][64] floats per (tile, buffer,
// category) -- row t holds states 2t (lanes 0..31) and 2t+1 (lanes 32..63) of the 32 patterns, which is the tile-major
// layout [S][32] -- in HBM and, for results the same wave consumes again, in LDS slots private to the wave.
// The A operand (transition matrix) comes from a table whose ROWS ARE PERMUTED so that the 32x32 output tile lands in
// exactly that layout: register r of lane-half h holds state 32 it + 2 r + h.  A result is therefore T contiguous
// 256-byte stores to HBM and T ds_writes to its slot straight from the accumulators -- no LDS turn-around, no cross-wave
// exchange, no barrier per operation (the level kernels of mbamd_kernels_mfma.h spend three).  A compact tip needs no
// MFMA: its factor is a gather from a second table laid out for the same registers.
//
// Pipelining.  All vector-memory operations are compiler-visible (the waitcnt pass counts them), and the operand loads and
// result stores of an entry are THE SAME NUMBER on every path (NOP entries store zeros to a scratch buffer, the loads sit
// outside every branch): vmcnt retires in order and is shared by loads and stores, so only with a uniform sequence can the
// compiler wait for "the operands issued two jobs ago" without also waiting for the stores issued since -- one conditional
// instruction (even an exec-masked byte store) makes every such wait one instruction stricter, and that instruction is the
// oldest store of the previous entry.  Only the rows of a child that lives in HBM are loaded conditionally (rare in a full
// evaluation; unconditional dummies had the kernel bound by the 64 bytes per clock of the vector L1, measured).  A job = one child factor.  While job u computes, the
// operands of job u+2 are fetched into the third of three register sets: the A rows (or the tip's gather rows) and, for
// a child that lives in HBM (result of an earlier launch, of another wave, or evicted), its B rows.  Tip states are
// fetched one entry earlier still, entry descriptors three entries ahead through the scalar cache.
#ifndef MBAMD_WALKG_H_
#define MBAMD_WALKG_H_

namespace mbamd {

// Walk4Entry::ctl in this kernel: NOP / BARRIER / TIP1 / TIP2 / KEEP as in mbamd_walk4.h, and
#define MBAMD_WG_MEM1     0x04u  // child 1 / 2 is read from HBM (c1 / c2 = byte offset inside the tile's partials)
#define MBAMD_WG_MEM2     0x08u
#define MBAMD_WG_LIST(ctl) (((ctl) >> 10) & 3u)   // which of the merged lists the entry belongs to (cumulative buffer)
#define MBAMD_WG_MAXLISTS 4
#define MBAMD_WG_LEAD     2      // leading NOP entries (they fill the operand pipeline)
#define MBAMD_WG_TAIL     3      // trailing NOP entries (descriptor read-ahead)
#define MBAMD_WG_STAGE    256    // bytes per wave in front of its slots (cumulative-exponent hand-over)
#define MBAMD_WG_PREV1    0x2000u   // child 1 / 2 is the result of the entry this wave executed just before (and sits in a slot too)
#define MBAMD_WG_PREV2    0x4000u

// Tile width: patterns per wave = 32 (v_mfma_f32_32x32x2_f32: two states per MFMA step, lane = 32 h + pattern).  A 16-pattern
// variant on v_mfma_f32_16x16x4_f32 (twice the waves, half the accumulators and epilogue each, 40- instead of 64-cycle dependent MFMA
// latency) was built in round 3, bit-for-bit as correct and SLOWER -- codon 100 x 5 000: 0.32 against 0.19 ms, protein 200 x 10 000:
// 0.34 against 0.23 ms (profiles/r03_exp_walkg_tile16_ablation.txt) -- because every wave fetches the whole A table of every child
// whatever its tile width: halving the tile doubles that traffic, and the operand fetch is what bounds the kernel.  Its branches
// were removed in round 5 (git history has them).
#define MBAMD_WG_TW 32
#define MBAMD_WG_SCRATCH_ROWS 32  // exponent rows behind the scale buffers that entries without SCALE_WRITE store to, in rotation (nobody reads them)
#define MBAMD_WG_KS (64 / MBAMD_WG_TW)    // states per row of a block = per MFMA step (2 or 4)
// ---- round 6: the contraction on the 16-BIT matrix cores, in fp32 arithmetic (profiles/r06_bf16x3.txt) ----------------------------
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 cycles for 4 096 flops) and shares the SIMD's vector issue port: nothing
// else of the SIMD runs beside it.  v_mfma_f32_32x32x16_bf16 does 8 x the contraction depth in half the time on a pipe of its own.
// From MBAMD_WG_BF_MIN states on a factor is therefore formed as   sum over pieces  a = a1 + a2 + a3,  b = b1 + b2 + b3   (each piece
// the round-to-nearest bf16 of what the larger pieces left: 3 x 8 = 24 significant bits, the split is EXACT)  of the six products of
// order <= 4 -- a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1, in THAT order, smallest first, K-block (16 states) by K-block: every
// product of two bf16 is exact, what is dropped (a2 b3, a3 b2, a3 b3) is below 2^-25 of the result.  Measured against fp64 on
// transition-matrix x partials products (profiles/r06_bf16x3.txt): rms error 9.4e-8 at 61 states where the fp32 MFMA chain has
// 10.1e-8; the matrix core aligns its addends to the largest and truncates below one guard bit, which shows as a bias of
// -3e-8 per factor (the fp32 chain: -2.5e-8).  The A pieces are made once, by the matrix kernel
// (wg_table_put); the B pieces by the consumer, from the fp32 rows in its registers (11 VALU instructions per two values, beside
// the MFMAs).  Blocks in HBM and in LDS, tips, exponents: unchanged -- an fp32 [T][64] block is both kernels' operand.
#if !defined(MBAMD_WG_BF_MIN)
#define MBAMD_WG_BF_MIN 40       // state counts from here on contract on the 16-bit matrix cores
#endif
__host__ __device__ inline bool wg_bf16(int S) { return S >= MBAMD_WG_BF_MIN; }
__host__ __device__ inline int wg_kblocks(int S) { return (S + 15) / 16; }                // NKB: contraction blocks of 16 states
// K-block kb holds the block rows t = 8 kb .. 8 kb + 7 (the lane's registers): element j of lane half h = state 16 kb + 2 j + h
__host__ __device__ inline int wg_pairs(int S) { return (S + MBAMD_WG_KS - 1) / MBAMD_WG_KS; }         // T: rows of a block (fp32 mode: MFMA steps)
__host__ __device__ inline int wg_tiles(int S) { return (S + MBAMD_WG_TW - 1) / MBAMD_WG_TW; }         // NT: output tiles of TW rows
__host__ __device__ inline int wg_vec(int S) { return S > 32 ? 4 : 2; }                   // V: floats per lane and memory instruction (blocks)
__host__ __device__ inline int wg_vec_a(int S) { return wg_bf16(S) ? 4 : wg_vec(S); }     // VA: the same for the tables
__host__ __device__ inline int wg_pairs_padded(int S) { return (wg_pairs(S) + wg_vec(S) - 1) / wg_vec(S) * wg_vec(S); }   // TP
// 256-byte rows of the A' area of a table.  fp32 mode: one row per (MFMA step, output tile).  bf16 mode: 4 rows = one 16-byte
// operand per lane for (K-block, output tile, piece): 3 NT NKB operands
__host__ __device__ inline int wg_rows(int S)                                             // NAP
{
    if (wg_bf16(S)) return 4 * 3 * wg_tiles(S) * wg_kblocks(S);
    const int n = wg_pairs_padded(S) * wg_tiles(S), va = wg_vec_a(S);
    return (n + va - 1) / va * va;
}
// 256-byte rows of ONE tip-gather table: register (it, r), r < min(16, T), of a compact tip's factor = row r NT + it
__host__ __device__ inline int wg_gather_rows(int S)                                      // NGR
{
    if (!wg_bf16(S)) return wg_rows(S);              // (fp32 mode: as many as A', the layout of rounds 2-5)
    const int t = wg_pairs(S) < 16 ? wg_pairs(S) : 16, n = t * wg_tiles(S);
    return (n + 3) / 4 * 4;
}
__host__ __device__ inline int wg_subtables(int S) { return S / MBAMD_WG_TW + 1; }        // gather tables: states 0..S in groups of TW (S = "missing")
__host__ __device__ inline unsigned wg_block_bytes(int S) { return (unsigned) wg_pairs_padded(S) * 256u; }   // one (tile, buffer, category) = one LDS slot
__host__ __device__ inline size_t wg_table_floats(int S) { return ((size_t) wg_rows(S) + (size_t) wg_subtables(S) * wg_gather_rows(S)) * 64; }   // per category
// W = subtree bins of a workgroup = its working waves
__host__ __device__ inline size_t wg_lds_bytes(int W, int nslots, int S) { return (size_t) W * (MBAMD_WG_STAGE + (size_t) nslots * wg_block_bytes(S)); }
// A block holds [TP rows][64 lanes]: row t, lane TW h + p = state KS t + h of pattern p; V consecutive rows are interleaved
// per lane so that one dword / dwordx2 / dwordx4 per lane moves V rows (256 B - 1 KiB contiguous per wave instruction).
// float offset of (row r, lane / column c) inside a block or table:
__host__ __device__ inline unsigned wg_at(int V, int r, int c) { return (unsigned) ((r / V) * 64 * V + c * V + r % V); }
// the same with V = 1 << sh: no integer division in a kernel's inner loop
__host__ __device__ inline unsigned wg_elem_sh(int sh, int i, int p)
{
    const int r = i / MBAMD_WG_KS, lane = (i % MBAMD_WG_KS) * MBAMD_WG_TW + p;
    return (unsigned) (((r >> sh) << (6 + sh)) + (lane << sh) + (r & ((1 << sh) - 1)));
}
__host__ __device__ inline int wg_vec_shift(int S) { const int v = wg_vec(S); return v == 4 ? 2 : (v == 2 ? 1 : 0); }
__host__ __device__ inline unsigned wg_elem(int S, int i, int p) { return wg_at(wg_vec(S), i / MBAMD_WG_KS, (i % MBAMD_WG_KS) * MBAMD_WG_TW + p); }   // state i, pattern p

// Tables of one (matrix, category), stored like blocks (wg_at with VA): the A' area (NAP rows), then wg_subtables gather tables G_u
// of NGR rows each.
//   fp32 mode:  A'  (n = t NT + it, lane)   MFMA A operand of step t, output tile it
//   bf16 mode:  A'  operand o = (kb NT + it) 3 + piece: 16 bytes per lane = the piece's eight bf16 P(i -> 16 kb + 2 j + h'), j = 0..7,
//                   of the lane (h', m) whose MFMA row m carries from-state i
//   G_u (n = r NT + it, column)   tip gather: the lane of a pattern with state TW u + s reads ITS column of these rows and has the
//                         factor registers (it, r) of a compact tip -- no MFMA (fp32 in both modes)
//       the column of state S ("missing") holds 1 for every existing from-state.
// The ROWS of A' are permuted so that the output tile lands in block layout (register r of lane group h = the state the
// next contraction wants there):
//   register r, half h = state 32 it + 2 r + h;   MFMA row = (r & 3) + 8 (r >> 2) + 4 h
//   G_u(n = r NT + it, 2 s + h)
__host__ __device__ inline uint16_t wg_bf16_rne(float v)         // the bf16 nearest to v (ties to even): what v_cvt_pk_bf16_f32 gives a finite v
{
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    return (uint16_t) ((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float wg_bf16_float(uint16_t p) { return __builtin_bit_cast(float, (uint32_t) p << 16); }
// scatter P_k(i -> j) = v into the tables of category k (tab = first float of that category's tables)
__host__ __device__ inline void wg_table_put(float* tab, int S, int i, int j, float v)
{
    const int NT = wg_tiles(S), NAP = wg_rows(S), NGR = wg_gather_rows(S), VA = wg_vec_a(S);
    const int it = i >> 5, r = (i & 31) >> 1, h = i & 1;
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;                        // MFMA row that carries state i
    if (wg_bf16(S)) {
        uint16_t* t16 = reinterpret_cast<uint16_t*>(tab);
        const int kb = j >> 4, jj = (j & 15) >> 1, lane = 32 * (j & 1) + row;
        const size_t at = ((size_t) (kb * NT + it) * 3 * 64 + lane) * 8 + jj;        // (operand, lane, element) -> bf16 index; + 512 per piece
        const uint16_t p1 = wg_bf16_rne(v);
        const float r1 = v - wg_bf16_float(p1);
        const uint16_t p2 = wg_bf16_rne(r1);
        const uint16_t p3 = wg_bf16_rne(r1 - wg_bf16_float(p2));
        t16[at] = p1; t16[at + 512] = p2; t16[at + 1024] = p3;
    } else {
        tab[wg_at(VA, (j >> 1) * NT + it, row + 32 * (j & 1))] = v;            // A'
    }
    tab[((size_t) NAP + (size_t) (j >> 5) * NGR) * 64 + wg_at(VA, r * NT + it, 2 * (j & 31) + h)] = v;   // G_u
}
// the "missing" column (constant): from-state i
__host__ __device__ inline void wg_table_put_missing(float* tab, int S, int i)
{
    const int NT = wg_tiles(S), NAP = wg_rows(S), NGR = wg_gather_rows(S), VA = wg_vec_a(S);
    const int it = i >> 5, r = (i & 31) >> 1, h = i & 1;
    tab[((size_t) NAP + (size_t) (S >> 5) * NGR) * 64 + wg_at(VA, r * NT + it, 2 * (S & 31) + h)] = 1.0f;
}
// one thread per (matrix, category, state): the constant column of every matrix buffer, once per instance
__global__ void __launch_bounds__(256)
k_wg_init_tables(float* __restrict__ matrices, size_t matrixFloats, size_t tabOffFloats, int S, int K, int total)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int i = g % S, mk = g / S;
    wg_table_put_missing(matrices + (size_t) (mk / K) * matrixFloats + tabOffFloats + (size_t) (mk % K) * wg_table_floats(S), S, i);
}

struct WalkGArgs {
    const Walk4Entry* prog;      // [W][entries]
    int entries;                 // per wave: MBAMD_WG_LEAD NOPs + program, padded to a multiple of 3, + MBAMD_WG_TAIL NOPs
    int nslots;
    float* partials;             // arena float [tile][buffer][K] blocks of wg_block_bytes (wg_at layout)
    unsigned long tileBytes;     // bytes between tiles
    const uint8_t* tips;         // arena uint8 [tile][buffer][TW]   state codes, S = missing
    unsigned tipTileBytes;
    int8_t* exps;                // arena int8 [tile / 2][scale buffer][K][64] (the 4-state path's format)
    unsigned estride;            // bytes between 64-pattern blocks
    const float* matrices;       // matrix buffers; entries hold the byte offset of a buffer
    unsigned tabOff, tabBytes;   // byte offset of the table area inside a matrix buffer / bytes per category
    int32_t* cum[MBAMD_WG_MAXLISTS];   // wide cumulative buffers int32 [K][Ppad] per merged list, or nullptr
    int cumFresh;                // bit q: list q's cumulative buffer holds nothing yet (store, do not add)
    int K, Ppad, ntiles, S, SP;
    int lists;                   // > 1: mutually independent lists run as separate workgroups (programs [list][W][entries]); else 1
    int spread;                  // 1: the workgroup is launched with 2 W waves and only the even ones work (see k_walkg)
    long long* reserved;         // (was: clock stamps of timing experiments)
};
__host__ __device__ inline unsigned walkg_grid(int ntiles, int KL) { return 8u * (unsigned) KL * (unsigned) ((ntiles + 7) / 8); }   // KL = categories x lists

// a short program (a partial update) in the kernel arguments instead of a device buffer, as in mbamd_walk4.h
struct WalkGArgsInline {
    WalkGArgs a;
    Walk4Entry inl[MBAMD_W4_INLINE];
};
__device__ __forceinline__ const WalkGArgs& wg_args(const WalkGArgs& a) { return a; }
__device__ __forceinline__ const WalkGArgs& wg_args(const WalkGArgsInline& a) { return a.a; }
__device__ __forceinline__ const Walk4Entry* wg_program(const WalkGArgs& a) { return a.prog; }

}  // namespace mbamd
#include <mbamd_dev_walkg_kernel.h>   // the kernels' device primitives (csrc/device/: MFMA, lane swap, waits; tests/hostemu/: the same on fibers)
#include "mbamd_walkg_kernel.h"       // k_walkg
#include "mbamd_pathg_kernel.h"       // k_pathg: a move's root-ward path on two waves (sibling factors ahead of the chain)
#endif
