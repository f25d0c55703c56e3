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
#define MBAMD_WG2_TAIL    4      // ... of k_walkg2's programs
#define MBAMD_WG_STAGE    256    // bytes per wave in front of its slots (cumulative-exponent hand-over)
#define MBAMD_WG_STAGE_SPLIT 1024   // row split, per bin: [0, 256) the same hand-over, [256, 768) column maxima float [entry parity][half][32], [768, 776) the pair's progress counters
#define MBAMD_WG_PREV1    0x2000u   // (row split) child 1 / 2 is the result of the operation this bin executed last: wait for the partner's rows
#define MBAMD_WG_PREV2    0x4000u
#define MBAMD_WG_DRAIN    0x1000u   // (row split) the result is re-read from HBM by this bin in this phase: stores complete before the pair moves on

// Tile width: patterns per wave = 32 (v_mfma_f32_32x32x2_f32: two states per MFMA step, lane = 32 h + pattern).  A 16-pattern
// variant on v_mfma_f32_16x16x4_f32 (twice the waves, half the accumulators and epilogue each, 40- instead of 64-cycle dependent MFMA
// latency) was built in round 3, bit-for-bit as correct and SLOWER -- codon 100 x 5 000: 0.32 against 0.19 ms, protein 200 x 10 000:
// 0.34 against 0.23 ms (profiles/r03_exp_walkg_tile16_ablation.txt) -- because every wave fetches the whole A table of every child
// whatever its tile width: halving the tile doubles that traffic, and the operand fetch is what bounds the kernel.  Its branches
// were removed in round 5 (git history has them).
#define MBAMD_WG_TW 32
#define MBAMD_WG_KS (64 / MBAMD_WG_TW)    // states per row of a block = per MFMA step (2 or 4)
// Row split (round 5, k_walkg2; an instance created with MBAMD_WALKG_PAIR=1): beyond 48 states (the sense codons: two 32-row
// output tiles) a (tile, category, subtree bin) is a PAIR of waves -- wave h of the pair owns output tile h: half of A', half of
// the MFMA chain, half of the accumulators, half of the epilogue and of the stores; the child's rows (the B operand) come from
// the LDS slots the pair shares.  Twice the working waves for the same operand traffic (16-pattern tiles doubled it), and the
// registers that frees hold the operands of two whole entries.  The tables of such an instance hold tile 0's rows in front of
// tile 1's (`split` below).  MEASURED (profiles/r05_walkg_pair.txt): parity-green and no faster than k_walkg; opt-in.
// the transition-matrix kernels take "where the tables start" as one size_t (wgTab, floats into a matrix buffer; 0: no tables):
// its top bit says that the instance's tables have the row-split layout
#define MBAMD_WG_TAB_SPLIT ((size_t) 1 << 63)
__host__ __device__ inline bool wg_split_states(int S) { return S > 48; }
// state counts k_walkg2 is instantiated for (one output tile per wave: up to 32 states, or the row split); 40 states: k_walkg only
__host__ __device__ inline bool wg2_states(int S) { return S == 2 || S == 8 || S == 16 || S == 20 || S > 48; }
__host__ __device__ inline int wg_pairs(int S) { return (S + MBAMD_WG_KS - 1) / MBAMD_WG_KS; }         // T: MFMA steps (rows of a block)
__host__ __device__ inline int wg_tiles(int S) { return (S + MBAMD_WG_TW - 1) / MBAMD_WG_TW; }         // NT: output tiles of TW rows
__host__ __device__ inline int wg_vec(int S) { return S > 32 ? 4 : 2; }                   // V: floats per lane and memory instruction (blocks)
__host__ __device__ inline int wg_vec_a(int S) { return wg_vec(S); }                      // VA: the same for the tables
__host__ __device__ inline int wg_pairs_padded(int S) { return (wg_pairs(S) + wg_vec(S) - 1) / wg_vec(S) * wg_vec(S); }   // TP
__host__ __device__ inline int wg_rows(int S)                                             // NAP: 256-byte rows of a table
{
    const int n = wg_pairs_padded(S) * wg_tiles(S), va = wg_vec_a(S);
    return (n + va - 1) / va * va;
}
__host__ __device__ inline int wg_subtables(int S) { return S / MBAMD_WG_TW + 1; }        // gather tables: states 0..S in groups of TW (S = "missing")
__host__ __device__ inline unsigned wg_block_bytes(int S) { return (unsigned) wg_pairs_padded(S) * 256u; }   // one (tile, buffer, category) = one LDS slot
__host__ __device__ inline size_t wg_table_floats(int S) { return (size_t) (1 + wg_subtables(S)) * wg_rows(S) * 64; }   // per category
__host__ __device__ inline unsigned wg_stage_bytes(bool split) { return split ? MBAMD_WG_STAGE_SPLIT : MBAMD_WG_STAGE; }
// W = subtree bins of a workgroup (a bin is one wave, or a pair of waves with the row split)
__host__ __device__ inline size_t wg_lds_bytes(int W, int nslots, int S, bool split = false) { return (size_t) W * (wg_stage_bytes(split) + (size_t) nslots * wg_block_bytes(S)); }
__host__ __device__ inline int wg_waves_per_bin(bool split) { return split ? 2 : 1; }
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

// Tables of one (matrix, category): rows n = t * NT + it (< NAP), 64 columns, stored like blocks (wg_at with VA):
//   A'  (n, lane)         MFMA A operand of step t, output tile it
//   G_u (n, column)       tip gather: the lane of a pattern with state TW u + s reads ITS column of the rows n = r NT + it
//                         and has the factor registers (it, r) of a compact tip -- no MFMA
//       the column of state S ("missing") holds 1 for every existing from-state.
// The ROWS of A' are permuted so that the output tile lands in block layout (register r of lane group h = the state the
// next MFMA step t = ... wants there):
//   register r, half h = state 32 it + 2 r + h;   A'(n, 32 (j & 1) + row) with row = (r & 3) + 8 (r >> 2) + 4 h
//   G_u(n = r NT + it, 2 s + h)
//   row split (wg_split): n = it TP + t for A', n = it TP + r for G_u -- an output tile's rows are contiguous
// scatter P_k(i -> j) = v into the tables of category k (tab = first float of that category's tables)
__host__ __device__ inline void wg_table_put(float* tab, int S, int i, int j, float v, bool sp = false)
{
    const int NT = wg_tiles(S), NAP = wg_rows(S), VA = wg_vec_a(S);
    const int it = i >> 5, r = (i & 31) >> 1, h = i & 1;
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;                        // MFMA row that carries state i
    const int TP = wg_pairs_padded(S);
    tab[wg_at(VA, sp ? it * TP + (j >> 1) : (j >> 1) * NT + it, row + 32 * (j & 1))] = v;            // A'
    tab[(size_t) (1 + (j >> 5)) * NAP * 64 + wg_at(VA, sp ? it * TP + r : r * NT + it, 2 * (j & 31) + h)] = v;   // G_u
}
// the "missing" column (constant): from-state i
__host__ __device__ inline void wg_table_put_missing(float* tab, int S, int i, bool sp = false)
{
    const int NT = wg_tiles(S), NAP = wg_rows(S), VA = wg_vec_a(S);
    const int it = i >> 5, r = (i & 31) >> 1, h = i & 1;
    tab[(size_t) (1 + (S >> 5)) * NAP * 64 + wg_at(VA, sp ? it * wg_pairs_padded(S) + r : r * NT + it, 2 * (S & 31) + h)] = 1.0f;
}
// one thread per (matrix, category, state): the constant column of every matrix buffer, once per instance
__global__ void __launch_bounds__(256)
k_wg_init_tables(float* __restrict__ matrices, size_t matrixFloats, size_t tabOffFloats, int S, int K, int total, int split)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int i = g % S, mk = g / S;
    wg_table_put_missing(matrices + (size_t) (mk / K) * matrixFloats + tabOffFloats + (size_t) (mk % K) * wg_table_floats(S), S, i, split != 0);
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
    int pair;                    // 1: row split (k_walkg2): a subtree bin is a pair of waves, the stage area is MBAMD_WG_STAGE_SPLIT bytes
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
#include "mbamd_walkg2_kernel.h"      // k_walkg2: a whole entry's operands in flight; the row-split pair (round 5)
#include "mbamd_pathg_kernel.h"       // k_pathg: a move's root-ward path on two waves (sibling factors ahead of the chain)
#endif
