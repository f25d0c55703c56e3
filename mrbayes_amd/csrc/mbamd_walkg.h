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

// Tile width: patterns per wave.  32 = v_mfma_f32_32x32x2_f32 (two states per MFMA step, lane = 32 h + pattern), 16 =
// v_mfma_f32_16x16x4_f32 (four states per step, lane = 16 g + pattern): twice the waves for the same alignment, each with half
// the accumulators, half the epilogue and a 40- instead of 64-cycle dependent MFMA latency.  One or the other per build.
// MEASURED (round 3, profiles/r03_exp_walkg_tile16_ablation.txt): 16 is bit-for-bit as correct (every GPU parity test) and
// SLOWER -- codon 100 x 5 000: 0.32 against 0.19 ms, protein 200 x 10 000: 0.34 (0.28 with one wave per workgroup) against
// 0.23 ms -- because every wave fetches the whole A table of every child whatever its tile width: halving the tile doubles
// that traffic, and the operand fetch is what bounds the kernel (with the fetch ablated 16 beats 32: 0.104 against 0.118 ms).
// The product is built with 32; -DMBAMD_WG_TW=16 keeps the other one buildable (tools/build_variants.py).
#if !defined(MBAMD_WG_TW)
#define MBAMD_WG_TW 32
#endif
#define MBAMD_WG_KS (64 / MBAMD_WG_TW)    // states per row of a block = per MFMA step (2 or 4)
__host__ __device__ inline int wg_pairs(int S) { return (S + MBAMD_WG_KS - 1) / MBAMD_WG_KS; }         // T: MFMA steps (rows of a block)
__host__ __device__ inline int wg_tiles(int S) { return (S + MBAMD_WG_TW - 1) / MBAMD_WG_TW; }         // NT: output tiles of TW rows
#if MBAMD_WG_TW == 32
__host__ __device__ inline int wg_vec(int S) { return S > 32 ? 4 : 2; }                   // V: floats per lane and memory instruction (blocks)
__host__ __device__ inline int wg_vec_a(int S) { return wg_vec(S); }                      // VA: the same for the tables
#else
// blocks: 61 states 16 rows (V 4), 20 states 5 rows (V 1: no padding bytes -- the time follows the bytes), 16 states 4 rows
__host__ __device__ inline int wg_vec(int S) { const int T = wg_pairs(S); return (S > 32 || T % 4 == 0) ? 4 : (T % 2 == 0 ? 2 : 1); }
// tables: rows n = t NT + it; 61 states 64 rows (VA 4), 20 states 10 rows (VA 2), 16 states 4 rows (VA 4)
__host__ __device__ inline int wg_vec_a(int S) { const int n = wg_pairs(S) * wg_tiles(S); return (S > 32 || n % 4 == 0) ? 4 : (n % 2 == 0 ? 2 : 1); }
#endif
__host__ __device__ inline int wg_pairs_padded(int S) { return (wg_pairs(S) + wg_vec(S) - 1) / wg_vec(S) * wg_vec(S); }   // TP
__host__ __device__ inline int wg_rows(int S)                                             // NAP: 256-byte rows of a table
{
    const int n = wg_pairs_padded(S) * wg_tiles(S), va = wg_vec_a(S);
    return (n + va - 1) / va * va;
}
__host__ __device__ inline int wg_subtables(int S) { return S / MBAMD_WG_TW + 1; }        // gather tables: states 0..S in groups of TW (S = "missing")
__host__ __device__ inline unsigned wg_block_bytes(int S) { return (unsigned) wg_pairs_padded(S) * 256u; }   // one (tile, buffer, category) = one LDS slot
__host__ __device__ inline size_t wg_table_floats(int S) { return (size_t) (1 + wg_subtables(S)) * wg_rows(S) * 64; }   // per category
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

// Tables of one (matrix, category): rows n = t * NT + it (< NAP), 64 columns, stored like blocks (wg_at with VA):
//   A'  (n, lane)         MFMA A operand of step t, output tile it
//   G_u (n, column)       tip gather: the lane of a pattern with state TW u + s reads ITS column of the rows n = r NT + it
//                         and has the factor registers (it, r) of a compact tip -- no MFMA
//       the column of state S ("missing") holds 1 for every existing from-state.
// The ROWS of A' are permuted so that the output tile lands in block layout (register r of lane group h = the state the
// next MFMA step t = ... wants there):
//   TW 32:  register r, half h           = state 32 it + 2 r + h;   A'(n, 32 (j & 1) + row) with row = (r & 3) + 8 (r >> 2) + 4 h
//           G_u(n = r NT + it, 2 s + h)
//   TW 16:  register r (0..3), group g   = state 16 it + 4 r + g;   A'(n, 16 (j & 3) + row) with row = 4 g + r
//           G_u(n = r NT + it, 4 s + g)
// scatter P_k(i -> j) = v into the tables of category k (tab = first float of that category's tables)
__host__ __device__ inline void wg_table_put(float* tab, int S, int i, int j, float v)
{
    const int NT = wg_tiles(S), NAP = wg_rows(S), VA = wg_vec_a(S);
#if MBAMD_WG_TW == 32
    const int it = i >> 5, r = (i & 31) >> 1, h = i & 1;
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;                        // MFMA row that carries state i
    tab[wg_at(VA, (j >> 1) * NT + it, row + 32 * (j & 1))] = v;            // A'
    tab[(size_t) (1 + (j >> 5)) * NAP * 64 + wg_at(VA, r * NT + it, 2 * (j & 31) + h)] = v;   // G_u
#else
    const int it = i >> 4, r = (i & 15) >> 2, g = i & 3;
    const int row = 4 * g + r;
    tab[wg_at(VA, (j >> 2) * NT + it, row + 16 * (j & 3))] = v;            // A'
    tab[(size_t) (1 + (j >> 4)) * NAP * 64 + wg_at(VA, r * NT + it, 4 * (j & 15) + g)] = v;   // G_u
#endif
}
// the "missing" column (constant): from-state i
__host__ __device__ inline void wg_table_put_missing(float* tab, int S, int i)
{
    const int NT = wg_tiles(S), NAP = wg_rows(S), VA = wg_vec_a(S);
#if MBAMD_WG_TW == 32
    const int it = i >> 5, r = (i & 31) >> 1, h = i & 1;
    tab[(size_t) (1 + (S >> 5)) * NAP * 64 + wg_at(VA, r * NT + it, 2 * (S & 31) + h)] = 1.0f;
#else
    const int it = i >> 4, r = (i & 15) >> 2, g = i & 3;
    tab[(size_t) (1 + (S >> 4)) * NAP * 64 + wg_at(VA, r * NT + it, 4 * (S & 15) + g)] = 1.0f;
#endif
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
    long long* trace;            // MBAMD_WALK_TRACE: clock stamps [entry][8 waves][3] of one workgroup (timing experiments), or nullptr
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
#if defined(MBAMD_HOST_EMU)
__device__ __forceinline__ const Walk4Entry* wg_program(const WalkGArgsInline& a) { return a.inl; }
#else
__device__ __forceinline__ const Walk4Entry* wg_program(const WalkGArgsInline&)
{
    return reinterpret_cast<const Walk4Entry*>((uintptr_t) __builtin_amdgcn_kernarg_segment_ptr() + offsetof(WalkGArgsInline, inl));
}
#endif

#if defined(MBAMD_HOST_EMU)
#include "mbamd_walkg_emu.h"     // tests/hostemu/ (test build only): a plain-loop twin of k_walkg for CPU CI of the host logic
#else

// V floats of one lane: a clang vector, or -- V = 1 -- a plain float (vectors of one element are not loadable types)
template <int V> struct WgVecT {
    typedef float type __attribute__((ext_vector_type(V)));
    static __device__ __forceinline__ float get(const type& v, int i) { return v[i]; }
    static __device__ __forceinline__ void set(type& v, int i, float x) { v[i] = x; }
    static __device__ __forceinline__ type splat(float x) { return (type) (x); }
};
template <> struct WgVecT<1> {
    typedef float type;
    static __device__ __forceinline__ float get(const type& v, int) { return v; }
    static __device__ __forceinline__ void set(type& v, int, float x) { v = x; }
    static __device__ __forceinline__ type splat(float x) { return x; }
};

template <int SC> struct WgShape {
    static constexpr int TW = MBAMD_WG_TW, KS = MBAMD_WG_KS;
    static constexpr int T = (SC + KS - 1) / KS, NT = (SC + TW - 1) / TW;
#if MBAMD_WG_TW == 32
    static constexpr int V = SC > 32 ? 4 : 2, VA = V;
    static constexpr int ACC = 16;                   // accumulator registers per output tile (32 x 32 / 64 lanes)
#else
    static constexpr int V = (SC > 32 || T % 4 == 0) ? 4 : (T % 2 == 0 ? 2 : 1);
    static constexpr int VA = (SC > 32 || (T * NT) % 4 == 0) ? 4 : ((T * NT) % 2 == 0 ? 2 : 1);
    static constexpr int ACC = 4;                    // 16 x 16 / 64 lanes
#endif
    static constexpr int TP = (T + V - 1) / V * V, NAP = (TP * NT + VA - 1) / VA * VA;
    typedef WgVecT<V> Vb;                            // block rows (B operand, results)
    typedef WgVecT<VA> Va;                           // table rows (A operand, tip gathers)
    typedef typename Vb::type vec;
    typedef typename Va::type vecA;
    typedef float acc __attribute__((ext_vector_type(ACC)));
};
// one CHUNK of a job's operands: a job (one child factor) is CH chunks of TP / CH MFMA steps
template <int SC, int CH> struct WgOperands {
    typename WgShape<SC>::vecA a[WgShape<SC>::NAP / WgShape<SC>::VA / CH];  // A rows of the chunk (or the tip's gather rows), VA rows per register group
    typename WgShape<SC>::vec b[WgShape<SC>::TP / WgShape<SC>::V / CH];     // B rows of a child read from HBM
};
struct WgDesc {
    Walk4Entry e;
    unsigned s1, s2;           // tip states of this lane's pattern (children 1, 2)
};
template <int I, class O> __device__ __forceinline__ O& wg_pick(O& a, O& b, O& c)
{
    if constexpr (I == 0) return a;
    else if constexpr (I == 1) return b;
    else return c;
}
template <int I> struct WgInt { static constexpr int value = I; };

// blockDim.x = 64 * W (W <= WMAX); grid = walkg_grid(ntiles, K); dynamic LDS = wg_lds_bytes(W, nslots, SC).
// The operand pipeline works in CHUNKS: a job (one child factor, T MFMA steps per row tile) is CH chunks, an entry 2 CH,
// and the operands of a chunk are fetched DEPTH chunks ahead into one of DEPTH + 1 rotating register sets.
//   20 states: CH 1, DEPTH 2 -- a job is 10 MFMAs = 640 cycles, less than a memory round trip; three sets of 15 registers;
//   61 states: CH 2, DEPTH 1 -- a chunk is 31 MFMAs = 2000 cycles; two sets of 48 registers fit beside the 64 accumulators
//              (whole jobs did not: the allocator shuttled LOADED operands through AccVGPRs, a vmcnt(0) per job).
template <int SC, int WMAX, int CH, int DEPTH, class ARGS = WalkGArgs>
__global__ void __launch_bounds__(64 * WMAX)
k_walkg(ARGS AA)
{
    const WalkGArgs& A = wg_args(AA);
    typedef WgShape<SC> Sh;
    typedef typename Sh::vec vec;
    typedef typename Sh::vecA vecA;
    typedef typename Sh::Vb Vb;
    typedef typename Sh::Va Va;
    typedef typename Sh::acc acc_t;
    typedef WgOperands<SC, CH> Ops;
    constexpr int TW = Sh::TW, KS = Sh::KS, ACC = Sh::ACC;
    constexpr int T = Sh::T, NT = Sh::NT, V = Sh::V, VA = Sh::VA, TP = Sh::TP, NAP = Sh::NAP, NAV = NAP / VA, TV = TP / V;
    constexpr int TPC = TP / CH, NAVC = NAV / CH, TVC = TV / CH;      // per chunk: MFMA steps, A register groups, B register groups
    constexpr int NQ = 2 * CH, NS = DEPTH + 1;                        // chunks per entry, register sets
    static_assert(TP % CH == 0 && TPC % V == 0 && NAV % CH == 0 && (TPC * NT) % VA == 0 && NS <= 3 && DEPTH <= NQ && TPC <= 16, "chunk geometry");
    static_assert((ACC < T ? ACC : T) * NT <= NAVC * VA, "a compact tip's gather rows must lie in the first chunk");
    constexpr unsigned SLOTB = TP * 256u;
    const unsigned lane = threadIdx.x & 63, half = lane / TW, col = lane % TW;      // half: which of the KS states of a row
    int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    int W = (int) (blockDim.x >> 6);
    // Two-wave workgroups land on one SIMD pair of the CU and leave the other pair's matrix cores idle (measured: 73 against
    // 140 TFLOP/s of dense MFMA with waves {0, 1} against {0, 2} of a four-wave workgroup): such workgroups are launched
    // with twice the waves, and the odd ones leave at once.
    if (A.spread) {
        if (wave & 1) return;
        wave >>= 1;
        W >>= 1;
    }
    extern __shared__ float lds_walkg[];
    const unsigned K = (unsigned) A.K, KL = K * (unsigned) A.lists;
    const unsigned xcd = blockIdx.x & 7u, pos = blockIdx.x >> 3;
    const unsigned tile = (pos / KL) * 8u + xcd, k = (pos % KL) % K, list = (pos % KL) / K;
    if (tile >= (unsigned) A.ntiles) return;
    char* const mine = reinterpret_cast<char*>(lds_walkg) + (size_t) wave * (MBAMD_WG_STAGE + (size_t) A.nslots * SLOTB);
    vec* const slots = reinterpret_cast<vec*>(mine + MBAMD_WG_STAGE) + lane;          // this lane's V rows of row group 0, slot 0
    // wave-uniform bases; the entries hold byte offsets from them
    char* const P0 = reinterpret_cast<char*>(A.partials) + (size_t) tile * A.tileBytes + (size_t) k * SLOTB;
    const uint8_t* const T0 = A.tips + (size_t) tile * A.tipTileBytes;
    int8_t* const E0 = A.exps + (size_t) ((tile * TW) >> 6) * A.estride + (size_t) k * 64 + ((tile * TW) & 63u);
    const char* const Mk = reinterpret_cast<const char*>(A.matrices) + A.tabOff + (size_t) k * A.tabBytes;

    const Walk4Entry* prog = wg_program(AA) + ((size_t) list * W + wave) * A.entries;
    const int n = A.entries - MBAMD_WG_TAIL;
#if defined(MBAMD_WGX_STAGGER)
    // Waves of different tiles run the SAME program: left alone they march in lockstep -- all waves of a SIMD in their MFMA
    // chains at once, then all in their epilogues with the matrix pipe idle.  A start offset per workgroup persists.
    for (unsigned d = ((blockIdx.x >> 3) * 2654435761u >> 16) % MBAMD_WGX_STAGGER; d > 0; --d) __builtin_amdgcn_s_sleep(8);
#endif
    WgDesc DA, DB, DC;
    DA.e = walk4_load_entry(prog); DB.e = walk4_load_entry(prog + 1); DC.e = walk4_load_entry(prog + 2);
    DA.s1 = DA.s2 = DB.s1 = DB.s2 = DC.s1 = DC.s2 = 0;
    Ops X, Y, Z;
#pragma unroll
    for (int i = 0; i < NAVC; ++i) X.a[i] = Y.a[i] = Z.a[i] = Va::splat(0.0f);
#pragma unroll
    for (int i = 0; i < TVC; ++i) X.b[i] = Y.b[i] = Z.b[i] = Vb::splat(0.0f);
    int er = 0;                                      // stored exponent of the entry about to run (SCALE_READ)
    int cum_e[MBAMD_WG_MAXLISTS] = {0, 0, 0, 0};

    // operands of chunk q (child q / CH, part q % CH) of entry d -> register set o: NAVC loads outside any branch
    // (+ TVC for a child that lives in HBM)
    auto fetch = [&](const WgDesc& d, int q, Ops& o) {
        const int ch = q / CH, h = q % CH;
        const unsigned ctl = d.e.ctl;
        const bool tip = ctl & (ch ? MBAMD_W4_TIP2 : MBAMD_W4_TIP1), mem = ctl & (ch ? MBAMD_WG_MEM2 : MBAMD_WG_MEM1);
        const unsigned coff = ch ? d.e.c2 : d.e.c1, moff = ch ? d.e.m2 : d.e.m1;
        const unsigned s = ch ? d.s2 : d.s1;
        // a: A' (column = lane) or the tip's gather table (column = KS * (state % TW) + half of sub-table state / TW)
        const unsigned aoff = tip ? (1u + s / TW) * (unsigned) (NAP * 256) + ((s % TW) * KS + half) * (unsigned) (VA * 4) : lane * (unsigned) (VA * 4);
        const MBAMD_AS_GLOBAL vecA* pa = reinterpret_cast<const MBAMD_AS_GLOBAL vecA*>((uintptr_t) (Mk + moff) + aoff) + h * NAVC * 64;
#if !defined(MBAMD_WGX_NOFETCH)
#pragma unroll
        for (int i = 0; i < NAVC; ++i) o.a[i] = pa[i * 64];
#else
        (void) pa;
#endif
        if (mem) {
            const MBAMD_AS_GLOBAL vec* pb = reinterpret_cast<const MBAMD_AS_GLOBAL vec*>((uintptr_t) (P0 + coff)) + lane + h * TVC * 64;
#pragma unroll
            for (int i = 0; i < TVC; ++i) o.b[i] = pb[i * 64];
        }
    };
    // chunk q of an entry: MFMA steps [h TPC, (h + 1) TPC) of one child factor; register (it, r) of f = the state of this lane's
    // pattern that block row ACC it + r holds in this lane.  No vector-memory operation in here.  Two halves: the B rows into
    // registers (the only LDS wait), then the MFMA chain (or the tip's gather rows) -- the caller may issue scalar loads in between.
    auto operandB = [&](const Walk4Entry& de, int q, const Ops& o, vec (&b)[TVC]) {
        const int ch = q / CH, h = q % CH;
        if (de.ctl & (ch ? MBAMD_WG_MEM2 : MBAMD_WG_MEM1)) {
#pragma unroll
            for (int i = 0; i < TVC; ++i) b[i] = o.b[i];
        } else {
            const vec* sl = reinterpret_cast<const vec*>(reinterpret_cast<const char*>(slots) + (ch ? de.c2 : de.c1)) + h * TVC * 64;
#if !defined(MBAMD_WGX_NOLDS)
#pragma unroll
            for (int i = 0; i < TVC; ++i) b[i] = sl[i * 64];
#else
            (void) sl;
#pragma unroll
            for (int i = 0; i < TVC; ++i) b[i] = Vb::splat(1.0f);
#endif
        }
    };
    auto compute = [&](bool tip, int q, const Ops& o, const vec (&b)[TVC], acc_t (&f)[NT]) {
        const int h = q % CH;
        if (tip) {
            if (h == 0) {                            // all registers of every output tile come from the first chunk's rows
#pragma unroll
                for (int it = 0; it < NT; ++it)
#pragma unroll
                    for (int r = 0; r < ACC; ++r) f[it][r] = (ACC * it + r < T) ? Va::get(o.a[(r * NT + it) / VA], (r * NT + it) % VA) : 0.0f;
            }
            return;
        }
        if (h == 0) {
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < ACC; ++r) f[it][r] = 0.0f;
        }
#if !defined(MBAMD_WGX_NOMFMA)
#pragma unroll
        for (int tc = 0; tc < TPC; ++tc)
            if (h * TPC + tc < T) {
#pragma unroll
                for (int it = 0; it < NT; ++it) {
                    const float av = Va::get(o.a[(tc * NT + it) / VA], (tc * NT + it) % VA), bv = Vb::get(b[tc / V], tc % V);
#if MBAMD_WG_TW == 32
                    f[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, f[it], 0, 0, 0);
#else
                    f[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, f[it], 0, 0, 0);
#endif
                }
            }
#else
#pragma unroll
        for (int tc = 0; tc < TPC; ++tc) f[0][tc % ACC] += Va::get(o.a[(tc * NT) / VA], (tc * NT) % VA) * Vb::get(b[tc / V], tc % V);
#endif
    };

    // One iteration = one entry `cur`: its NQ chunks run on the register sets (S0, S1, S2, S0, ...) while the chunks DEPTH
    // further on -- of cur, then of entry `n1` -- are fetched; the tip states of entry `n2` are fetched, and cur's descriptor
    // is replaced by entry j + 3.  Vector-memory sequence, identical on every path:
    //     NQ x NAVC operand loads | TV + 1 stores      (+ the rare conditional loads)
#if defined(MBAMD_WG_TRACE)      // timing experiments only (MBAMD_BUILD_DEFINES=MBAMD_WG_TRACE): the stamps cost scalar-memory waits
    const bool tracing = A.trace != nullptr && blockIdx.x == 8 && lane == 0;
#else
    constexpr bool tracing = false;
#endif
    auto step = [&](WgDesc& cur, const WgDesc& n1, WgDesc& n2, Ops& S0, Ops& S1, Ops& S2, int j) {
        const unsigned ctl = cur.e.ctl;
        if (tracing) A.trace[((size_t) j * 8 + wave) * 3 + 0] = (long long) __builtin_amdgcn_s_memtime();
        if (ctl & MBAMD_W4_BARRIER) {
            // values other waves produced in the previous phase are read from here on: drain this wave's stores, meet
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        const bool run = !(ctl & MBAMD_W4_NOP);
        const unsigned mode = (ctl >> 8) & 3u;
        // (issued before the operand fetches: what the next entry needs first must not queue behind them)
        // (three tiny loads, unconditional: every conditional vector-memory instruction makes the compiler's vmcnt waits one
        //  instruction stricter -- and the instruction they then wait for is the oldest STORE of the previous entry)
#if !defined(MBAMD_WGX_NOTINY)
        const int er_next = as_global(E0 + n1.e.eread)[col];
        n2.s1 = as_global(T0 + ((n2.e.ctl & MBAMD_W4_TIP1) ? n2.e.c1 : 0u))[col];
        n2.s2 = as_global(T0 + ((n2.e.ctl & MBAMD_W4_TIP2) ? n2.e.c2 : 0u))[col];
#else
        const int er_next = 0;
#endif
        acc_t f1[NT], f2[NT];
        const Walk4Entry ce = cur.e;
        auto chunk = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q < NQ) {
                constexpr int qf = q + DEPTH;        // the chunk fetched now
                if constexpr (qf < NQ) {
                    WgDesc t;
                    t.e = ce; t.s1 = cur.s1; t.s2 = cur.s2;
                    fetch(t, qf, wg_pick<qf % NS>(S0, S1, S2));
                } else {
                    fetch(n1, qf - NQ, wg_pick<qf % NS>(S0, S1, S2));
                }
                const bool tip = ce.ctl & (q / CH ? MBAMD_W4_TIP2 : MBAMD_W4_TIP1);
                // The descriptor of entry j + 3 replaces this entry's: a scalar load, and scalar loads share lgkmcnt with LDS
                // and return out of order -- any LDS wait while it is in flight is a wait for IT.  It is issued when the
                // last LDS read of the entry has landed (the epilogue has none): the last chunk's MFMA chain covers it.
                if (run && !tip) {
                    vec b[TVC];
                    operandB(ce, q, wg_pick<q % NS>(S0, S1, S2), b);
                    if constexpr (q == NQ - 1) {
#pragma unroll
                        for (int i = 0; i < TVC; ++i) asm volatile("" :: "v"(b[i]) : "memory");
                        cur.e = walk4_load_entry(prog + j + 3);
                    }
                    compute(false, q, wg_pick<q % NS>(S0, S1, S2), b, q < CH ? f1 : f2);
                } else {
                    if constexpr (q == NQ - 1) cur.e = walk4_load_entry(prog + j + 3);
                    vec b[TVC];
                    if (run) compute(true, q, wg_pick<q % NS>(S0, S1, S2), b, q < CH ? f1 : f2);
                }
            }
        };
        chunk(WgInt<0>{}); chunk(WgInt<1>{}); chunk(WgInt<2>{}); chunk(WgInt<3>{});
        if (tracing) A.trace[((size_t) j * 8 + wave) * 3 + 1] = (long long) __builtin_amdgcn_s_memtime();
        const unsigned dst = ce.dst, ewrite = ce.ewrite;
        float out[TP];
        float mx = 0.0f;
#pragma unroll
        for (int t = 0; t < TP; ++t) {
#if !defined(MBAMD_WGX_NOEPI)
            out[t] = (run && t < T) ? f1[t / ACC][t % ACC] * f2[t / ACC][t % ACC] : 0.0f;
#else
            out[t] = (t == 0 && run) ? f1[0][0] + f2[0][0] : 0.0f;
#endif
            mx = fmaxf(mx, out[t]);
        }
        {   // the other states of this pattern sit TW lanes apart: lane swaps in the VALU, no LDS round trip
#if MBAMD_WG_TW == 16
            const auto sq = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sq[0]), __uint_as_float(sq[1]));
#endif
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const int wm = mode == SCALE_WRITE ? -1 : 0, rm = mode == SCALE_READ ? -1 : 0;
        const int e = (scale_exponent(mx) & wm) | (er & rm);
        er = er_next;
        const unsigned list = MBAMD_WG_LIST(ctl);
#pragma unroll
        for (int q = 0; q < MBAMD_WG_MAXLISTS; ++q) cum_e[q] += (list == (unsigned) q) ? (e & wm) : 0;
        vec ov[TV];
#pragma unroll
        for (int t = 0; t < TP; ++t) Vb::set(ov[t / V], t % V, scale_pow2(out[t], -e));   // (2^0 is exact: no branch)
#if !defined(MBAMD_WGX_NOLDS)
        if (ctl & MBAMD_W4_KEEP) {
            vec* keep = reinterpret_cast<vec*>(reinterpret_cast<char*>(slots) + ((ctl >> 16) & 0xFFu) * SLOTB);
#pragma unroll
            for (int i = 0; i < TV; ++i) keep[i * 64] = ov[i];
        }
#endif
        MBAMD_AS_GLOBAL vec* pd = reinterpret_cast<MBAMD_AS_GLOBAL vec*>((uintptr_t) (P0 + dst)) + lane;
#if !defined(MBAMD_WGX_NOSTORE)
#if defined(MBAMD_WGX_PLAINSTORE)
#pragma unroll
        for (int i = 0; i < TV; ++i) pd[i * 64] = ov[i];
#else
#pragma unroll
        for (int i = 0; i < TV; ++i) __builtin_nontemporal_store(ov[i], pd + i * 64);   // 64 * V * 4 contiguous bytes per instruction
#endif
#else
        if (Vb::get(ov[0], 0) == 123.456f) __builtin_nontemporal_store(ov[0], pd);
#endif
#if !defined(MBAMD_WGX_NOTINY)
        __builtin_nontemporal_store((int8_t) e, as_global(E0 + ewrite) + col);   // (every lane group holds the same e: no exec-mask branch)
#else
        if (e == 12345) __builtin_nontemporal_store((int8_t) e, as_global(E0 + ewrite) + col);
#endif
        if (tracing) A.trace[((size_t) j * 8 + wave) * 3 + 2] = (long long) (ctl & 0xFFFu);   // (flags and mode of the entry, for the reader)
    };
    // the sets rotate by NQ positions per entry; three entries bring every (NS <= 3) rotation back to the start
    for (int j = 0; j < n; j += 3) {
        step(DA, DB, DC, wg_pick<0>(X, Y, Z), wg_pick<1 % NS>(X, Y, Z), wg_pick<2 % NS>(X, Y, Z), j);
        step(DB, DC, DA, wg_pick<NQ % NS>(X, Y, Z), wg_pick<(NQ + 1) % NS>(X, Y, Z), wg_pick<(NQ + 2) % NS>(X, Y, Z), j + 1);
        step(DC, DA, DB, wg_pick<(2 * NQ) % NS>(X, Y, Z), wg_pick<(2 * NQ + 1) % NS>(X, Y, Z), wg_pick<(2 * NQ + 2) % NS>(X, Y, Z), j + 2);
    }
    // cumulative exponents of this workgroup's TW columns: the waves' sums meet in LDS, wave 0 owns the memory update
    int* const stage = reinterpret_cast<int*>(mine);
#pragma unroll
    for (int q = 0; q < MBAMD_WG_MAXLISTS; ++q) {
        if (A.cum[q] == nullptr || (A.lists > 1 && q != (int) list)) continue;      // (separate lists: a workgroup holds one list)
        int sum = cum_e[q];
        if (W > 1) {
            if (q > 0) __syncthreads();
            stage[lane] = sum;
            __syncthreads();
            if (wave == 0)
                for (int w = 1; w < W; ++w)
                    sum += reinterpret_cast<const int*>(reinterpret_cast<const char*>(lds_walkg) + (size_t) w * (MBAMD_WG_STAGE + (size_t) A.nslots * SLOTB))[lane];
        }
        if (wave == 0 && half == 0) {
            int32_t* d = A.cum[q] + (size_t) k * A.Ppad + (size_t) tile * TW + col;
            if (A.cumFresh >> q & 1) *d = sum;
            else if (sum != 0) *d += sum;
        }
    }
}
#endif

}  // namespace mbamd
#endif
