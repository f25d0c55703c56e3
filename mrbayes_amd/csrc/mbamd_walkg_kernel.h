// mbamd_walkg_kernel.h -- k_walkg, the 20/61-state tree-walk kernel on the matrix cores: the kernel half of mbamd_walkg.h (which
// holds the layouts, tables and arguments, shared with the host).  Written against the primitives of <mbamd_dev_walkg_kernel.h>
// (csrc/device/ for gfx950; the TEST-ONLY host emulation supplies the same names on fibers), so this file is compiled into the
// product AND run, as is, by the CPU CI.
#ifndef MBAMD_WALKG_KERNEL_H_
#define MBAMD_WALKG_KERNEL_H_
namespace mbamd {

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
    static constexpr bool BF = SC >= MBAMD_WG_BF_MIN;   // wg_bf16: the contraction on v_mfma_f32_32x32x16_bf16, three bf16 pieces per value (mbamd_walkg.h)
    static constexpr int NKB = (SC + 15) / 16;       // K-blocks of 16 states (BF)
    static constexpr int V = SC > 32 ? 4 : 2, VA = BF ? 4 : V;
    static constexpr int ACC = 16;                   // accumulator registers per output tile (32 x 32 / 64 lanes)
    static constexpr int TP = (T + V - 1) / V * V, NAP = BF ? 4 * 3 * NT * NKB : (TP * NT + VA - 1) / VA * VA;
    static constexpr int NGR = BF ? ((T < ACC ? T : ACC) * NT + 3) / 4 * 4 : NAP;   // rows of one tip-gather table
    static constexpr int NG = ((T < ACC ? T : ACC) * NT + VA - 1) / VA;             // register groups a compact tip's gather needs
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
// the three bf16 pieces of two values, each pair in one dword (v0's piece in bits 0..15, v1's in 16..31): p + q + w = v EXACTLY --
// a remainder of a 24-bit value after its nearest 8-bit neighbour has 16 significant bits, after the next 8 (mbamd_walkg.h: the
// same arithmetic as wg_table_put's, which makes the A pieces)
__device__ __forceinline__ void wg_split_pair(float v0, float v1, unsigned& p, unsigned& q, unsigned& w)
{
    p = mbd_cvt_pk_bf16(v0, v1);
    const float r0 = v0 - __builtin_bit_cast(float, p << 16), r1 = v1 - __builtin_bit_cast(float, p & 0xFFFF0000u);
    q = mbd_cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, q << 16), s1 = r1 - __builtin_bit_cast(float, q & 0xFFFF0000u);
    w = mbd_cvt_pk_bf16(s0, s1);
}
// One child factor (or one chunk of its K-blocks) on the 16-bit matrix cores: f[it] += sum over the chunk's K-blocks, K-BLOCK BY
// K-BLOCK, and inside a K-block over the six piece pairs SMALLEST PRODUCTS FIRST (the small ones meet while the accumulator's own
// contribution from this K-block is still small).  The order does not depend on how a kernel cuts a child into chunks: k_walkg,
// k_walkb and k_pathg give the same bits.
//   a     the chunk's table operands, index (kb NT + it) 3 + piece (wg_table_put)
//   rows  the child's block rows of the chunk: K-block kb = rows 8 kb .. 8 kb + 7 (NR of them exist; the rest are zero)
template <int NT, int NKBC, int NR, class VECA>
__device__ __forceinline__ void wg_contract_bf16(const VECA* a, const float* rows, mbd_acc16 (&f)[NT])
{
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1
#pragma unroll
    for (int kb = 0; kb < NKBC; ++kb) {
        mbd_f4 pc[3];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int t0 = 8 * kb + 2 * jj;
            unsigned p = 0, q = 0, w = 0;
            if (t0 < NR) wg_split_pair(rows[t0 < NR ? t0 : 0], t0 + 1 < NR ? rows[t0 + 1 < NR ? t0 + 1 : 0] : 0.0f, p, q, w);
            pc[0][jj] = __builtin_bit_cast(float, p); pc[1][jj] = __builtin_bit_cast(float, q); pc[2][jj] = __builtin_bit_cast(float, w);
        }
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int it = 0; it < NT; ++it) f[it] = mbd_mfma_bf16_32x32x16(a[(kb * NT + it) * 3 + PA[pr]], pc[PB[pr]], f[it]);
    }
}
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

// Second launch bound = waves per SIMD the register budget must allow.  Beyond 32 states "two" makes the register file ONE file
// (206 registers, accumulators included) instead of 174 + 32 accumulation registers: a compact tip's factor is then written where
// the product reads it, not copied in through v_accvgpr_write (codon 100 x 5 000: 0.195 -> 0.188 ms, profiles/r04_exp_walkgs.txt).
#if !defined(MBAMD_WG_MINWAVES)
#define MBAMD_WG_MINWAVES(SC) ((SC) > 48 ? 2 : 1)     // (40 states, bf16 tables: 72 + 20 operand registers per set -- one wave per SIMD, no scratch)
#endif
// blockDim.x = 64 * W (W <= WMAX); grid = walkg_grid(ntiles, K); dynamic LDS = wg_lds_bytes(W, nslots, SC).
// The operand pipeline works in CHUNKS: a job (one child factor, T MFMA steps per row tile) is CH chunks, an entry 2 CH,
// and the operands of a chunk are fetched DEPTH chunks ahead into one of DEPTH + 1 rotating register sets.
//   20 states: CH 1, DEPTH 2 -- a job is 10 MFMAs = 640 cycles, less than a memory round trip; three sets of 15 registers;
//   61 states: CH 2, DEPTH 1 -- a chunk is 31 MFMAs = 2000 cycles; two sets of 48 registers fit beside the 64 accumulators
//              (whole jobs did not: the allocator shuttled LOADED operands through AccVGPRs, a vmcnt(0) per job).
template <int SC, int WMAX, int CH, int DEPTH, class ARGS = WalkGArgs>
__global__ void __launch_bounds__(64 * WMAX, MBAMD_WG_MINWAVES(SC))
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
    constexpr bool BF = Sh::BF;
    constexpr int NKB = Sh::NKB, NKBC = NKB / CH, NGR = Sh::NGR, NG = Sh::NG;
    constexpr int TPC = TP / CH, NAVC = NAV / CH, TVC = TV / CH;      // per chunk: block rows (fp32 mode: MFMA steps), A register groups, B register groups
    constexpr int NQ = 2 * CH, NS = DEPTH + 1;                        // chunks per entry, register sets
    static_assert(TP % CH == 0 && TPC % V == 0 && NAV % CH == 0 && NS <= 3 && DEPTH <= NQ && TPC <= 20, "chunk geometry");
    static_assert(BF || (TPC * NT) % VA == 0, "chunk geometry (fp32 tables)");
    static_assert(!BF || (NKB % CH == 0 && (CH == 1 || TP == 8 * NKB)), "chunk geometry (bf16 tables: whole K-blocks)");
    static_assert(NG <= NAVC, "a compact tip's gather rows must lie in the first chunk");
    constexpr unsigned SLOTB = TP * 256u;
    const unsigned lane = threadIdx.x & 63, half = lane / TW, col = lane % TW;      // half: which of the KS states of a row
    int wave = mbd_wave_index();
    int W = (int) (blockDim.x >> 6);
    // Two-wave workgroups land on one SIMD pair of the CU and leave the other pair's matrix cores idle (measured: 73 against
    // 140 TFLOP/s of dense MFMA with waves {0, 1} against {0, 2} of a four-wave workgroup): such workgroups are launched
    // with twice the waves, and the odd ones leave at once.
    if (A.spread) {
        if (wave & 1) return;
        wave >>= 1;
        W >>= 1;
    }
    float* const lds_walkg = mbd_dyn_lds<float>();
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
        // (a compact tip takes every register from the FIRST chunk's rows: where a job has several chunks -- 60..63 states -- its
        //  later chunks and no-op entries keep their loads, so that the vector-memory sequence stays uniform, but every lane reads one
        //  and the same 16 bytes: one L2 request instead of eight lines.  A quarter of the kernel's L2 traffic, which bounds it
        //  (profiles/r03_c5_pmc.txt).  With one chunk per job the plain form stays: a run-time stride costs the immediate offsets
        //  of the loads -- +10 % at 20 states, measured.)
        const bool idle = CH > 1 && ((tip && h > 0) || (ctl & MBAMD_W4_NOP));
        const unsigned aoff = idle ? 0u : (tip ? ((unsigned) NAP + (s / TW) * (unsigned) NGR) * 256u + ((s % TW) * KS + half) * (unsigned) (VA * 4) : lane * (unsigned) (VA * 4));
        const MBAMD_AS_GLOBAL vecA* pa = reinterpret_cast<const MBAMD_AS_GLOBAL vecA*>((uintptr_t) (Mk + moff) + aoff) + (idle ? 0 : h * NAVC * 64);
        if constexpr (BF) {
            // bf16 tables: an interior child's operands are NAVC loads, a tip's gather rows the first NG of them (first chunk only).
            // All of them go through a window on the child's table (mbd_buf): one 32-bit lane offset instead of a 64-bit address per
            // load, and the loads a chunk does not need keep their place in the sequence with a lane offset OUTSIDE the window --
            // zeros, no memory request.
            const bool none = (tip && h > 0) || (ctl & MBAMD_W4_NOP);
            const mbd_buf win = mbd_make_buffer(Mk + moff, A.tabBytes);
            const unsigned front = none ? MBD_OUTSIDE : (tip ? (s / TW) * (unsigned) (NGR * 256) + ((s % TW) * KS + half) * 16u : lane * 16u);
            const unsigned back = (none || tip) ? MBD_OUTSIDE : lane * 16u;
            const unsigned base = tip ? (unsigned) NAP * 256u : (unsigned) (h * NAVC) * 1024u;
#pragma unroll
            for (int i = 0; i < NG && i < NAVC; ++i) o.a[i] = mbd_buffer_load_f4(win, front, base + i * 1024u);
#pragma unroll
            for (int i = NG; i < NAVC; ++i) o.a[i] = mbd_buffer_load_f4(win, back, base + i * 1024u);
        } else if constexpr (CH > 1) {
            const int stride = idle ? 0 : 64;
#pragma unroll
            for (int i = 0; i < NAVC; ++i) o.a[i] = pa[i * stride];
        } else {
#pragma unroll
            for (int i = 0; i < NAVC; ++i) o.a[i] = pa[i * 64];
        }
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
        // (5..8 states, two row registers per set: "a load from the set or a load from the slot" became ONE load through a pointer that is
        //  either, and the three sets lived in scratch memory from then on -- 64 bytes per lane.  The set's rows are taken outside the
        //  branch, as values the optimiser cannot turn back into loads.)
        vec ob[TVC];
#pragma unroll
        for (int i = 0; i < TVC; ++i) { ob[i] = o.b[i]; if constexpr (TVC == 2) MBD_OPAQUE_VGPR(ob[i]); }
        if (de.ctl & (ch ? MBAMD_WG_MEM2 : MBAMD_WG_MEM1)) {
#pragma unroll
            for (int i = 0; i < TVC; ++i) b[i] = ob[i];
        } else {
            const vec* sl = reinterpret_cast<const vec*>(reinterpret_cast<const char*>(slots) + (ch ? de.c2 : de.c1)) + h * TVC * 64;
#pragma unroll
            for (int i = 0; i < TVC; ++i) b[i] = sl[i * 64];
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
        if constexpr (BF) {
            float rows[TPC];
#pragma unroll
            for (int tc = 0; tc < TPC; ++tc) rows[tc] = Vb::get(b[tc / V], tc % V);
            wg_contract_bf16<NT, NKBC, TPC>(o.a, rows, f);
        } else {
#pragma unroll
            for (int tc = 0; tc < TPC; ++tc)
                if (h * TPC + tc < T) {
#pragma unroll
                    for (int it = 0; it < NT; ++it) {
                        const float av = Va::get(o.a[(tc * NT + it) / VA], (tc * NT + it) % VA), bv = Vb::get(b[tc / V], tc % V);
                        f[it] = mbd_mfma_f32_32x32x2(av, bv, f[it]);
                    }
                }
        }
    };

    // One iteration = one entry `cur`: its NQ chunks run on the register sets (S0, S1, S2, S0, ...) while the chunks DEPTH
    // further on -- of cur, then of entry `n1` -- are fetched; the tip states of entry `n2` are fetched, and cur's descriptor
    // is replaced by entry j + 3.  Vector-memory sequence, identical on every path:
    //     NQ x NAVC operand loads | TV + 1 stores      (+ the rare conditional loads)
    auto step = [&](WgDesc& cur, const WgDesc& n1, WgDesc& n2, Ops& S0, Ops& S1, Ops& S2, int j) {
        const unsigned ctl = cur.e.ctl;
        if (ctl & MBAMD_W4_BARRIER) {
            // values other waves produced in the previous phase are read from here on: drain this wave's stores, meet
            MBD_DRAIN_ALL();
            MBD_WG_BARRIER();
            MBD_COMPILER_FENCE();
        }
        const bool run = !(ctl & MBAMD_W4_NOP);
        const unsigned mode = (ctl >> 8) & 3u;
        // (issued before the operand fetches: what the next entry needs first must not queue behind them)
        // (three tiny loads, unconditional: every conditional vector-memory instruction makes the compiler's vmcnt waits one
        //  instruction stricter -- and the instruction they then wait for is the oldest STORE of the previous entry)
        // Exponents of the next entry, tip states of the entry after: 32 bytes per tile each.  16 / 20 states take them through
        // the SCALAR path -- one s_load_dwordx8 each, issued with the descriptor load below, every lane picks its byte in the
        // epilogue: as vector loads they sit in the in-order vmcnt queue behind the previous entry's result stores and the
        // gather addresses of a tip's operand fetch wait for them (-5 % at 20 states, -7 % at 16).  At 61 states the 24 extra
        // scalar registers spill (+3 %), at 8 the entry is too short to cover the scalar latency (+6 %): those keep the vector
        // loads (profiles/r03_exp_walkg_tiny.txt).
        constexpr bool SCALAR_TINY = SC >= 16 && SC <= 32;
        typedef unsigned wg_u8v __attribute__((ext_vector_type(8)));
        wg_u8v tinyE = {}, tiny1 = {}, tiny2 = {};
        auto tiny_issue = [&]() {
            if constexpr (SCALAR_TINY) {
                tinyE = *reinterpret_cast<const MBAMD_AS_CONST wg_u8v*>((uintptr_t) (E0 + n1.e.eread));
                tiny1 = *reinterpret_cast<const MBAMD_AS_CONST wg_u8v*>((uintptr_t) (T0 + ((n2.e.ctl & MBAMD_W4_TIP1) ? n2.e.c1 : 0u)));
                tiny2 = *reinterpret_cast<const MBAMD_AS_CONST wg_u8v*>((uintptr_t) (T0 + ((n2.e.ctl & MBAMD_W4_TIP2) ? n2.e.c2 : 0u)));
            }
        };
        auto tiny_pick = [&](const wg_u8v& w) {
            const unsigned sel = col >> 2;
            unsigned d = w[0];
#pragma unroll
            for (unsigned i = 1; i < 8; ++i) d = sel == i ? w[i] : d;
            return (d >> ((col & 3u) * 8u)) & 0xFFu;
        };
        int er_next = 0;
        if constexpr (!SCALAR_TINY) {
            er_next = as_global(E0 + n1.e.eread)[col];
            n2.s1 = as_global(T0 + ((n2.e.ctl & MBAMD_W4_TIP1) ? n2.e.c1 : 0u))[col];
            n2.s2 = as_global(T0 + ((n2.e.ctl & MBAMD_W4_TIP2) ? n2.e.c2 : 0u))[col];
        }
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
                        for (int i = 0; i < TVC; ++i) MBD_PIN_VGPR(b[i]);
                        cur.e = walk4_load_entry(prog + j + 3);
                        tiny_issue();
                    }
                    compute(false, q, wg_pick<q % NS>(S0, S1, S2), b, q < CH ? f1 : f2);
                } else {
                    if constexpr (q == NQ - 1) {
                        cur.e = walk4_load_entry(prog + j + 3);
                        tiny_issue();
                    }
                    vec b[TVC];
                    if (run) compute(true, q, wg_pick<q % NS>(S0, S1, S2), b, q < CH ? f1 : f2);
                }
            }
        };
        chunk(WgInt<0>{}); chunk(WgInt<1>{}); chunk(WgInt<2>{}); chunk(WgInt<3>{});
        const unsigned dst = ce.dst, ewrite = ce.ewrite;
        float out[TP];
        float mx = 0.0f;
#pragma unroll
        for (int t = 0; t < TP; ++t) {
            out[t] = (run && t < T) ? f1[t / ACC][t % ACC] * f2[t / ACC][t % ACC] : 0.0f;
            mx = fmaxf(mx, out[t]);
        }
        mx = mbd_max_lane_xor32(mx);                 // the other states of this pattern sit 32 lanes apart: a lane swap in the VALU, no LDS round trip
        if constexpr (SCALAR_TINY) {
            er_next = (int) (int8_t) tiny_pick(tinyE);
            n2.s1 = tiny_pick(tiny1);
            n2.s2 = tiny_pick(tiny2);
        }
        const int wm = mode == SCALE_WRITE ? -1 : 0, rm = mode == SCALE_READ ? -1 : 0;
        const int e = (scale_exponent(mx) & wm) | (er & rm);
        er = er_next;
        const unsigned list = MBAMD_WG_LIST(ctl);
#pragma unroll
        for (int q = 0; q < MBAMD_WG_MAXLISTS; ++q) cum_e[q] += (list == (unsigned) q) ? (e & wm) : 0;
        const float sc = mbd_pow2(-e);
        vec ov[TV];
#pragma unroll
        for (int t = 0; t < TP; ++t) Vb::set(ov[t / V], t % V, out[t] * sc);   // (exact: |e| <= 126; 2^0 needs no branch)
        if (ctl & MBAMD_W4_KEEP) {
            vec* keep = reinterpret_cast<vec*>(reinterpret_cast<char*>(slots) + ((ctl >> 16) & 0xFFu) * SLOTB);
#pragma unroll
            for (int i = 0; i < TV; ++i) keep[i * 64] = ov[i];
        }
        // (a no-operation entry -- padding of the shorter programs of a workgroup -- keeps its TV stores, so that the vector-memory sequence
        //  stays the same on every path, but all its lanes write one and the same 16 bytes of the tile's spare buffer, a different place from
        //  entry to entry: TV lines instead of TV KiB.  The zeros of these entries were 12-19 % of the kernel's HBM writes at 61 states.)
        MBAMD_AS_GLOBAL vec* pd = reinterpret_cast<MBAMD_AS_GLOBAL vec*>((uintptr_t) (P0 + dst)) + (run ? lane : ((unsigned) j & 63u));
#pragma unroll
        for (int i = 0; i < TV; ++i) __builtin_nontemporal_store(ov[i], pd + i * 64);   // 64 * V * 4 contiguous bytes per instruction
        __builtin_nontemporal_store((int8_t) e, as_global(E0 + ewrite) + col);   // (every lane group holds the same e: no exec-mask branch)
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
            if (q > 0) MBAMD_SYNC();
            stage[lane] = sum;
            MBAMD_SYNC();
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
#if defined(MBAMD_WG_ABL_TAIL_FENCE)
    // (experiment, call 30: what a device-scope release + one atomic per workgroup costs at the end of the walk -- the entry fee of
    //  an integration run by the last workgroup of a tile; A.reserved = a counter per tile)
    if (A.reserved != nullptr) {
        __threadfence();
        if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(A.reserved) + tile, 1ull);
    }
#endif
}
}  // namespace mbamd
#endif
