// mbamd_pathg_kernel.h -- k_pathg: a ROOT-WARD PATH of the general-state tree walk (round 5; the k_path4 of mbamd_walk4.h for 20 and
// 60..63 states).  The list of a move that dirtied one branch: every operation has the previous result as one child -- the CHAIN --
// and a sibling the list does not write (a buffer in HBM or a compact tip) as the other.  k_walkg runs such a list on ONE wave per
// (tile, category): both MFMA chains of every operation one after the other.  Only M1 x previous-result depends on the previous
// operation; the sibling's factor M2 x sibling is half of an operation's matrix work and independent of the chain.  Here a workgroup
// is TWO waves on one (32-pattern tile, category, list):
//   wave 1  forms the sibling factors of the operations, one after the other and ahead of the chain, into a ring of factor tiles in
//           LDS (a factor = the NT x 16 accumulator registers of every lane);
//   wave 0  runs the chain: its B operand is the previous result STILL IN ITS REGISTERS (the accumulator layout is the B layout,
//           mbamd_walkg.h), so an operation is one MFMA chain, the product with the factor from the ring, the rescale and the stores.
// The two waves keep in step through two counters in LDS (factors produced / factors consumed).
// Entries (Walk4Entry, as k_path4's): c1 = the chain's INPUT (entry 0 only: tip states or a buffer), c2 = the sibling, m1 / m2 their
// matrix buffers; ctl: TIP1 (entry 0), TIP2, [9:8] the scale mode, [11:10] the list.  Same arithmetic, operation by operation, as
// k_walkg: the same bits.  blockDim.x = 128; grid = walkg_grid(ntiles, K * lists); dynamic LDS = pathg_lds_bytes(S).
#ifndef MBAMD_PATHG_KERNEL_H_
#define MBAMD_PATHG_KERNEL_H_
namespace mbamd {

#define MBAMD_PG_RING 4          // factor tiles in flight between the two waves
__host__ __device__ inline size_t pathg_lds_bytes(int S) { return 64 + (size_t) MBAMD_PG_RING * wg_tiles(S) * 16 * 64 * sizeof(float); }

template <int SC, class ARGS = WalkGArgs>
__global__ void __launch_bounds__(128)
k_pathg(ARGS AA)
{
    const WalkGArgs& A = wg_args(AA);
    typedef WgShape<SC> Sh;
    typedef typename Sh::vec vec;
    typedef typename Sh::vecA vecA;
    typedef typename Sh::Vb Vb;
    typedef typename Sh::Va Va;
    typedef typename Sh::acc acc_t;
    constexpr int TW = Sh::TW, KS = Sh::KS, ACC = Sh::ACC;
    constexpr int T = Sh::T, NT = Sh::NT, V = Sh::V, VA = Sh::VA, TP = Sh::TP, NAP = Sh::NAP, NAV = NAP / VA, TV = TP / V;
    constexpr int NG = Sh::NG;                       // register groups a compact tip's gather needs
    constexpr bool BF = Sh::BF;
    constexpr int NKB = Sh::NKB, NGR = Sh::NGR;
    constexpr unsigned SLOTB = TP * 256u;
    const unsigned lane = threadIdx.x & 63, half = lane / TW, col = lane % TW;
    const int wave = mbd_wave_index();
    const unsigned K = (unsigned) A.K, KL = K * (unsigned) A.lists;
    const unsigned xcd = blockIdx.x & 7u, pos = blockIdx.x >> 3;
    const unsigned tile = (pos / KL) * 8u + xcd, k = (pos % KL) % K, list = (pos % KL) / K;
    if (tile >= (unsigned) A.ntiles) return;
    char* const lds = reinterpret_cast<char*>(mbd_dyn_lds<float>());
    typedef MBAMD_AS_LDS volatile int pg_vint;
    pg_vint* const produced = (pg_vint*) lds;                 // factors wave 1 has put into the ring
    pg_vint* const consumed = (pg_vint*) (lds + 4);           // factors wave 0 has taken out of it
    float* const ring = reinterpret_cast<float*>(lds + 64);   // [MBAMD_PG_RING][NT * 16][64]
    char* const P0 = reinterpret_cast<char*>(A.partials) + (size_t) tile * A.tileBytes + (size_t) k * SLOTB;
    const uint8_t* const T0 = A.tips + (size_t) tile * A.tipTileBytes;
    int8_t* const E0 = A.exps + (size_t) ((tile * TW) >> 6) * A.estride + (size_t) k * 64 + ((tile * TW) & 63u);
    const char* const Mk = reinterpret_cast<const char*>(A.matrices) + A.tabOff + (size_t) k * A.tabBytes;
    const Walk4Entry* const prog = wg_program(AA) + (size_t) list * A.entries;
    const int n = A.entries;

    if (lane == 0) { if (wave == 0) produced[0] = 0; else consumed[0] = 0; }
    MBAMD_SYNC();

    // a child's factor F[i][p] = sum_j P(i -> j) cl[j][p]: a compact tip gathers its column from the tip table, a buffer is a chain of
    // T MFMA steps over the A' table (k_walkg's `fetch` + `compute`, whole jobs at once: this wave has nothing else to hold)
    auto load_table = [&](unsigned moff, vecA (&a)[NAV]) {
        const MBAMD_AS_GLOBAL vecA* pa = reinterpret_cast<const MBAMD_AS_GLOBAL vecA*>((uintptr_t) (Mk + moff)) + lane;
#pragma unroll
        for (int i = 0; i < NAV; ++i) a[i] = pa[i * 64];
    };
    auto contract = [&](const vecA (&a)[NAV], const float (&b)[TP], acc_t (&f)[NT]) {
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int r = 0; r < ACC; ++r) f[it][r] = 0.0f;
        if constexpr (BF) {
            wg_contract_bf16<NT, NKB, TP>(a, b, f);      // (k_walkg's arithmetic: the same bits)
        } else {
#pragma unroll
            for (int tc = 0; tc < TP; ++tc)
                if (tc < T) {
#pragma unroll
                    for (int it = 0; it < NT; ++it) f[it] = mbd_mfma_f32_32x32x2(Va::get(a[(tc * NT + it) / VA], (tc * NT + it) % VA), b[tc], f[it]);
                }
        }
    };
    auto load_rows = [&](unsigned coff, float (&b)[TP]) {
        const MBAMD_AS_GLOBAL vec* pb = reinterpret_cast<const MBAMD_AS_GLOBAL vec*>((uintptr_t) (P0 + coff)) + lane;
#pragma unroll
        for (int i = 0; i < TV; ++i) {
            const vec v = pb[i * 64];
#pragma unroll
            for (int u = 0; u < V; ++u) b[i * V + u] = Vb::get(v, u);
        }
    };

    // One operand set = everything a factor needs from memory: the A' table (or, for a compact tip, its gather rows in the first NG
    // groups) and, for a buffer, its rows.  Both waves keep TWO sets: the next operation's operands are requested before the current
    // operation's MFMA chain starts -- a table is 16 KB per wave at 61 states and comes from HBM (the tables of a codon model are
    // 25 MB): requested any later, its round trip is the operation.
    struct Operands { vecA a[NAV]; float b[TP]; };
    auto request = [&](bool tip, unsigned moff, unsigned coff, Operands& o) {
        if (tip) {
            const unsigned s = as_global(T0 + coff)[col];
            const unsigned aoff = ((unsigned) NAP + (s / TW) * (unsigned) NGR) * 256u + ((s % TW) * KS + half) * (unsigned) (VA * 4);
            const MBAMD_AS_GLOBAL vecA* pa = reinterpret_cast<const MBAMD_AS_GLOBAL vecA*>((uintptr_t) (Mk + moff) + aoff);
#pragma unroll
            for (int i = 0; i < NG; ++i) o.a[i] = pa[i * 64];
        } else {
            load_table(moff, o.a);
            load_rows(coff, o.b);
        }
    };
    auto factor = [&](bool tip, const Operands& o, acc_t (&f)[NT]) {
        if (tip) {
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < ACC; ++r) f[it][r] = (ACC * it + r < T) ? Va::get(o.a[(r * NT + it) / VA], (r * NT + it) % VA) : 0.0f;
        } else {
            contract(o.a, o.b, f);
        }
    };

    if (wave == 1) {
        // ---- the sibling factors, ahead of the chain
        Operands X, Y;
        Walk4Entry e = walk4_load_entry(prog);
        request((e.ctl & MBAMD_W4_TIP2) != 0, e.m2, e.c2, X);
        auto produce = [&](int j, const Operands& cur, Operands& nxt) {
            const Walk4Entry en = walk4_load_entry(prog + (j + 1 < n ? j + 1 : j));
            if (j + 1 < n) request((en.ctl & MBAMD_W4_TIP2) != 0, en.m2, en.c2, nxt);
            acc_t f[NT];
            factor((e.ctl & MBAMD_W4_TIP2) != 0, cur, f);
            while (j - mbd_uniform(consumed[0]) >= MBAMD_PG_RING) MBD_SPIN_PAUSE();     // the slot's last factor has been taken
            MBD_COMPILER_FENCE();
            float* slot = ring + (size_t) (j % MBAMD_PG_RING) * (NT * 16 * 64) + lane;
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < ACC; ++r) slot[(it * 16 + r) * 64] = f[it][r];
            MBAMD_WAVE_SYNC();                       // (every lane's part of the factor is in front of the counter)
            produced[0] = j + 1;
            MBD_COMPILER_FENCE();
            e = en;
        };
        for (int j = 0; j < n; j += 2) {
            produce(j, X, Y);
            if (j + 1 < n) produce(j + 1, Y, X);
        }
        return;
    }

    // ---- the chain
    float prev[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) prev[t] = 0.0f;
    int cum_e = 0;
    Walk4Entry e = walk4_load_entry(prog);
    Operands X, Y;                                   // (of the chain's operations only the tables are used: its rows are `prev`)
    const bool tipInput = (e.ctl & MBAMD_W4_TIP1) != 0;
    request(tipInput, e.m1, e.c1, X);
    if (!tipInput) {
#pragma unroll
        for (int t = 0; t < TP; ++t) prev[t] = X.b[t];
    }
    auto link = [&](int j, const Operands& cur, Operands& nxt) {
        const Walk4Entry en = walk4_load_entry(prog + (j + 1 < n ? j + 1 : j));
        const unsigned ctl = e.ctl;
        const unsigned mode = (ctl >> 8) & 3u;
        int er = 0;
        if (mode == SCALE_READ) er = as_global(E0 + e.eread)[col];
        if (j + 1 < n) load_table(en.m1, nxt.a);     // the next operation's table: in flight under this operation's MFMA chain
        acc_t f1[NT], f2[NT];
        if (j == 0 && tipInput) factor(true, cur, f1);
        else contract(cur.a, prev, f1);
        while (mbd_uniform(produced[0]) <= j) MBD_SPIN_PAUSE();
        MBD_COMPILER_FENCE();
        {
            const float* slot = ring + (size_t) (j % MBAMD_PG_RING) * (NT * 16 * 64) + lane;
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < ACC; ++r) f2[it][r] = slot[(it * 16 + r) * 64];
        }
        MBAMD_WAVE_SYNC();                           // (every lane has its factor: the slot may be written again)
        consumed[0] = j + 1;
        MBD_COMPILER_FENCE();
        float mx = 0.0f;
#pragma unroll
        for (int t = 0; t < TP; ++t) {
            prev[t] = (t < T) ? f1[t / ACC][t % ACC] * f2[t / ACC][t % ACC] : 0.0f;
            mx = fmaxf(mx, prev[t]);
        }
        mx = mbd_max_lane_xor32(mx);
        const int wm = mode == SCALE_WRITE ? -1 : 0, rm = mode == SCALE_READ ? -1 : 0;
        const int ex = (scale_exponent(mx) & wm) | (er & rm);
        cum_e += ex & wm;
        const float sc = mbd_pow2(-ex);
        MBAMD_AS_GLOBAL vec* pd = reinterpret_cast<MBAMD_AS_GLOBAL vec*>((uintptr_t) (P0 + e.dst)) + lane;
#pragma unroll
        for (int i = 0; i < TV; ++i) {
            vec ov;
#pragma unroll
            for (int u = 0; u < V; ++u) { prev[i * V + u] *= sc; Vb::set(ov, u, prev[i * V + u]); }
            __builtin_nontemporal_store(ov, pd + i * 64);
        }
        __builtin_nontemporal_store((int8_t) ex, as_global(E0 + e.ewrite) + col);
        e = en;
    };
    for (int j = 0; j < n; j += 2) {
        link(j, X, Y);
        if (j + 1 < n) link(j + 1, Y, X);
    }
    // cumulative exponents of this tile's columns (one chain wave per (tile, category, list))
    if (A.cum[list] != nullptr && half == 0) {
        int32_t* d = A.cum[list] + (size_t) k * A.Ppad + (size_t) tile * TW + col;
        if (A.cumFresh >> list & 1) *d = cum_e;
        else if (cum_e != 0) *d += cum_e;
    }
}

}  // namespace mbamd
#endif
