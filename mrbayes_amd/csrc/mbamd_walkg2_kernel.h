// mbamd_walkg2_kernel.h -- k_walkg2: the general-state tree walk of mbamd_walkg.h with the operands of a whole ENTRY in
// flight (round 5).  Same programs (Walk4Entry), arenas, tables, slots and decomposition as k_walkg; what differs is the pipeline:
//
//   * What round 5 measured (profiles/r05_mfma_shadow.txt, r05_walkg_ablation.txt): for fp32 the matrix core and the vector ALU
//     are ONE issue resource of a SIMD -- while a wave issues back-to-back MFMAs every other wave of the SIMD stalls at its next
//     VALU instruction, whatever the wave priorities -- so a second wave per SIMD hides latency, never VALU work; and k_walkg's
//     waves spend ~4 400 cycles per entry OUTSIDE their MFMA chains (operand waits of a pipeline two CHUNKS deep -- two tip
//     children are two short chunks --, copies between its three rotating register sets, LDS round trips one after the other).
//     The time of the kernel is one wave's chain: MFMA + that.
//   * Here an entry's operands (the A' rows or tip gather rows of BOTH children) are requested at the start of the entry before,
//     into one of two register sets that alternate with the loop's two halves; the factors of a tip are used where they landed
//     (one code path per combination of child kinds, no copies); a child's B rows are read from LDS in two halves, the second
//     requested while the first half's MFMAs run; the epilogue works on float pairs (v_pk_mul_f32, v_max3_f32).
//   * Row split (wg_split: 60..63 states): a (tile, category, subtree bin) is a PAIR of waves, wave h owning output tile h -- half
//     of A', of the MFMA chain, of the accumulators, of the epilogue, of the stores; the child's rows come from the slots the pair
//     shares.  That halves the registers of an entry's operands (two whole sets fit) and puts two working waves on every SIMD.
//     The pair keeps in step through two progress counters in LDS (see `progress` below) and exchanges its column maxima.
//
// blockDim.x = 64 * W * wg_waves_per_bin(S) (* 2 with `spread`); grid = walkg_grid(ntiles, K * lists); dynamic LDS = wg_lds_bytes.
// Programs: no leading NOPs, padded to a multiple of 2, MBAMD_WG2_TAIL trailing NOPs (descriptor read-ahead).
#ifndef MBAMD_WALKG2_KERNEL_H_
#define MBAMD_WALKG2_KERNEL_H_
namespace mbamd {

#if !defined(MBAMD_WG2_MINWAVES)
#define MBAMD_WG2_MINWAVES(SC) ((SC) > 32 ? 2 : 3)   // waves per SIMD the register budget must allow
#endif
template <int SC, bool PAIR> struct Wg2Ops {
    typedef WgShape<SC, PAIR> Sh;
    typename Sh::vecA a1[Sh::NAW / Sh::VA];          // child 1: its A' rows (this wave's output tile), or the tip's gather rows
    typename Sh::vecA a2[Sh::NAW / Sh::VA];
};

template <int SC, int WMAX, bool PAIR, class ARGS = WalkGArgs>
__global__ void __launch_bounds__(64 * WMAX, MBAMD_WG2_MINWAVES(SC))
k_walkg2(ARGS AA)
{
    const WalkGArgs& A = wg_args(AA);
    typedef WgShape<SC, PAIR> Sh;
    typedef typename Sh::vec vec;
    typedef typename Sh::vecA vecA;
    typedef typename Sh::Vb Vb;
    typedef typename Sh::Va Va;
    typedef typename Sh::acc acc_t;
    typedef Wg2Ops<SC, PAIR> Ops;
    static_assert(Sh::SPLIT || Sh::NT == 1, "k_walkg2: one output tile per wave");
    constexpr bool SPLIT = Sh::SPLIT;
    constexpr int TW = 32, KS = 2, ACC = Sh::ACC;
    constexpr int T = Sh::T, V = Sh::V, VA = Sh::VA, TP = Sh::TP, NAP = Sh::NAP, TV = TP / V;
    constexpr int NA = Sh::NAW / VA;                              // register groups (memory instructions) of a child's operand rows
    constexpr int NG = ((ACC < T ? ACC : T) + VA - 1) / VA;       // ... of which a compact tip's gather needs the first NG
    constexpr int TPO = Sh::TPO, TVO = TPO / V;                   // block rows of a result this wave owns
    constexpr int TH = (TV + 1) / 2;                              // a child's rows are read in two halves: TH groups, then TV - TH
    constexpr unsigned SLOTB = TP * 256u;
    constexpr unsigned STAGE = SPLIT ? MBAMD_WG_STAGE_SPLIT : MBAMD_WG_STAGE;
    static_assert(NG <= NA && TPO % V == 0, "operand geometry");

    const unsigned lane = threadIdx.x & 63, half = lane / TW, col = lane % TW;
    int wave = mbd_wave_index();
    int W = (int) (blockDim.x >> 6);
    if (A.spread) {                                  // two-wave workgroups are launched as four (see k_walkg)
        if (wave & 1) return;
        wave >>= 1;
        W >>= 1;
    }
    const int hw = SPLIT ? (wave & 1) : 0;           // row split: the output tile this wave owns; from here on `wave`, `W` count bins
    if constexpr (SPLIT) { wave >>= 1; W >>= 1; }
    float* const lds_walkg = mbd_dyn_lds<float>();
    const unsigned K = (unsigned) A.K, KL = K * (unsigned) A.lists;
    const unsigned xcd = blockIdx.x & 7u, pos = blockIdx.x >> 3;
    const unsigned tile = (pos / KL) * 8u + xcd, k = (pos % KL) % K, list = (pos % KL) / K;
    if (tile >= (unsigned) A.ntiles) return;
    char* const mine = reinterpret_cast<char*>(lds_walkg) + (size_t) wave * (STAGE + (size_t) A.nslots * SLOTB);
    vec* const slots = reinterpret_cast<vec*>(mine + STAGE) + lane;
    char* const P0 = reinterpret_cast<char*>(A.partials) + (size_t) tile * A.tileBytes + (size_t) k * SLOTB;
    const uint8_t* const T0 = A.tips + (size_t) tile * A.tipTileBytes;
    int8_t* const E0 = A.exps + (size_t) ((tile * TW) >> 6) * A.estride + (size_t) k * 64 + ((tile * TW) & 63u);
    const char* const Mk = reinterpret_cast<const char*>(A.matrices) + A.tabOff + (size_t) k * A.tabBytes +
                           (SPLIT ? (size_t) hw * (TP * 256) : 0);                         // (row split: this wave's rows of every table)
    const Walk4Entry* const prog = wg_program(AA) + ((size_t) list * W + wave) * A.entries;
    const int n = A.entries - MBAMD_WG2_TAIL;

    // ---- row split: how the two waves of a pair keep in step ----------------------------------------------------------------
    // Each wave publishes a progress counter in LDS (its DS operations execute in program order: whatever it wrote before the
    // counter is visible to whoever sees the counter) and polls the other's.  Both run the same program, so "what the partner
    // must have reached" is what this wave has reached itself.  Two steps per operation:
    //   2 n + 1  the MFMA chains of operation n are done (nothing of its children will be read any more) and this wave's column
    //            maxima are in the exchange area                      -> the partner may take the maxima and overwrite child slots
    //   2 n + 2  this wave's rows of result n are in their slot       -> the partner may read result n
    // (an LDS pointer in so many words: address-space inference leaves volatile accesses flat)
    typedef MBAMD_AS_LDS volatile int wg_lds_vint;
    float* const xmax = reinterpret_cast<float*>(mine + 256);
    wg_lds_vint* const progress = (wg_lds_vint*) (mine + 768);
    int sig = 0;
    bool drained = false;
    auto pair_signal = [&]() {
        MBAMD_WAVE_SYNC();                           // (every lane's LDS writes are in front of the counter)
        progress[hw] = ++sig;
        MBD_COMPILER_FENCE();
    };
    auto pair_wait = [&]() {
        while (mbd_uniform(progress[hw ^ 1]) < sig) MBD_SPIN_PAUSE();
        MBD_COMPILER_FENCE();
    };
    if constexpr (SPLIT) {
        progress[hw] = 0;
        MBAMD_SYNC();
    }

    // the operand rows of entry e (both children) -> register set o: 2 NA loads on every path
    auto fetch = [&](const Walk4Entry& e, unsigned s1, unsigned s2, Ops& o) {
        const unsigned ctl = e.ctl;
        const bool nop = ctl & MBAMD_W4_NOP;
        auto child = [&](bool tip, unsigned moff, unsigned s, vecA (&a)[NA]) {
            const MBAMD_AS_GLOBAL vecA* base = reinterpret_cast<const MBAMD_AS_GLOBAL vecA*>((uintptr_t) (Mk + (nop ? 0u : moff)));
            if (nop) {
                // (the sequence of vector-memory instructions stays the same on every path, for the wait counts: a no-operation
                //  entry reads NA pieces of one cache line, every lane the same -- distinct addresses, or the compiler makes ONE
                //  load of them, waits for it on the spot and copies)
#pragma unroll
                for (int i = 0; i < NA; ++i) a[i] = base[i];
            } else if (tip) {
                // gather table u = s / 32: the lane's column 2 (s % 32) + half of the rows n = r; the rest of the sequence as above
                const MBAMD_AS_GLOBAL vecA* g = base + (1u + s / TW) * (unsigned) (NAP * 64 / VA) + ((s % TW) * KS + half);
#pragma unroll
                for (int i = 0; i < NA; ++i) a[i] = i < NG ? g[i * 64] : base[i];
            } else {
                const MBAMD_AS_GLOBAL vecA* p = base + lane;
#pragma unroll
                for (int i = 0; i < NA; ++i) a[i] = p[i * 64];
            }
        };
        child(ctl & MBAMD_W4_TIP1, e.m1, s1, o.a1);
        child(ctl & MBAMD_W4_TIP2, e.m2, s2, o.a2);
    };
    // the rows of a child that lives in HBM (result of an earlier launch, of another bin, or evicted): all of them, into registers
    auto fetch_mem = [&](unsigned coff, vec (&bm)[TV]) {
        const MBAMD_AS_GLOBAL vec* pb = reinterpret_cast<const MBAMD_AS_GLOBAL vec*>((uintptr_t) (P0 + coff)) + lane;
#pragma unroll
        for (int i = 0; i < TV; ++i) bm[i] = pb[i * 64];
    };
    // one child factor on the matrix core: f = A' (this wave's output tile) x the child's rows -- from its LDS slot in two halves (the
    // second requested as soon as the first half's MFMAs are under way), or from the registers `bm`
    auto factor = [&](bool mem, unsigned coff, const vecA (&a)[NA], const vec (&bm)[TV], acc_t& f) {
#pragma unroll
        for (int r = 0; r < ACC; ++r) f[r] = 0.0f;
        if (mem) {
#pragma unroll
            for (int t = 0; t < T; ++t)
                f = mbd_mfma_f32_32x32x2(Va::get(a[t / VA], t % VA), Vb::get(bm[t / V], t % V), f);
            return;
        }
        const vec* sl = reinterpret_cast<const vec*>(reinterpret_cast<const char*>(slots) + coff);
        vec b0[TH], b1[TV - TH > 0 ? TV - TH : 1];
#pragma unroll
        for (int i = 0; i < TH; ++i) b0[i] = sl[i * 64];
#pragma unroll
        for (int i = 0; i < TV - TH; ++i) b1[i] = sl[(TH + i) * 64];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float bv = t < TH * V ? Vb::get(b0[t / V], t % V) : Vb::get(b1[(t - TH * V) / V], (t - TH * V) % V);
            f = mbd_mfma_f32_32x32x2(Va::get(a[t / VA], t % VA), bv, f);
        }
    };

    Walk4Entry d0 = walk4_load_entry(prog), d1 = walk4_load_entry(prog + 1), d2 = walk4_load_entry(prog + 2), d3 = walk4_load_entry(prog + 3);
    Ops X, Y;
    int er = 0;                                      // stored exponent of the entry about to run (SCALE_READ)
    unsigned t1 = 0, t2 = 0;                         // tip states of the children of the entry AFTER the next (this lane's pattern)
    int cum_e[MBAMD_WG_MAXLISTS] = {0, 0, 0, 0};
    auto tips = [&](const Walk4Entry& e, unsigned& a, unsigned& b) {
        a = as_global(T0 + ((e.ctl & MBAMD_W4_TIP1) ? e.c1 : 0u))[col];
        b = as_global(T0 + ((e.ctl & MBAMD_W4_TIP2) ? e.c2 : 0u))[col];
    };
    {   // prologue: the operands of entries 0 and 1 (waits in the open, once), entry 2's tip states
        unsigned a, b;
        tips(d0, a, b);
        er = as_global(E0 + d0.eread)[col];
        fetch(d0, a, b, X);
        tips(d1, a, b);
        fetch(d1, a, b, Y);
        tips(d2, t1, t2);
    }

    // One iteration = entry j = d0 on the operands in `cur`; once its factors are multiplied the set takes the operands of entry
    // j + 2.  Vector-memory sequence, the same on every path:
    //     3 one-byte loads (exponents of entry j + 1, tip states of entry j + 3) | [rows of a child in HBM] |
    //     2 NA operand loads of entry j + 2 | TVO + 1 stores
    // vmcnt counts loads and stores in one order: a load completes, for the wave, no earlier than the stores issued before it.
    // Requested HERE -- in front of this entry's stores, one and a half entries ahead of their use -- the operands of entry j + 2
    // queue behind the stores of entry j - 1, which have had a whole entry to land, and behind nothing younger (at the start of
    // an entry they queued behind the stores just issued: ~600 cycles of wait per entry, profiles/r05_walkg2_stamps.txt).
    auto step = [&](Ops& cur, int j) {
        const Walk4Entry ce = d0;
        const unsigned ctl = ce.ctl;
        if (ctl & MBAMD_W4_BARRIER) {
            // values other bins produced in the previous phase are read from here on: drain this wave's stores, meet
            MBD_DRAIN_ALL();
            MBD_WG_BARRIER();
            MBD_COMPILER_FENCE();
        }
        const bool run = !(ctl & MBAMD_W4_NOP);
        const unsigned mode = (ctl >> 8) & 3u;
        const int er_next = as_global(E0 + d1.eread)[col];
        unsigned n1, n2;
        tips(d3, n1, n2);
        vec bm[TV];
        const bool mem1 = ctl & MBAMD_WG_MEM1, mem2 = ctl & MBAMD_WG_MEM2;
        if (run && (mem1 || mem2)) fetch_mem(mem2 ? ce.c2 : ce.c1, bm);      // (two such children: the second one below, in the open)
        const Walk4Entry d4 = walk4_load_entry(prog + j + 4);
        if constexpr (SPLIT) {
            // a child that the previous operation produced: the partner's rows of it must be in the slot (step 2 n + 2; older
            // results were there before the partner's step 2 n' + 1 of any later operation, which the exchange below waited for);
            // after a drained operation: the partner's stores must have landed before the NEXT entry's rows are requested
            if ((run && (ctl & (MBAMD_WG_PREV1 | MBAMD_WG_PREV2))) || drained) pair_wait();
            drained = false;
        }
        acc_t f1, f2;
        float out[TPO];
        float mx = 0.0f;
        if (run) {
            const bool tip1 = ctl & MBAMD_W4_TIP1, tip2 = ctl & MBAMD_W4_TIP2;
            // the factor of a compact tip is the gathered rows where they landed; one code path per combination, no copies
            auto product = [&](auto g1, auto g2) {
#pragma unroll
                for (int t = 0; t < TPO; ++t) {
                    out[t] = (SPLIT || t < T) ? g1(t) * g2(t) : 0.0f;
                    mx = fmaxf(mx, out[t]);
                }
            };
            auto tipf1 = [&](int t) { return Va::get(cur.a1[t / VA], t % VA); };
            auto tipf2 = [&](int t) { return Va::get(cur.a2[t / VA], t % VA); };
            auto intf1 = [&](int t) { return f1[t]; };
            auto intf2 = [&](int t) { return f2[t]; };
            if (tip1 && tip2) product(tipf1, tipf2);
            else if (tip2) {
                factor(mem1, ce.c1, cur.a1, bm, f1);
                product(intf1, tipf2);
            } else if (tip1) {
                factor(mem2, ce.c2, cur.a2, bm, f2);
                product(tipf1, intf2);
            } else {
                if (mem1 && mem2) {                  // (rare: both children in HBM -- `bm` holds child 2's rows)
                    factor(true, ce.c2, cur.a2, bm, f2);
                    fetch_mem(ce.c1, bm);
                    factor(true, ce.c1, cur.a1, bm, f1);
                } else {
                    factor(mem1, ce.c1, cur.a1, bm, f1);
                    factor(mem2, ce.c2, cur.a2, bm, f2);
                }
                product(intf1, intf2);
            }
        } else {
#pragma unroll
            for (int t = 0; t < TPO; ++t) out[t] = 0.0f;
        }
        fetch(d2, t1, t2, cur);                      // (the set is free: entry j + 2's operands)
        mx = mbd_max_lane_xor32(mx);                 // the other states of this pattern sit 32 lanes apart
        if constexpr (SPLIT) {
            if (run) {                               // the column maximum over all states = the larger of the two waves' maxima
                float* const xm = xmax + ((sig >> 1) & 1) * 64;
                xm[hw * 32 + col] = mx;
                pair_signal();
                pair_wait();
                mx = fmaxf(mx, xm[(hw ^ 1) * 32 + col]);
            }
        }
        const int wm = mode == SCALE_WRITE ? -1 : 0, rm = mode == SCALE_READ ? -1 : 0;
        const int e = (scale_exponent(mx) & wm) | (er & rm);
        er = er_next;
        if (wm) {
            switch (MBAMD_WG_LIST(ctl)) {            // (wave-uniform: a scalar branch, one addition)
                case 0: cum_e[0] += e; break;
                case 1: cum_e[1] += e; break;
                case 2: cum_e[2] += e; break;
                default: cum_e[3] += e; break;
            }
        }
        const float sc = mbd_pow2(-e);
        vec ov[TVO];
#pragma unroll
        for (int t = 0; t < TPO; ++t) Vb::set(ov[t / V], t % V, out[t] * sc);   // (exact: |e| <= 126)
        const int mygroups = SPLIT ? hw * TVO * 64 : 0;      // row split: this wave's row groups of a block
        if (ctl & MBAMD_W4_KEEP) {
            vec* keep = reinterpret_cast<vec*>(reinterpret_cast<char*>(slots) + ((ctl >> 16) & 0xFFu) * SLOTB) + mygroups;
#pragma unroll
            for (int i = 0; i < TVO; ++i) keep[i * 64] = ov[i];
        }
        if constexpr (SPLIT) {
            if (run && !(ctl & MBAMD_WG_DRAIN)) pair_signal();
        }
        MBAMD_AS_GLOBAL vec* pd = reinterpret_cast<MBAMD_AS_GLOBAL vec*>((uintptr_t) (P0 + ce.dst)) + lane + mygroups;
#pragma unroll
        for (int i = 0; i < TVO; ++i) __builtin_nontemporal_store(ov[i], pd + i * 64);   // 64 * V * 4 contiguous bytes per instruction
        __builtin_nontemporal_store((int8_t) e, as_global(E0 + ce.ewrite) + col);       // (every lane group holds the same e: no exec-mask branch)
        if constexpr (SPLIT) {
            if (run && (ctl & MBAMD_WG_DRAIN)) {     // (rare: a result this bin re-reads from HBM in this phase -- evicted from its slots)
                MBD_DRAIN_VMEM();
                pair_signal();
                drained = true;
            }
        }
        d0 = d1; d1 = d2; d2 = d3; d3 = d4;
        t1 = n1; t2 = n2;
    };
    for (int j = 0; j < n; j += 2) {
        step(X, j);
        step(Y, j + 1);
    }
    // cumulative exponents of this workgroup's 32 columns: the bins' sums meet in LDS, bin 0 owns the memory update
    int* const stage = reinterpret_cast<int*>(mine);
#pragma unroll
    for (int q = 0; q < MBAMD_WG_MAXLISTS; ++q) {
        if (A.cum[q] == nullptr || (A.lists > 1 && q != (int) list)) continue;      // (separate lists: a workgroup holds one list)
        int sum = cum_e[q];
        if (W > 1) {
            if (q > 0) MBAMD_SYNC();
            if (hw == 0) stage[lane] = sum;
            MBAMD_SYNC();
            if (wave == 0)
                for (int w = 1; w < W; ++w)
                    sum += reinterpret_cast<const int*>(reinterpret_cast<const char*>(lds_walkg) + (size_t) w * (STAGE + (size_t) A.nslots * SLOTB))[lane];
        }
        if (wave == 0 && half == 0 && hw == 0) {
            int32_t* d = A.cum[q] + (size_t) k * A.Ppad + (size_t) tile * TW + col;
            if (A.cumFresh >> q & 1) *d = sum;
            else if (sum != 0) *d += sum;
        }
    }
}
}  // namespace mbamd
#endif
