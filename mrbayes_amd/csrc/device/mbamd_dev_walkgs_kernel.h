// mbamd_dev_walkgs_kernel.h (gfx950) -- k_walkg_s: the general-state tree walk on the matrix cores with the transition tables
// staged in LDS and shared by the G waves of a workgroup (layouts, arguments and the why: mbamd_walkg.h).  The TEST-ONLY host
// emulation has a plain-loop twin under the same name in front on its include path (tests/hostemu/).
//
// Replaces CondLikeDown_Gen[_SSE] / CondLikeDown_NY98[_SSE] + CondLikeScaler_Gen / _NY98 (reference src/likelihood.c:204-588,
// 1575-1900, 4939-5070, 5413-5545) like k_walkg, with the same programs, arenas, tables and results.
//
// The vector-memory stream of a wave, in issue order (vmcnt retires in order and counts loads, LDS-DMAs and stores alike):
//     entry:   chunk 0:        [3 byte-DMAs: stored exponents + tip states of the NEXT entry -> the wave's landing area]
//              every chunk:    [NDMA LDS-DMAs: this wave's share of the table chunk D chunks ahead -> ring]
//                              [the B rows of the next chunk if its child lives in HBM (rare; registers)]
//              last chunk:     [TV result stores + 1 exponent store]
//     and at the end of every chunk   s_waitcnt vmcnt(N) ; s_barrier   with N = the instructions of that list issued after
//     this wave's share of the NEXT chunk's table -- a compile-time number per chunk position (the conditional loads only make
//     the wait stricter).  The DMAs are inline assembly: the compiler's own waits (for the conditional loads) cannot see them,
//     which again only makes those stricter.
#ifndef MBAMD_DEV_WALKGS_KERNEL_H_
#define MBAMD_DEV_WALKGS_KERNEL_H_
namespace mbamd {
__device__ __forceinline__ const Walk4Entry* wgs_program(const WalkGSArgsInline&)
{
    return reinterpret_cast<const Walk4Entry*>((uintptr_t) __builtin_amdgcn_kernarg_segment_ptr() + offsetof(WalkGSArgsInline, inl));
}
// 64 lanes x 4 bytes (base + lane4 each) -> 256 bytes at LDS byte address lds_dst (lane-linear); the 16-byte form is walk4_dma
__device__ __forceinline__ void wgs_dma4(const char* base, unsigned lane4, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 sc0 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane4), "s"(base), "s"(lds_dst) : "memory");
}
// this wave's share of a cooperative linear copy of PIECE bytes (a multiple of 256) into LDS: 1 KiB units (dwordx4), then 256-byte
// units; unit u belongs to wave u % G; waves without a unit in the last round repeat an earlier one (same bytes: harmless) so that
// every wave issues exactly wgs_dma_count instructions
template <int PIECE, int G> struct WgsDmaPlan {
    static constexpr int N16 = PIECE / 1024, N4 = (PIECE % 1024) / 256, U = N16 + N4, PER = (U + G - 1) / G;
    static_assert(PIECE % 256 == 0 && U >= 1, "table chunks are whole 256-byte rows");
};
// In LDS the 16-byte units of row group i (RG bytes: VA table rows of 64 columns) are ROTATED by i units: the MFMA operand
// reads stay one contiguous unit per lane, and the lanes of a tip's gather -- each wants ITS column of a row that depends on
// its pattern's state -- spread over the banks instead of meeting on the few the column index alone selects (unrotated:
// 16-way conflicts, a tip chunk took as long as an interior chunk's 32 MFMAs; profiles/r04_exp_walkgs.txt).  The rotation is
// free: an LDS-DMA lands lane-linearly, its SOURCE address is per lane.
template <int RG> __device__ __forceinline__ unsigned wgs_rot(unsigned d, int sign)        // piece-relative byte d -> rotated by +-16 i inside its row group
{
    const unsigned i = d / RG;
    return i * RG + ((d % RG + (unsigned) sign * 16u * i) & (unsigned) (RG - 1));
}
template <int PIECE, int G, int RG>
__device__ __forceinline__ void wgs_dma_piece(const char* src, unsigned lds_dst, int wave, unsigned lane, bool rotate)
{
    typedef WgsDmaPlan<PIECE, G> Pl;
#pragma unroll
    for (int i = 0; i < Pl::PER; ++i) {
        int u = wave + i * G;
        if ((i + 1) * G > Pl::U) u = u < Pl::U ? u : u % Pl::U;
        if (Pl::N4 == 0 || u < Pl::N16) {
            const unsigned d = (unsigned) u * 1024u, dl = d + lane * 16u;
            walk4_dma(reinterpret_cast<const f4*>(src), rotate ? wgs_rot<RG>(dl, -1) : dl, lds_dst + d);
        } else {
            const unsigned d = (unsigned) Pl::N16 * 1024u + (unsigned) (u - Pl::N16) * 256u, dl = d + lane * 4u;
            wgs_dma4(src, rotate ? wgs_rot<RG>(dl, -1) : dl, lds_dst + d);
        }
    }
}
template <int N> __device__ __forceinline__ void wgs_wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// blockDim.x = 64 * G; grid = walkgs_grid(ntiles / G, K * lists * bins); dynamic LDS = wgs_lds_bytes(G, nslots, SC, CH, D + 1).
// ntiles is a multiple of G (the engine pads the pattern count).  A job (one child factor) is CH chunks; a chunk's table piece is
// fetched D chunks ahead into a ring of D + 1 buffers.
// (second launch bound = waves per SIMD the register budget must allow: two workgroups of four waves per CU beyond 32 states)
template <int SC, int G, int CH, int D, class ARGS = WalkGSArgs>
__global__ void __launch_bounds__(64 * G, (SC > 32 ? 2 : 4))
k_walkg_s(ARGS AA)
{
    const WalkGSArgs& AS = wgs_args(AA);
    const WalkGArgs& A = AS.a;
    typedef WgShape<SC> Sh;
    typedef typename Sh::vec vec;
    typedef typename Sh::vecA vecA;
    typedef typename Sh::Vb Vb;
    typedef typename Sh::Va Va;
    typedef typename Sh::acc acc_t;
    constexpr int TW = Sh::TW, KS = Sh::KS, ACC = Sh::ACC;
    constexpr int T = Sh::T, NT = Sh::NT, V = Sh::V, VA = Sh::VA, TP = Sh::TP, NAP = Sh::NAP, NAV = NAP / VA, TV = TP / V;
    constexpr int TPC = TP / CH, NAVC = NAV / CH, TVC = TV / CH;      // per chunk: MFMA steps, A register groups, B register groups
    constexpr int NQ = 2 * CH, NB = D + 1;
    constexpr int PIECE = NAVC * VA * 256;                            // bytes of a table chunk
    constexpr int RG = VA * 256;                                      // bytes of a row group of the table (VA rows x 64 columns)
    constexpr int NDMA = WgsDmaPlan<PIECE, G>::PER;                   // LDS-DMA instructions per wave and chunk
    constexpr int NSTORE = TV + 1;                                    // result stores + the exponent store of an entry
    static_assert(TP % CH == 0 && TPC % V == 0 && NAV % CH == 0 && (TPC * NT) % VA == 0 && D >= 1 && D <= 2 && D <= NQ, "chunk geometry");
    static_assert(VA % NT == 0, "the rows of one MFMA step lie in one register group of the table");
    static_assert(CH == 1 || NAP == TP * NT, "chunks cut the table at a row-pair boundary");
    static_assert(SC / TW + 1 == CH, "one gather table (TW states) per chunk of a job");
    static_assert((ACC < T ? ACC : T) * NT <= NAVC * VA, "a compact tip's gather rows lie in the head of its table");
    constexpr unsigned SLOTB = TP * 256u;
    const unsigned lane = threadIdx.x & 63, half = lane / TW, col = lane % TW;      // half: which of the KS states of a row
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    extern __shared__ float lds_walkg[];
    const unsigned K = (unsigned) A.K, KL = K * (unsigned) A.lists, KLB = KL * (unsigned) AS.bins;
    const unsigned xcd = blockIdx.x & 7u, pos = blockIdx.x >> 3;
    const unsigned tg = (pos / KLB) * 8u + xcd, rem = pos % KLB, k = rem % K, list = (rem / K) % (unsigned) A.lists, bin = rem / KL;
    if (tg * G >= (unsigned) A.ntiles) return;
    const unsigned rg = AS.range[list][bin];
    const int len = (int) (rg & 0xFFFFu);
    if (len == 0) return;
    const unsigned tile = tg * G + (unsigned) wave;
    char* const ldsBase = reinterpret_cast<char*>(lds_walkg);
    char* const mine = ldsBase + NB * PIECE + (size_t) wave * (MBAMD_WGS_STAGE + (size_t) A.nslots * SLOTB);
    const int* const stage = reinterpret_cast<const int*>(mine);
    vec* const slots = reinterpret_cast<vec*>(mine + MBAMD_WGS_STAGE) + lane;         // this lane's V rows of row group 0, slot 0
    const unsigned ringLds = (unsigned) (uintptr_t) (__attribute__((address_space(3))) char*) ldsBase;
    const unsigned stageLds = (unsigned) (uintptr_t) (__attribute__((address_space(3))) char*) mine;
    // wave-uniform bases; the entries hold byte offsets from them
    char* const P0 = reinterpret_cast<char*>(A.partials) + (size_t) tile * A.tileBytes + (size_t) k * SLOTB;
    const uint8_t* const T0 = A.tips + (size_t) tile * A.tipTileBytes;
    int8_t* const E0 = A.exps + (size_t) ((tile * TW) >> 6) * A.estride + (size_t) k * 64 + ((tile * TW) & 63u);
    const char* const Mk = reinterpret_cast<const char*>(A.matrices) + A.tabOff + (size_t) k * A.tabBytes;
    const Walk4Entry* prog = wgs_program(AA) + ((size_t) list * AS.progW + bin) * A.entries + (rg >> 16);

    Walk4Entry cur = walk4_load_entry(prog), n1 = walk4_load_entry(prog + 1);
    int cum_e[MBAMD_WG_MAXLISTS] = {0, 0, 0, 0};
    // B rows of a chunk whose child lives in HBM (an earlier launch's result, an evicted value): loaded one chunk ahead into the
    // register set of the chunk's parity (an entry has an even number of chunks).  The MFMA chain exists twice, once per operand
    // source: with one chain behind a register copy the compiler hoists the copy -- and the vmcnt(0) it needs -- onto the LDS path
    vec bm0[TVC], bm1[TVC];
#pragma unroll
    for (int i = 0; i < TVC; ++i) bm0[i] = bm1[i] = Vb::splat(0.0f);

    // byte offset of this lane's unit of row group i inside a staged piece (rotated, see wgs_rot)
    unsigned aoff[NAVC];
#pragma unroll
    for (int i = 0; i < NAVC; ++i) aoff[i] = wgs_rot<RG>((unsigned) i * RG + lane * (unsigned) (VA * 4), +1);

    // stored exponents and tip states of entry e: one byte per lane -> a dword per lane in the landing area of `parity`
    auto tiny = [&](const Walk4Entry& e, unsigned parity) {
        const unsigned dst = stageLds + parity * 768u;
        walk4_dma_exps(E0 + e.eread, col, dst);
        walk4_dma_exps(reinterpret_cast<const int8_t*>(T0 + ((e.ctl & MBAMD_W4_TIP1) ? e.c1 : 0u)), col, dst + 256u);
        walk4_dma_exps(reinterpret_cast<const int8_t*>(T0 + ((e.ctl & MBAMD_W4_TIP2) ? e.c2 : 0u)), col, dst + 512u);
    };
    // this wave's share of the table piece of chunk q (child q / CH, part q % CH) of entry e -> ring buffer rs.  An interior child
    // needs part q % CH of the MFMA operand table A' (staged rotated); a compact tip the head of its GATHER table q % CH -- the one
    // for the states [TW (q % CH), TW (q % CH + 1)), whose rows are laid out so that a lane reads the factor registers of ITS
    // pattern's state as whole 16-byte units of its column (staged as it is).
    auto table = [&](const Walk4Entry& e, int q, int rs) {
        const bool tipc = e.ctl & ((q / CH) ? MBAMD_W4_TIP2 : MBAMD_W4_TIP1);
        const char* src = Mk + ((q / CH) ? e.m2 : e.m1) + (tipc ? (size_t) (1 + q % CH) * (NAP * 256) : (size_t) (q % CH) * PIECE);
        wgs_dma_piece<PIECE, G, RG>(src, ringLds + (unsigned) rs * PIECE, wave, lane, !tipc);
    };
    // B rows of chunk q of entry e, if that child lives in HBM
    auto memrows = [&](const Walk4Entry& e, int q, vec (&bm)[TVC]) {
        if (e.ctl & ((q / CH) ? MBAMD_WG_MEM2 : MBAMD_WG_MEM1)) {
            const MBAMD_AS_GLOBAL vec* pb = reinterpret_cast<const MBAMD_AS_GLOBAL vec*>((uintptr_t) (P0 + ((q / CH) ? e.c2 : e.c1))) + lane + (q % CH) * TVC * 64;
#pragma unroll
            for (int i = 0; i < TVC; ++i) bm[i] = pb[i * 64];
        }
    };

    // ---- prologue: entry 0's bytes, the first D table chunks, chunk 0's rows; everything lands before the first barrier ----
    tiny(cur, 0u);
#pragma unroll
    for (int q = 0; q < D; ++q) table(cur, q, q);
    memrows(cur, 0, bm0);
    int rsC = 0, rsD = D % NB;                       // ring buffer of the chunk computed next / fetched next
    wgs_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    for (int j = 0; j < len; ++j) {
        const Walk4Entry ce = cur;
        const unsigned ctl = ce.ctl;
        const bool run = !(ctl & MBAMD_W4_NOP);
        const unsigned mode = (ctl >> 8) & 3u;
        const int* const st = stage + (j & 1) * 192 + lane;
        const int er = st[0];
        const unsigned s1 = (unsigned) st[64], s2 = (unsigned) st[128];
        Walk4Entry nn = n1;                          // becomes entry j + 2 in the last chunk
        acc_t f1[NT], f2[NT];
        auto chunk = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q < NQ) {
                constexpr int ch = q / CH, h = q % CH;
                const bool tip = ctl & (ch ? MBAMD_W4_TIP2 : MBAMD_W4_TIP1), mem = ctl & (ch ? MBAMD_WG_MEM2 : MBAMD_WG_MEM1);
                // rows loaded from HBM for THIS chunk: waited for here, in front of this chunk's DMAs -- the compiler's wait cannot see
                // those, and placed at the MFMA chain it would also wait for the table piece issued a moment ago
                if (run && mem) {
#pragma unroll
                    for (int i = 0; i < TVC; ++i) {
                        if constexpr (q & 1) asm volatile("" : "+v"(bm1[i]));
                        else asm volatile("" : "+v"(bm0[i]));
                    }
                }
                if constexpr (q == 0) tiny(n1, (unsigned) ((j + 1) & 1));
                {   // the table piece D chunks ahead
                    constexpr int qf = q + D;
                    if constexpr (qf < NQ) table(ce, qf, rsD);
                    else table(n1, qf - NQ, rsD);
                    rsD = rsD + 1 == NB ? 0 : rsD + 1;
                }
                acc_t (&f)[NT] = *(ch ? &f2 : &f1);
                const char* const piece = ldsBase + rsC * PIECE;
                // the rows of the next chunk (into the other register set)
                if constexpr (q + 1 < NQ) memrows(ce, q + 1, (q & 1) ? bm0 : bm1);
                else memrows(n1, 0, (q & 1) ? bm0 : bm1);
                auto chain = [&](const vec (&b)[TVC]) {
                    if constexpr (h == 0) {
#pragma unroll
                        for (int it = 0; it < NT; ++it)
#pragma unroll
                            for (int r = 0; r < ACC; ++r) f[it][r] = 0.0f;
                    }
                    vecA a[NAVC];
#pragma unroll
                    for (int i = 0; i < NAVC; ++i) a[i] = *reinterpret_cast<const vecA*>(piece + aoff[i]);
                    // entry j + 2's descriptor: a scalar load shares lgkmcnt with LDS and returns out of order -- issued when this
                    // chunk's operands are in registers, covered by its MFMA chain.  (Pinning the reads in front of the chain in
                    // EVERY chunk costs 22 registers -- 260, one wave per SIMD -- and buys nothing: 2 760 against 2 640 cycles.)
                    if constexpr (q == NQ - 1) {
#pragma unroll
                        for (int i = 0; i < NAVC; ++i) asm volatile("" :: "v"(a[i]) : "memory");
#pragma unroll
                        for (int i = 0; i < TVC; ++i) asm volatile("" :: "v"(b[i]) : "memory");
                        nn = walk4_load_entry(prog + j + 2);
                    }
#pragma unroll
                    for (int tc = 0; tc < TPC; ++tc)
                        if (h * TPC + tc < T) {
#pragma unroll
                            for (int it = 0; it < NT; ++it) {
                                const float av = Va::get(a[(tc * NT + it) / VA], (tc * NT + it) % VA), bv = Vb::get(b[tc / V], tc % V);
#if MBAMD_WG_TW == 32
                                f[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, f[it], 0, 0, 0);
#else
                                f[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, f[it], 0, 0, 0);
#endif
                            }
                        }
                };
                if (run && tip) {
                    // the factor of a compact tip: register (it, r) = P(state of block row ACC it + r in this lane -> s), i.e. element
                    // r NT + it of the column KS (s % TW) + half of gather table s / TW -- VA consecutive elements per 16-byte unit
                    const unsigned s = ch ? s2 : s1;
                    if constexpr (h == 0) {
#pragma unroll
                        for (int it = 0; it < NT; ++it)
#pragma unroll
                            for (int r = 0; r < ACC; ++r) f[it][r] = 0.0f;
                    }
                    if (s / TW == (unsigned) h) {
                        const vecA* gp = reinterpret_cast<const vecA*>(piece + (KS * (s % TW) + half) * (unsigned) (VA * 4));
                        constexpr int NR = (ACC < T ? ACC : T) * NT;              // elements a lane needs
#pragma unroll
                        for (int g = 0; g < (NR + VA - 1) / VA; ++g) {
                            const vecA v = gp[g * 64];
#pragma unroll
                            for (int el = 0; el < VA; ++el) {
                                const int n = g * VA + el, r = n / NT, it = n % NT;
                                if (n < NR && ACC * it + r < T) f[it][r] = Va::get(v, el);
                            }
                        }
                    }
                    if constexpr (q == NQ - 1) nn = walk4_load_entry(prog + j + 2);
                } else if (run && mem) {
                    chain((q & 1) ? bm1 : bm0);
                } else if (run) {
                    const vec* sl = reinterpret_cast<const vec*>(reinterpret_cast<const char*>(slots) + (ch ? ce.c2 : ce.c1)) + h * TVC * 64;
                    vec b[TVC];
#pragma unroll
                    for (int i = 0; i < TVC; ++i) b[i] = sl[i * 64];
                    chain(b);
                } else {
                    if constexpr (q == NQ - 1) nn = walk4_load_entry(prog + j + 2);
                }
                rsC = rsC + 1 == NB ? 0 : rsC + 1;
                if constexpr (q == NQ - 1) {
                    // ---- the entry's result: product, rescale by its own power of two, LDS slot and HBM ---------------------------
                    float out[TP];
                    float mx = 0.0f;
#pragma unroll
                    for (int t = 0; t < TP; ++t) {
                        out[t] = (run && t < T) ? f1[t / ACC][t % ACC] * f2[t / ACC][t % ACC] : 0.0f;
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
                    const unsigned lq = MBAMD_WG_LIST(ctl);
#pragma unroll
                    for (int qq = 0; qq < MBAMD_WG_MAXLISTS; ++qq) cum_e[qq] += (lq == (unsigned) qq) ? (e & wm) : 0;
                    const float sc = mbd_pow2(-e);
                    vec ov[TV];
#pragma unroll
                    for (int t = 0; t < TP; ++t) Vb::set(ov[t / V], t % V, out[t] * sc);   // (exact: |e| <= 126; 2^0 needs no branch)
                    if (ctl & MBAMD_W4_KEEP) {
                        vec* keep = reinterpret_cast<vec*>(reinterpret_cast<char*>(slots) + ((ctl >> 16) & 0xFFu) * SLOTB);
#pragma unroll
                        for (int i = 0; i < TV; ++i) keep[i * 64] = ov[i];
                    }
                    MBAMD_AS_GLOBAL vec* pd = reinterpret_cast<MBAMD_AS_GLOBAL vec*>((uintptr_t) (P0 + ce.dst)) + lane;
#pragma unroll
                    for (int i = 0; i < TV; ++i) __builtin_nontemporal_store(ov[i], pd + i * 64);   // 64 * V * 4 contiguous bytes per instruction
                    __builtin_nontemporal_store((int8_t) e, as_global(E0 + ce.ewrite) + col);       // (every lane group holds the same e: no exec-mask branch)
                }
                // ---- the next chunk's table piece has landed (this wave's share; the barrier adds the others'), and nobody reads
                //      the buffer this chunk used any more
                if constexpr (D == 1) wgs_wait_vm<(q == NQ - 1 ? NSTORE : 0)>();
                else wgs_wait_vm<NDMA + (q == 0 ? NSTORE + 3 : 0) + (q == NQ - 1 ? NSTORE : 0)>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        };
        chunk(WgInt<0>{}); chunk(WgInt<1>{}); chunk(WgInt<2>{}); chunk(WgInt<3>{});
        cur = n1;
        n1 = nn;
    }
    wgs_wait_vm<0>();                                // (table pieces fetched beyond the end are still landing in this workgroup's LDS)
    // cumulative exponents of this wave's TW columns
    if (half == 0) {
#pragma unroll
        for (int q = 0; q < MBAMD_WG_MAXLISTS; ++q) {
            if (A.cum[q] == nullptr || (A.lists > 1 && q != (int) list)) continue;      // (separate lists: a workgroup holds one list)
            int32_t* d = A.cum[q] + (size_t) k * A.Ppad + (size_t) tile * TW + col;
            const int sum = cum_e[q];
            if (AS.atomicCum) { if (sum != 0) atomicAdd(d, sum); }
            else if (A.cumFresh >> q & 1) *d = sum;
            else if (sum != 0) *d += sum;
        }
    }
}
}  // namespace mbamd
#endif
