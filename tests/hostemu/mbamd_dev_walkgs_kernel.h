// mbamd_dev_walkgs_kernel.h -- TEST ONLY (tests/hostemu): plain-loop twin of k_walkg_s (the general-state tree walk with the
// transition tables staged in LDS), compiled into the host-emulation build instead of the gfx950 code of
// mrbayes_amd/csrc/device/mbamd_dev_walkgs_kernel.h.  It reads the same arguments, programs (bins as workgroups, phases as
// launches, the per-launch entry ranges), arenas and LDS slot schedule, so the engine's host logic is exercised on the CPU; the
// arithmetic order differs from the MFMA kernel and is compared with a tolerance.  Never part of the product.
#ifndef MBAMD_DEV_WALKGS_KERNEL_H_
#define MBAMD_DEV_WALKGS_KERNEL_H_
namespace mbamd {
inline const Walk4Entry* wgs_program(const WalkGSArgsInline& a) { return a.inl; }
template <int SC, int G, int CH, int D, class ARGS = WalkGSArgs>
__global__ void k_walkg_s(ARGS AA)
{
    const WalkGSArgs& AS = wgs_args(AA);
    const WalkGArgs& A = AS.a;
    const unsigned lane = threadIdx.x & 63;
    const int wave = (int) (threadIdx.x >> 6);
    if (lane != 0) return;
    const int S = A.S, SP = A.SP, TP = wg_pairs_padded(S);
    const unsigned SLOTB = wg_block_bytes(S);
    constexpr int TW = MBAMD_WG_TW;
    const unsigned K = (unsigned) A.K, KL = K * (unsigned) A.lists, KLB = KL * (unsigned) AS.bins;
    const unsigned xcd = blockIdx.x & 7u, pos = blockIdx.x >> 3;
    const unsigned tg = (pos / KLB) * 8u + xcd, rem = pos % KLB, k = rem % K, list = (rem / K) % (unsigned) A.lists, bin = rem / KL;
    if (tg * G >= (unsigned) A.ntiles) return;
    const unsigned rg = AS.range[list][bin];
    const int len = (int) (rg & 0xFFFFu);
    if (len == 0) return;
    const unsigned tile = tg * G + (unsigned) wave;
    char* lds = reinterpret_cast<char*>(mbamd_emu_dyn_lds());
    char* const mine = lds + (size_t) (D + 1) * wgs_chunk_bytes(S, CH) + (size_t) wave * (MBAMD_WGS_STAGE + (size_t) A.nslots * SLOTB);
    float* const slots = reinterpret_cast<float*>(mine + MBAMD_WGS_STAGE);
    char* const P0 = reinterpret_cast<char*>(A.partials) + (size_t) tile * A.tileBytes + (size_t) k * SLOTB;
    const uint8_t* const T0 = A.tips + (size_t) tile * A.tipTileBytes;
    int8_t* const E0 = A.exps + (size_t) ((tile * TW) >> 6) * A.estride + (size_t) k * 64 + ((tile * TW) & 63u);
    const Walk4Entry* prog = wgs_program(AA) + ((size_t) list * AS.progW + bin) * A.entries + (rg >> 16);
    int cum_e[MBAMD_WG_MAXLISTS][TW];
    for (auto& row : cum_e) for (int& v : row) v = 0;
    for (int j = 0; j < len; ++j) {
        const Walk4Entry e = prog[j];
        if (e.ctl & MBAMD_W4_NOP) continue;
        const unsigned mode = (e.ctl >> 8) & 3u;
        float* dst = reinterpret_cast<float*>(P0 + e.dst);
        float res[64][TW];
        for (int c = 0; c < TW; ++c) {
            float f[2][64];
            for (int ch = 0; ch < 2; ++ch) {
                const bool tip = e.ctl & (ch ? MBAMD_W4_TIP2 : MBAMD_W4_TIP1), mem = e.ctl & (ch ? MBAMD_WG_MEM2 : MBAMD_WG_MEM1);
                const unsigned coff = ch ? e.c2 : e.c1;
                const float* mT = reinterpret_cast<const float*>(reinterpret_cast<const char*>(A.matrices) + (ch ? e.m2 : e.m1)) + (size_t) k * SP * SP;
                if (tip) {
                    const unsigned s = T0[coff + c];
                    for (int i = 0; i < S; ++i) f[ch][i] = s >= (unsigned) S ? 1.0f : mT[(size_t) s * SP + i];
                } else {
                    const float* cl = mem ? reinterpret_cast<const float*>(P0 + coff) : slots + coff / 4;
                    for (int i = 0; i < S; ++i) {
                        float acc = 0.0f;
                        for (int jj = 0; jj < S; ++jj) acc = fmaf(mT[(size_t) jj * SP + i], cl[wg_elem(S, jj, c)], acc);
                        f[ch][i] = acc;
                    }
                }
            }
            float mx = 0.0f;
            for (int i = 0; i < S; ++i) { res[i][c] = f[0][i] * f[1][i]; mx = fmaxf(mx, res[i][c]); }
            int ex = 0;
            if (mode == SCALE_WRITE) { ex = scale_exponent(mx); cum_e[MBAMD_WG_LIST(e.ctl)][c] += ex; }
            else if (mode == SCALE_READ) ex = E0[e.eread + c];
            for (int i = 0; i < S; ++i) res[i][c] = scale_pow2(res[i][c], -ex);
            E0[e.ewrite + c] = (int8_t) ex;
        }
        for (int i = 0; i < MBAMD_WG_KS * TP; ++i)
            for (int c = 0; c < TW; ++c) {
                const float v = i < S ? res[i][c] : 0.0f;
                dst[wg_elem(S, i, c)] = v;
                if (e.ctl & MBAMD_W4_KEEP) slots[((e.ctl >> 16) & 0xFFu) * (SLOTB / 4) + wg_elem(S, i, c)] = v;
            }
    }
    for (int q = 0; q < MBAMD_WG_MAXLISTS; ++q) {
        if (A.cum[q] == nullptr || (A.lists > 1 && q != (int) list)) continue;
        for (int c = 0; c < TW; ++c) {
            int32_t* d = A.cum[q] + (size_t) k * A.Ppad + (size_t) tile * TW + c;
            if (AS.atomicCum) *d += cum_e[q][c];
            else if (A.cumFresh >> q & 1) *d = cum_e[q][c];
            else *d += cum_e[q][c];
        }
    }
}
}  // namespace mbamd
#endif
