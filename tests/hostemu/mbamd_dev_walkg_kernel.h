// mbamd_dev_walkg_kernel.h -- TEST ONLY (tests/hostemu): plain-loop twins of the 20/61-state tree-walk kernel and its root
// integration, compiled into the host-emulation build instead of the gfx950 code of mrbayes_amd/csrc/mbamd_walkg.h /
// mbamd_kernels_mfma.h.  They read the same arguments, programs, arenas and LDS slot schedule, so the engine's host logic
// (arenas, program compiler, slots, phases, merged lists, exponents) is exercised on the CPU; the arithmetic order differs
// from the MFMA kernel and is compared with a tolerance.  Never part of the product.
#ifndef MBAMD_DEV_WALKG_KERNEL_H_
#define MBAMD_DEV_WALKG_KERNEL_H_
namespace mbamd {
inline const Walk4Entry* wg_program(const WalkGArgsInline& a) { return a.inl; }
// ---- host-emulation twin (CPU CI of the host logic: arenas, programs, slots, phases): lane 0 of every wave walks the
// program with plain loops over the tile's patterns; children come from the emulated LDS slots exactly as scheduled
template <int SC, int WMAX, int CH, int DEPTH, class ARGS = WalkGArgs>
__global__ void k_walkg(ARGS AA)
{
    const WalkGArgs& A = wg_args(AA);
    const unsigned lane = threadIdx.x & 63;
    // row split (A.pair, k_walkg2): a subtree bin is a pair of waves; here the first wave of a pair does the bin's work, the second
    // only keeps the barriers' company
    const int S = A.S, SP = A.SP, TP = wg_pairs_padded(S);
    const int wpb = wg_waves_per_bin(A.pair != 0), hw = (int) (threadIdx.x >> 6) % wpb;
    const int wave = (int) (threadIdx.x >> 6) / wpb, W = (int) (blockDim.x >> 6) / wpb;
    const unsigned STAGE = wg_stage_bytes(A.pair != 0);
    const unsigned SLOTB = wg_block_bytes(S);
    const unsigned K = (unsigned) A.K, KL = K * (unsigned) A.lists;
    const unsigned xcd = blockIdx.x & 7u, pos = blockIdx.x >> 3;
    const unsigned tile = (pos / KL) * 8u + xcd, k = (pos % KL) % K, list = (pos % KL) / K;
    if (tile >= (unsigned) A.ntiles) return;
    char* lds = reinterpret_cast<char*>(mbamd_emu_dyn_lds());
    char* const mine = lds + (size_t) wave * (STAGE + (size_t) A.nslots * SLOTB);
    float* const slots = reinterpret_cast<float*>(mine + STAGE);
    char* const P0 = reinterpret_cast<char*>(A.partials) + (size_t) tile * A.tileBytes + (size_t) k * SLOTB;
    const uint8_t* const T0 = A.tips + (size_t) tile * A.tipTileBytes;
    constexpr int TW = MBAMD_WG_TW;
    int8_t* const E0 = A.exps + (size_t) ((tile * TW) >> 6) * A.estride + (size_t) k * 64 + ((tile * TW) & 63u);
    const Walk4Entry* prog = wg_program(AA) + ((size_t) list * W + wave) * A.entries;
    int cum_e[MBAMD_WG_MAXLISTS][TW];
    for (auto& row : cum_e) for (int& v : row) v = 0;
    for (int j = 0; j < A.entries; ++j) {
        const Walk4Entry e = prog[j];
        if (e.ctl & MBAMD_W4_BARRIER) mbamd_emu_barrier();
        if (lane != 0 || hw != 0 || (e.ctl & MBAMD_W4_NOP)) continue;
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
    // cumulative exponents: the waves' sums meet in LDS, wave 0 owns the memory update
    for (int q = 0; q < MBAMD_WG_MAXLISTS; ++q) {
        if (A.cum[q] == nullptr || (A.lists > 1 && q != (int) list)) continue;
        int* stage = reinterpret_cast<int*>(mine);
        if (W > 1) {
            if (q > 0) mbamd_emu_barrier();
            if (lane == 0 && hw == 0) for (int c = 0; c < TW; ++c) stage[c] = cum_e[q][c];
            mbamd_emu_barrier();
        }
        if (wave == 0 && lane == 0 && hw == 0)
            for (int c = 0; c < TW; ++c) {
                int sum = cum_e[q][c];
                for (int w = 1; w < W; ++w) sum += reinterpret_cast<const int*>(lds + (size_t) w * (STAGE + (size_t) A.nslots * SLOTB))[c];
                int32_t* d = A.cum[q] + (size_t) k * A.Ppad + (size_t) tile * TW + c;
                if (A.cumFresh >> q & 1) *d = sum; else *d += sum;
            }
    }
}

}  // namespace mbamd
#endif
