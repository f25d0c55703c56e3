// mbamd_integrate_wg.h -- root / edge integration for the 20/61-state tree-walk layout (included by mbamd_kernels.h inside namespace
// mbamd).  Product and TEST-ONLY host emulation compile this same kernel (the emulation runs its 256 threads as fibers).
#ifndef MBAMD_INTEGRATE_WG_H_
#define MBAMD_INTEGRATE_WG_H_
#define MBAMD_INTEGRATE_WG_THREADS 256
#define MBAMD_INTEGRATE_WG_PATTERNS 32          // patterns per workgroup (one block sum each)
// The same for the tree-walk layout (mbamd_walkg.h: partials [tile][buffer][K] blocks in wg_at order, tip states
// [tile][buffer][TW], cumulative exponents per pattern AND category): eight threads per pattern, the categories recombined
// exactly as in k_integrate_lnl_s4.  grid = P_pad/32, block = 256 (32 patterns: one or two tiles).
__global__ void __launch_bounds__(256)
k_integrate_lnl_wg_wide(IntegrateArgs4 a, int S, int SP, int K, int P, int Ppad, WgGeom geo, const double* __restrict__ pattern_weights,
                        double* __restrict__ site, double* __restrict__ wsite)
{
    __shared__ double part[8][32];
    const int p = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int sh = wg_vec_shift(S);                  // rows interleaved per lane: wg_vec(S) = 1 << sh
    const int c = blockIdx.x * 32 + p;
    const bool live = c < P;
    const int tile = c / MBAMD_WG_TW, pt = c % MBAMD_WG_TW;      // this pattern's tile and its column there
    const size_t pb = (size_t) tile * geo.tileFloats, kstride = (size_t) geo.TP * 64;
    int emax = -2147483647;
    double like = 0.0;
    if (live) {
        for (int n = 0; n < a.count; ++n)
            for (int k = 0; k < K; ++k) {
                const int e = a.cum[n] ? a.cum[n][(size_t) k * Ppad + c] : 0;
                emax = e > emax ? e : emax;
            }
        for (int n = 0; n < a.count; ++n) {
            const float* __restrict__ par = reinterpret_cast<const float*>(a.parent[n]) + pb;
            const double* __restrict__ fr = a.freqs[n];
            unsigned s = 0;
            if (a.child[n] != nullptr && a.child_kind[n] == CHILD_STATES) {
                // The form every unrooted evaluation ends with: the root tip as compact states.  S <= 64 on this layout: thread g of
                // a pattern owns states g, g+8, ..., g+56.  Its loads for FOUR categories are issued together (as loops over the
                // states and the categories these were up to 8 K dependent round trips: 17 us at 157 workgroups); the sums run in
                // the order of the plain loops (the bits do not change).
                s = reinterpret_cast<const uint8_t*>(a.child[n])[(size_t) tile * geo.tipTileBytes + pt];
                double f[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) f[u] = (g + 8 * u < S) ? fr[g + 8 * u] : 0.0;
                for (int k0 = 0; k0 < K; k0 += 4) {
                    float v[4][8], pc[4][8];
                    int e[4];
                    double w[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (k0 + q >= K) break;                       // (wave-uniform: a single category loads once, not four times)
                        const int k = k0 + q;
                        const float* __restrict__ pk = par + (size_t) k * kstride;
                        const float* __restrict__ mrow = a.matrix[n] + (size_t) k * SP * SP + (size_t) (s < (unsigned) S ? s : 0) * SP;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = g + 8 * u;
                            const bool ok = i < S;
                            v[q][u] = ok ? pk[wg_elem_sh(sh, ok ? i : 0, pt)] : 0.0f;
                            pc[q][u] = (s >= (unsigned) S) ? 1.0f : (ok ? mrow[i] : 0.0f);
                        }
                        e[q] = a.cum[n] ? a.cum[n][(size_t) k * Ppad + c] : 0;
                        w[q] = a.weights[n][k];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (k0 + q >= K) break;
                        double cat = 0.0;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (g + 8 * u < S) cat += (double) (v[q][u] * pc[q][u]) * f[u];
                        like += ldexp(cat * w[q], e[q] - emax);
                    }
                }
                continue;
            }
            for (int k = 0; k < K; ++k) {
                const float* __restrict__ pk = par + (size_t) k * kstride;
                double cat = 0.0;
                if (a.child[n] == nullptr) {
                    for (int i0 = 0; i0 < S; i0 += 64) {
                        float v[8];
                        double f[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = i0 + g + 8 * u;
                            v[u] = (i < S) ? pk[wg_elem_sh(sh, i, pt)] : 0.0f;
                            f[u] = (i < S) ? fr[i] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) cat += (double) v[u] * f[u];
                    }
                } else {
                    const float* __restrict__ ch = reinterpret_cast<const float*>(a.child[n]) + pb + (size_t) k * kstride;
                    const float* __restrict__ m = a.matrix[n] + (size_t) k * SP * SP;
                    for (int i = g; i < S; i += 8) {
                        float acc = 0.0f;
                        for (int j = 0; j < S; ++j) acc = fmaf(m[(size_t) j * SP + i], ch[wg_elem_sh(sh, j, pt)], acc);
                        cat += (double) (pk[wg_elem_sh(sh, i, pt)] * acc) * fr[i];
                    }
                }
                const int e = a.cum[n] ? a.cum[n][(size_t) k * Ppad + c] : 0;
                like += ldexp(cat * a.weights[n][k], e - emax);
            }
        }
    }
    part[g][p] = like;
    MBAMD_SYNC();
    if (g != 0) return;                              // lanes 0..31 of wave 0 finish
    double wl = 0.0;
    if (live) {
        const double total = ((part[0][p] + part[1][p]) + (part[2][p] + part[3][p])) + ((part[4][p] + part[5][p]) + (part[6][p] + part[7][p]));
        const double lnl = log(total) + (double) emax * 0.69314718055994530942;
        site[c] = lnl;
        wl = lnl * pattern_weights[c];
    } else if (c < Ppad) {
        site[c] = 0.0;
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) wl += mbd_shfl_down_32(wl, off);
    if (p == 0) wsite[blockIdx.x] = wl;
}

#endif
