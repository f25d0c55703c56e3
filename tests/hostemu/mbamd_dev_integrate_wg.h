// mbamd_dev_integrate_wg.h -- TEST ONLY (tests/hostemu): root / edge integration for the 20/61-state tree-walk layout with one
// thread per pattern, for the host-emulation build (the product kernel, k_integrate_lnl_wg_wide in mbamd_kernels_mfma.h, uses
// eight threads per pattern and wave shuffles).  Never part of the product.
#ifndef MBAMD_DEV_INTEGRATE_WG_H_
#define MBAMD_DEV_INTEGRATE_WG_H_
#define MBAMD_INTEGRATE_WG_KERNEL k_integrate_lnl_wg
#define MBAMD_INTEGRATE_WG_THREADS 64
#define MBAMD_INTEGRATE_WG_PATTERNS 64

// root / edge integration for the tree-walk layout, one thread per pattern (the product: k_integrate_lnl_wg_wide)
__global__ void __launch_bounds__(64)
k_integrate_lnl_wg(IntegrateArgs4 a, int S, int SP, int K, int P, int Ppad, WgGeom g,
                   const double* __restrict__ pattern_weights, double* __restrict__ site, double* __restrict__ wsite)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    const size_t pb = (size_t) (c / MBAMD_WG_TW) * g.tileFloats;               // float index of (tile, category 0) of the buffer
    const size_t kstride = (size_t) g.TP * 64;
    const int p = c % MBAMD_WG_TW;
    double wl = 0.0;
    if (c < P) {
        int emax = -2147483647;
        for (int n = 0; n < a.count; ++n)
            for (int k = 0; k < K; ++k) {
                const int e = a.cum[n] ? a.cum[n][(size_t) k * Ppad + c] : 0;
                emax = e > emax ? e : emax;
            }
        double total = 0.0;
        for (int n = 0; n < a.count; ++n) {
            const double* __restrict__ pi = a.freqs[n];
            const float* __restrict__ par = reinterpret_cast<const float*>(a.parent[n]) + pb;
            unsigned s = 0;
            if (a.child[n] != nullptr && a.child_kind[n] == CHILD_STATES)
                s = reinterpret_cast<const uint8_t*>(a.child[n])[(size_t) (c / MBAMD_WG_TW) * g.tipTileBytes + (c % MBAMD_WG_TW)];
            for (int k = 0; k < K; ++k) {
                const float* __restrict__ pk = par + (size_t) k * kstride;
                double cat = 0.0;
                if (a.child[n] == nullptr) {
                    for (int i = 0; i < S; ++i) cat += (double) pk[wg_elem(S, i, p)] * pi[i];
                } else if (a.child_kind[n] == CHILD_STATES) {
                    const float* __restrict__ mrow = a.matrix[n] + (size_t) k * SP * SP + (size_t) (s < (unsigned) S ? s : 0) * SP;
                    for (int i = 0; i < S; ++i) {
                        const float pc = (s >= (unsigned) S) ? 1.0f : mrow[i];
                        cat += (double) (pk[wg_elem(S, i, p)] * pc) * pi[i];
                    }
                } else {
                    const float* __restrict__ ch = reinterpret_cast<const float*>(a.child[n]) + pb + (size_t) k * kstride;
                    const float* __restrict__ m = a.matrix[n] + (size_t) k * SP * SP;
                    for (int i = 0; i < S; ++i) {
                        float acc = 0.0f;
                        for (int j = 0; j < S; ++j) acc = fmaf(m[(size_t) j * SP + i], ch[wg_elem(S, j, p)], acc);
                        cat += (double) (pk[wg_elem(S, i, p)] * acc) * pi[i];
                    }
                }
                const int e = a.cum[n] ? a.cum[n][(size_t) k * Ppad + c] : 0;
                total += ldexp(cat * a.weights[n][k], e - emax);
            }
        }
        const double lnl = log(total) + (double) emax * 0.69314718055994530942;
        site[c] = lnl;
        wl = lnl * pattern_weights[c];
    } else if (c < Ppad) {
        site[c] = 0.0;
    }
    mbd_wave_sum_store(wl, wsite + blockIdx.x);
}


#endif
