// mbamd_reports.h -- the final ("up") pass and the scaled read-out of conditional likelihoods (included by mbamd_engine.cpp).
//
// what MrBayes computes when a run reports ancestral states, site rates, positively selected sites or
// site omegas -- CondLikeUp_Bin / _Gen / _NUC4 (reference src/likelihood.c:4574-4795) and the read-outs PrintAncStates_*,
// PrintSiteRates_Gen, PosSelProbs, SiteOmegas (src/mcmc.c:10108-11070, 12212).  MrBayes switches BEAGLE off for such
// divisions (src/mcmc.c:5760-5765) because the BEAGLE API cannot serve them; this engine can: include/libhmsbeagle/mbamd_reports.h.
//
// These run once per SAMPLED generation (every `samplefreq`-th, 500 by default), over the root-ward paths of the reported
// nodes: a thread owns one (pattern, category) column and walks the states in registers -- no tuning beyond coalesced
// access; the three partials layouts of the engine are served by one templated accessor.
//
//   final(top)  = down(top) o (P_top . tip_root)          unrooted trees: the 3-way product CondLikeRoot_* computes natively
//               = down(top)                               rooted trees
//   final(p)[a] = (sum_i up[i] P_p[a][i]) down(p)[a],   up[a] = final(anc)[a] / (sum_i P_p[a][i] down(p)[i])   (0 where that sum is 0)
// exactly as the reference writes it (including its P[a][i] in the second sum).  Every value of category k keeps the factor
// 2^-E_k(c) of the whole tree's cumulative exponent -- down(p) carries the exponents of p's subtree, the quotient those of
// the rest -- so the pass needs no exponent arithmetic of its own; the read-out brings the categories of a pattern to their
// common largest exponent (exact: powers of two) and reports it as the natural-log site scaler the reference's read-outs expect.
// One exception, since round 4: under MrBayes' DYNAMIC rescaling scheme (src/mbbeagle.c: no rescaling until a likelihood
// underflows) the down-pass values of a 60-taxon tree are legitimately 1e-30 ... 1e-44 floats -- the log-likelihood is fine, but
// the final pass multiplies and divides them (quotients beyond 1e38, products below 1e-45) and the reference's read-outs then
// divide 0 by 0 (found by the 500 x 20 000 golden, tests/test_fullsize_dropin.py).  So the arithmetic runs in DOUBLE, and the top
// node's column is brought to [0.5, 1) by its own power of two, recorded per (pattern, category) next to the buffer (`fexp`);
// everything below inherits that scale (the pass is linear in the ancestor's values) and the read-out adds it to the exponents.
#ifndef MBAMD_REPORTS_H_
#define MBAMD_REPORTS_H_

namespace mbamd {

// LAYOUT as in k_import_partials: 0 general tile-major buffer, 1 4-state arena, 2 20/61-state tree-walk arena
template <int LAYOUT>
__device__ __forceinline__ size_t rep_index(int S, int K, size_t pstride, int k, int i, int c)
{
    if (LAYOUT == 1) return (blk_index(c, pstride) + (size_t) k * 64) * 4 + (size_t) i;
    if (LAYOUT == 2) return wg_index(S, pstride, k, i, c);
    return gen_index(K, S, k, i, c);
}
// compatibility of state j with the compact tip's observation at pattern c (1 / 0)
template <int LAYOUT>
__device__ __forceinline__ float rep_tip(const void* tip, int S, size_t tstride, int c, int j)
{
    if (LAYOUT == 1) {
        const uint64_t* planes = reinterpret_cast<const uint64_t*>(tip) + (size_t) (c >> 6) * tstride;
        return (float) (planes[j] >> (c & 63) & 1u);
    }
    const unsigned s = LAYOUT == 2 ? reinterpret_cast<const uint8_t*>(tip)[(size_t) (c / MBAMD_WG_TW) * tstride + (c % MBAMD_WG_TW)]
                                   : reinterpret_cast<const uint8_t*>(tip)[c];
    return (s >= (unsigned) S || s == (unsigned) j) ? 1.0f : 0.0f;
}

struct FinalOp {
    float* dst;
    const float* anc;          // final partials of the ancestor, or nullptr (top node)
    const float* down;         // down-pass partials of this node
    const float* matrix;       // transposed [K][SP][SP]: this node's branch
    const void* tip;           // top node of an unrooted tree: the root tip (compact states or partials), else nullptr
    int tipKind;               // CHILD_STATES / CHILD_PARTIALS
    int Ppad;
    int32_t* fexp;             // top node: the column's own exponent is stored here, int32 [K][Ppad]
};

#define MBAMD_REP_MAXS 64
template <int LAYOUT>
__global__ void __launch_bounds__(64)
k_final_pass(FinalOp op, int S, int SP, int K, int P, size_t pstride, size_t tstride)
{
    // a column's down-pass values (floats) and the quotients / products of the pass (doubles) live in LDS, [state][thread]: a
    // thread walks its own column with a run-time state index -- as private arrays these were 1 040 bytes of scratch per lane
    __shared__ float d_all[MBAMD_REP_MAXS][64];
    __shared__ double u_all[MBAMD_REP_MAXS][64];
    const int c = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
    if (c >= P) return;
    float (*d)[64] = reinterpret_cast<float (*)[64]>(&d_all[0][threadIdx.x]);
    double (*u)[64] = reinterpret_cast<double (*)[64]>(&u_all[0][threadIdx.x]);
#define D_(i) ((double) d[i][0])
#define U_(i) (u[i][0])
    for (int i = 0; i < S; ++i) d[i][0] = op.down[rep_index<LAYOUT>(S, K, pstride, k, i, c)];
    const float* m = op.matrix;
    if (op.anc == nullptr) {
        double mx = 0.0;
        for (int a = 0; a < S; ++a) {
            double f = 1.0;
            if (op.tip != nullptr) {
                f = 0.0;
                for (int j = 0; j < S; ++j) {
                    const float t = op.tipKind == CHILD_STATES ? rep_tip<LAYOUT>(op.tip, S, tstride, c, j)
                                                              : reinterpret_cast<const float*>(op.tip)[rep_index<LAYOUT>(S, K, pstride, k, j, c)];
                    f += (double) mat_at(m, SP, k, a, j) * (double) t;
                }
            }
            U_(a) = D_(a) * f;
            mx = U_(a) > mx ? U_(a) : mx;
        }
        int e = 0;
        if (mx > 0.0 && mx < 1.0e300) (void) frexp(mx, &e);
        op.fexp[(size_t) k * op.Ppad + c] = e;
        for (int a = 0; a < S; ++a) op.dst[rep_index<LAYOUT>(S, K, pstride, k, a, c)] = (float) ldexp(U_(a), -e);
        return;
    }
    for (int a = 0; a < S; ++a) {
        double sum = 0.0;
        for (int i = 0; i < S; ++i) sum += (double) mat_at(m, SP, k, a, i) * D_(i);
        const double fa = (double) op.anc[rep_index<LAYOUT>(S, K, pstride, k, a, c)];
        U_(a) = sum != 0.0 ? fa / sum : 0.0;
    }
    for (int a = 0; a < S; ++a) {
        double sum = 0.0;
        for (int i = 0; i < S; ++i) sum += U_(i) * (double) mat_at(m, SP, k, a, i);
        op.dst[rep_index<LAYOUT>(S, K, pstride, k, a, c)] = (float) (sum * D_(a));
    }
#undef D_
#undef U_
}

// out[k][c][i] = buffer[k][c][i] 2^(E_kc - Emax_c), lnScale[c] = Emax_c ln 2.  Exponents: `wide` int32 [K][Ppad] (arena paths,
// per pattern and category), or `narrow` int32 [Ppad] (general path, per pattern), or neither (all zero); plus `extra` (final partials).
template <int LAYOUT>
__global__ void __launch_bounds__(256)
k_export_scaled(const float* __restrict__ in, const int32_t* __restrict__ wide, const int32_t* __restrict__ narrow,
                const int32_t* __restrict__ extra, int S, int K, int P, int Ppad, size_t pstride, float* __restrict__ out, float* __restrict__ lnScale)
{
    const size_t total = (size_t) K * P * S;
    const size_t g = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int i = (int) (g % S);
    const int c = (int) ((g / S) % P);
    const int k = (int) (g / ((size_t) S * P));
    // exponent of (category q, this pattern): the cumulative buffer's, plus the final pass's own (`extra`, int32 [K][Ppad])
    auto expo = [&](int q) { return (wide ? wide[(size_t) q * Ppad + c] : (narrow ? narrow[c] : 0)) + (extra ? extra[(size_t) q * Ppad + c] : 0); };
    // (a category whose column is all zero -- under dynamic rescaling the slowest categories of a variable site underflow, rightly:
    //  they carry nothing -- has no exponent to speak of: left in, its stale one can sit hundreds of binades above the others and
    //  the common scale would flush every category that does carry the site to zero)
    int emax = -2147483647;
    for (int q = 0; q < K; ++q) {
        bool any = false;
        for (int j = 0; j < S && !any; ++j) any = in[rep_index<LAYOUT>(S, K, pstride, q, j, c)] != 0.0f;
        const int v = expo(q);
        if (any && v > emax) emax = v;
    }
    if (emax == -2147483647) emax = 0;
    const int e = expo(k);
    const float v = in[rep_index<LAYOUT>(S, K, pstride, k, i, c)];
    out[g] = v == 0.0f ? 0.0f : ldexpf(v, e - emax);
    if (i == 0 && k == 0) lnScale[c] = (float) ((double) emax * 0.69314718055994530942);
}

}  // namespace mbamd
#endif
