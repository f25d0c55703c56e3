// mbamd_matrices_mfma.h -- transition matrices of the general-state paths on the fp64 matrix cores (included by mbamd_kernels.h's
// users after mbamd_walkg.h: it fills the tree walk's tables).  Written against mbd_mfma_f64_16x16x4 (<mbamd_dev_base.h>): the
// product and the TEST-ONLY host emulation compile this same kernel.
#ifndef MBAMD_MATRICES_MFMA_H_
#define MBAMD_MATRICES_MFMA_H_
namespace mbamd {

// ---------------------------------------------------------------------------------------------
// Transition matrices for larger state counts (TiProbs_Gen / TiProbs_GenCov, reference
// src/likelihood.c:9424-9700): P_k = U diag(exp(lambda t r_k)) U^-1 in fp64, clamped at 0, stored as
// fp32 transposed + in MFMA A-operand order.  The S x S x S contraction runs on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64): one wave per 16 rows of P,
//     A (16 x 4)  = U[i0 + (lane&15)][4 st + (lane>>4)] * exp(lambda_s t r_k)
//     B (4 x 16)  = U^-1[4 st + (lane>>4)][16 jt + (lane&15)]
//     D (16 x 16) : lane holds column 16 jt + (lane&15), rows i0 + (lane>>4) + 4 reg
// with all operands of a chunk of 8 contraction steps loaded before its MFMAs.  The eigen-system (2 S^2
// doubles, <= 64 KiB) is L2 resident.  grid = count * K workgroups of ceil(S/16) waves; NJ = ceil(S/16).
// ---------------------------------------------------------------------------------------------

template <int NJ>
__global__ void __launch_bounds__(64 * NJ, 3)     // (three workgroups per CU: at 61 states 141 + 32 registers allowed two, and the 594 workgroups of codon 100 x 5 000 ran in two rounds)
k_transition_matrices_mfma(const MatrixJob* __restrict__ jobs, RatesArg rates, int S, int SP, int K, int packedT, size_t wgTab)
{
    __shared__ double ev[64];
    const int b = blockIdx.x / K, k = blockIdx.x % K;
    const MatrixJob job = jobs[b];
    const MBAMD_AS_GLOBAL double* __restrict__ U = as_global(job.eig);
    const MBAMD_AS_GLOBAL double* __restrict__ Ui = U + (size_t) S * S;
    const MBAMD_AS_GLOBAL double* __restrict__ lam = U + (size_t) 2 * S * S;
    if ((int) threadIdx.x < S) ev[threadIdx.x] = exp(lam[threadIdx.x] * job.length * rates.r[k]);
    MBAMD_SYNC();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, ls = lane >> 4;
    const int i = 16 * wave + li;                    // A row of this lane
    const int ic = min(i, S - 1);                    // (out-of-range operands: load a valid address, feed zero)
    f64x4 acc[NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) acc[jt] = (f64x4) (0.0);
    constexpr int CH = 8;
    const int steps = (S + 3) / 4;
    for (int st0 = 0; st0 < steps; st0 += CH) {
        double a[CH], bb[NJ][CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int s = 4 * (st0 + u) + ls;
            const int sc = min(s, S - 1);
            a[u] = U[(size_t) ic * S + sc];
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) bb[jt][u] = Ui[(size_t) sc * S + min(16 * jt + li, S - 1)];
        }
        MBD_SCHED_BARRIER();                        // (all operand loads of the chunk in front of its MFMAs)
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int s = 4 * (st0 + u) + ls;
            const double av = (s < S && i < S) ? a[u] * ev[min(s, S - 1)] : 0.0;      // zero A kills the padded terms
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) acc[jt] = mbd_mfma_f64_16x16x4(av, bb[jt][u], acc[jt]);
        }
    }
    MBAMD_AS_GLOBAL float* __restrict__ out = as_global(job.out) + (size_t) k * SP * SP;
    MBAMD_AS_GLOBAL float* __restrict__ packed = as_global(job.out) + (size_t) K * SP * SP;
    const int NT = (S + 31) / 32;
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wave + ls + 4 * r, j = 16 * jt + li;
            if (row < S && j < S) {
                const double sum = acc[jt][r];
                const float v = (sum < 0.0) ? 0.0f : (float) sum;
                out[(size_t) j * SP + row] = v;
                if (packedT > 0) packed[((size_t) (k * NT + row / 32) * packedT + j / 2) * 64 + (row % 32) + 32 * (j % 2)] = v;
                if (wgTab > 0) wg_table_put(job.out + wgTab + (size_t) k * wg_table_floats(S), S, row, j, v);   // tree-walk tables (mbamd_walkg.h)
            }
        }
}

}  // namespace mbamd
#endif
