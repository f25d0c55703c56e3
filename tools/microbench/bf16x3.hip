// Microbenchmark (gfx950): can a 61 x 61 (or 20 x 20) fp32 matrix product run on the 16-bit matrix cores WITHOUT giving up fp32
// arithmetic?  Three things, in this order:
//   1. the operand layout of v_mfma_f32_32x32x16_bf16 (probed, not assumed), how it accumulates (is a product of two bf16 exact
//      in the fp32 accumulator?  are bf16 denormals read?), and what its chain costs;
//   2. numerics of the split  a = a1 + a2 + a3  (three bf16 pieces, exact by truncation) on transition-matrix x partials products
//      shaped like the tree walk's (64 x 64 rows summing to one, times 64 x 32 columns scaled to a maximum in [0.5, 1)): the six
//      products of order <= 4 and all nine, against an fp64 reference and against the fp32 MFMA chain the product kernel uses now;
//   3. what ONE child factor costs a wave in steady state: fp32 (62 v_mfma_f32_32x32x2_f32) against the split (the B pieces formed
//      in the VALU between the MFMAs, 48 or 72 v_mfma_f32_32x32x16_bf16), one wave per SIMD and two.
//   hipcc --offload-arch=gfx950 -O3 bf16x3.hip -o bf16x3 && ./bf16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef float v16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bf16x8 as_bf(u4 x) { return __builtin_bit_cast(bf16x8, x); }
// two fp32 -> their top halves in one dword (lo16 = a, hi16 = b): truncation to bf16
__device__ __forceinline__ unsigned pack_hi(float a, float b) { return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u); }
__device__ __forceinline__ float top(float a) { return __uint_as_float(__float_as_uint(a) & 0xFFFF0000u); }

// ---------------------------------------------------------------------------------------------------------------- 1. layout
// D = A B with A [32][16], B [16][32] given as fp32 values that are exact in bf16.  kmap(lane half h, j) = the k index element j
// of lane 32 h + m holds, for variant v:  0: 8 h + j;  1: 4 h + (j & 3) + 8 (j >> 2)
__global__ void k_layout(const float* A, const float* B, float* D, int variant)
{
    const unsigned lane = threadIdx.x, h = lane >> 5, m = lane & 31;
    unsigned pa[4], pb[4];
    for (int jj = 0; jj < 4; ++jj) {
        int k0, k1;
        const int j0 = 2 * jj, j1 = 2 * jj + 1;
        if (variant == 0) { k0 = 8 * h + j0; k1 = 8 * h + j1; }
        else { k0 = 4 * h + (j0 & 3) + 8 * (j0 >> 2); k1 = 4 * h + (j1 & 3) + 8 * (j1 >> 2); }
        pa[jj] = pack_hi(A[m * 16 + k0], A[m * 16 + k1]);
        pb[jj] = pack_hi(B[k0 * 32 + m], B[k1 * 32 + m]);
    }
    u4 ua = {pa[0], pa[1], pa[2], pa[3]}, ub = {pb[0], pb[1], pb[2], pb[3]};
    v16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ua), as_bf(ub), acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + m] = acc[r];
}

// ---------------------------------------------------------------------------------------------------------------- 2. numerics
// One wave: D[64][32] = A[64][64] B[64][32].  mode 0: fp32 MFMA chain (k ascending, as k_walkg);  1: six products;  2: nine.
// A, B plain row-major fp32 in memory; the split is done here, by truncation.
__device__ __forceinline__ void split3(float v, float& a, float& b, float& c)
{
    a = top(v);
    const float r1 = v - a;
    b = top(r1);
    c = top(r1 - b);        // (what is left after two pieces has at most 8 significant bits: exact -- unless it went denormal)
}
__global__ void k_numerics(const float* A, const float* B, float* D, int mode, int order)
{
    const unsigned lane = threadIdx.x, h = lane >> 5, m = lane & 31;
    for (int it = 0; it < 2; ++it) {
        v16 acc = {};
        if (mode == 0) {
            for (int t = 0; t < 32; ++t)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(32 * it + m) * 64 + 2 * t + h], B[(2 * t + h) * 32 + m], acc, 0, 0, 0);
        } else {
            for (int kb = 0; kb < 4; ++kb) {
                unsigned a1[4], a2[4], a3[4], b1[4], b2[4], b3[4];
                for (int jj = 0; jj < 4; ++jj) {
                    float x[2][3], y[2][3];
                    for (int e = 0; e < 2; ++e) {
                        const int k = 16 * kb + 8 * h + 2 * jj + e;
                        split3(A[(32 * it + m) * 64 + k], x[e][0], x[e][1], x[e][2]);
                        split3(B[k * 32 + m], y[e][0], y[e][1], y[e][2]);
                    }
                    a1[jj] = pack_hi(x[0][0], x[1][0]); a2[jj] = pack_hi(x[0][1], x[1][1]); a3[jj] = pack_hi(x[0][2], x[1][2]);
                    b1[jj] = pack_hi(y[0][0], y[1][0]); b2[jj] = pack_hi(y[0][1], y[1][1]); b3[jj] = pack_hi(y[0][2], y[1][2]);
                }
                const bf16x8 A1 = as_bf(u4{a1[0], a1[1], a1[2], a1[3]}), A2 = as_bf(u4{a2[0], a2[1], a2[2], a2[3]}), A3 = as_bf(u4{a3[0], a3[1], a3[2], a3[3]});
                const bf16x8 B1 = as_bf(u4{b1[0], b1[1], b1[2], b1[3]}), B2 = as_bf(u4{b2[0], b2[1], b2[2], b2[3]}), B3 = as_bf(u4{b3[0], b3[1], b3[2], b3[3]});
#define MM(X, Y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X, Y, acc, 0, 0, 0)
                if (order == 0) {            // small terms first
                    if (mode == 2) { MM(A3, B3); MM(A3, B2); MM(A2, B3); }
                    MM(A3, B1); MM(A1, B3); MM(A2, B2); MM(A2, B1); MM(A1, B2); MM(A1, B1);
                } else {                     // large terms first
                    MM(A1, B1); MM(A1, B2); MM(A2, B1); MM(A2, B2); MM(A1, B3); MM(A3, B1);
                    if (mode == 2) { MM(A2, B3); MM(A3, B2); MM(A3, B3); }
                }
            }
        }
        for (int r = 0; r < 16; ++r) D[(32 * it + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + m] = acc[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------- 2b. numerics, variants
// split: 0 truncation (pieces all of one sign: what is dropped is a BIAS), 1 round-to-nearest-even (v_cvt_pk_bf16_f32: signed pieces,
// half the magnitude);  nprod 6 | 9;  order: 0 small terms first inside each K-block, 1 inside each half of K (the walk's chunk),
// 2 over the whole of K (all a3 b1 products, then all a1 b3, ... , all a1 b1 last: only the last few MFMAs round at full magnitude)
__device__ __forceinline__ unsigned cvt_pk_rne(float a, float b)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// k_numerics3: rounded pieces, six products, K-block by K-block -- into NACC separate accumulators (K-block kb -> accumulator kb % NACC),
// added up by the VALU (round-to-nearest) at the end: does keeping the matrix core's running sum small reduce what its alignment drops?
template <int NACC> __global__ void k_numerics3(const float* A, const float* B, float* D, int nkb, int ntile)
{
    const unsigned lane = threadIdx.x, h = lane >> 5, m = lane & 31;
    const int KW = 16 * nkb;
    for (int it = 0; it < ntile; ++it) {
        v16 acc[NACC];
        for (int a = 0; a < NACC; ++a) acc[a] = (v16) (0.0f);
        for (int kb = 0; kb < nkb; ++kb) {
            u4 Ap[3], Bp[3];
            for (int jj = 0; jj < 4; ++jj) {
                const int k0 = 16 * kb + 8 * h + 2 * jj, k1 = k0 + 1;
                float x0 = A[(32 * it + m) * KW + k0], x1 = A[(32 * it + m) * KW + k1], y0 = B[k0 * 32 + m], y1 = B[k1 * 32 + m];
                for (int s = 0; s < 3; ++s) {
                    const unsigned pa = cvt_pk_rne(x0, x1), pb = cvt_pk_rne(y0, y1);
                    Ap[s][jj] = pa; Bp[s][jj] = pb;
                    x0 -= __uint_as_float(pa << 16); x1 -= __uint_as_float(pa & 0xFFFF0000u);
                    y0 -= __uint_as_float(pb << 16); y1 -= __uint_as_float(pb & 0xFFFF0000u);
                }
            }
            const int pairs[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
            for (int pr = 0; pr < 6; ++pr)
                acc[kb % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ap[pairs[pr][0]]), as_bf(Bp[pairs[pr][1]]), acc[kb % NACC], 0, 0, 0);
        }
        v16 sum = acc[0];
        if (NACC == 4) sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        else if (NACC == 2) sum = acc[0] + acc[1];
        for (int r = 0; r < 16; ++r) D[(32 * it + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + m] = sum[r];
    }
}
__global__ void k_numerics2(const float* A, const float* B, float* D, int split, int nprod, int order, int nkb, int ntile)
{
    const unsigned lane = threadIdx.x, h = lane >> 5, m = lane & 31;
    const int KW = 16 * nkb;
    for (int it = 0; it < ntile; ++it) {
        u4 Ap[3][4], Bp[3][4];
        for (int kb = 0; kb < nkb; ++kb)
            for (int jj = 0; jj < 4; ++jj) {
                const int k0 = 16 * kb + 8 * h + 2 * jj, k1 = k0 + 1;
                float x0 = A[(32 * it + m) * KW + k0], x1 = A[(32 * it + m) * KW + k1], y0 = B[k0 * 32 + m], y1 = B[k1 * 32 + m];
                for (int s = 0; s < 3; ++s) {
                    unsigned pa, pb;
                    if (split == 0) { pa = pack_hi(x0, x1); pb = pack_hi(y0, y1); }
                    else { pa = cvt_pk_rne(x0, x1); pb = cvt_pk_rne(y0, y1); }
                    Ap[s][kb][jj] = pa; Bp[s][kb][jj] = pb;
                    x0 -= __uint_as_float(pa << 16); x1 -= __uint_as_float(pa & 0xFFFF0000u);
                    y0 -= __uint_as_float(pb << 16); y1 -= __uint_as_float(pb & 0xFFFF0000u);
                }
            }
        v16 acc = {};
        // pairs (a piece, b piece), small first
        const int pairs9[9][2] = {{2, 2}, {2, 1}, {1, 2}, {2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
        const int first = nprod == 9 ? 0 : 3;
        const int span = order == 0 ? 1 : (order == 1 ? (nkb >= 2 ? nkb / 2 : 1) : nkb);
        for (int k0 = 0; k0 < nkb; k0 += span)
            for (int pr = first; pr < 9; ++pr)
                for (int kb = k0; kb < k0 + span && kb < nkb; ++kb)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ap[pairs9[pr][0]][kb]), as_bf(Bp[pairs9[pr][1]][kb]), acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) D[(32 * it + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + m] = acc[r];
    }
}
// fp32 chain at a general shape (k ascending, two states per step)
__global__ void k_numerics_f32(const float* A, const float* B, float* D, int nkb, int ntile)
{
    const unsigned lane = threadIdx.x, h = lane >> 5, m = lane & 31;
    const int KW = 16 * nkb;
    for (int it = 0; it < ntile; ++it) {
        v16 acc = {};
        for (int t = 0; t < KW / 2; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(32 * it + m) * KW + 2 * t + h], B[(2 * t + h) * 32 + m], acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) D[(32 * it + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + m] = acc[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------- 3. timing
// `n` child factors per wave.  B of factor i+1 = a cheap function of factor i's result (keeps the dependence of a tree walk).
// mode 0: 62 fp32 MFMAs (T = 31 steps x 2 output tiles);  1 / 2: the split, 4 K-blocks x 2 tiles x 6 / 9 bf16 MFMAs, the B pieces
// formed K-block by K-block;  3: as 1 without the split work (B pieces constant): the bare MFMA stream;  4: split work only.
template <int MODE> __device__ __forceinline__ void child(const u4 (&Ap)[3][2][4], const float (&Af)[2][32], float (&b)[32], v16 (&acc)[2])
{
    if constexpr (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 31; ++t)
#pragma unroll
            for (int it = 0; it < 2; ++it) acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[it][t], b[t], acc[it], 0, 0, 0);
    } else {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            u4 B1, B2, B3;
            if constexpr (MODE == 3) {
                B1 = Ap[0][0][kb]; B2 = Ap[1][0][kb]; B3 = Ap[2][0][kb];
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float v0 = b[8 * kb + 2 * jj], v1 = b[8 * kb + 2 * jj + 1];
                    const float p0 = top(v0), p1 = top(v1), r0 = v0 - p0, r1 = v1 - p1, q0 = top(r0), q1 = top(r1), s0 = r0 - q0, s1 = r1 - q1;
                    B1[jj] = pack_hi(v0, v1); B2[jj] = pack_hi(r0, r1); B3[jj] = pack_hi(s0, s1);
                }
            }
            if constexpr (MODE != 4) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ap[0][it][kb]), as_bf(B1), acc[it], 0, 0, 0);
                    acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ap[0][it][kb]), as_bf(B2), acc[it], 0, 0, 0);
                    acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ap[1][it][kb]), as_bf(B1), acc[it], 0, 0, 0);
                    acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ap[1][it][kb]), as_bf(B2), acc[it], 0, 0, 0);
                    acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ap[0][it][kb]), as_bf(B3), acc[it], 0, 0, 0);
                    acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ap[2][it][kb]), as_bf(B1), acc[it], 0, 0, 0);
                    if constexpr (MODE == 2) {
                        acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ap[1][it][kb]), as_bf(B3), acc[it], 0, 0, 0);
                        acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ap[2][it][kb]), as_bf(B2), acc[it], 0, 0, 0);
                        acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ap[2][it][kb]), as_bf(B3), acc[it], 0, 0, 0);
                    }
                }
            } else {
                acc[0][kb] += __uint_as_float(B1[0] ^ B2[1] ^ B3[2] ^ B1[3] ^ B2[0] ^ B3[1] ^ B1[2] ^ B2[3] ^ B3[0] ^ B1[1] ^ B2[2] ^ B3[3]);
            }
        }
    }
}
template <int MODE> __global__ void __launch_bounds__(512) k_time(int n, int waves, float* sink, long long* out)
{
    const int wave = threadIdx.x >> 6;
    if (wave >= waves) return;
    const unsigned lane = threadIdx.x & 63;
    u4 Ap[3][2][4];
    float Af[2][32];
    for (int s = 0; s < 3; ++s) for (int it = 0; it < 2; ++it) for (int kb = 0; kb < 4; ++kb) Ap[s][it][kb] = u4{0x3c003c00u + lane + s, 0x3c003c10u + it, 0x3c003c20u + kb, 0x3c003c30u};
    for (int it = 0; it < 2; ++it) for (int t = 0; t < 32; ++t) Af[it][t] = 1.0f / 64 + 1e-4f * (lane + t + it);
    float b[32];
    for (int t = 0; t < 32; ++t) b[t] = 0.01f * (1 + ((lane + t) & 15));
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        v16 acc[2] = {};
        child<MODE>(Ap, Af, b, acc);
#pragma unroll
        for (int t = 0; t < 32; ++t) b[t] = acc[t >> 4][t & 15] * 0.015625f;      // the next child's rows depend on this result
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    for (int t = 0; t < 32; ++t) s += b[t];
    if (s == 123.456f) sink[lane] = s;
    if (lane == 0) out[(size_t) blockIdx.x * 8 + wave] = t1 - t0;
}

static float bf16_exact(std::mt19937& g)
{
    std::uniform_int_distribution<int> mant(0, 127), ex(-6, 0);
    return std::ldexp(1.0f + mant(g) / 128.0f, ex(g));
}
int main()
{
    std::mt19937 g(7);
    float *dA, *dB, *dD;
    hipMalloc(&dA, 64 * 64 * 4); hipMalloc(&dB, 64 * 32 * 4); hipMalloc(&dD, 64 * 32 * 4);
    printf("1. layout / accumulation of v_mfma_f32_32x32x16_bf16\n");
    {
        std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32);
        for (auto& x : A) x = bf16_exact(g);
        for (auto& x : B) x = bf16_exact(g);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        for (int v = 0; v < 2; ++v) {
            hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, v);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0;
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double s = 0;
                for (int k = 0; k < 16; ++k) s += (double) A[i * 16 + k] * B[k * 32 + j];
                worst = std::max(worst, std::fabs(s - D[i * 32 + j]) / s);
            }
            printf("   k index of element j in lane half h = %s : max relative difference to the exact product %.3g\n", v == 0 ? "8 h + j" : "4 h + (j & 3) + 8 (j >> 2)", worst);
        }
        // one product a*b with 16 significant bits, added to a large accumulator term: is it exact in fp32?
        std::fill(A.begin(), A.end(), 0.0f); std::fill(B.begin(), B.end(), 0.0f);
        A[0] = 1.0f + 127.0f / 128; B[0] = 1.0f + 125.0f / 128;                 // (255/128)(253/128) = 64515 / 16384: 16 bits
        A[1] = std::ldexp(1.0f + 1.0f / 128, -20); B[32] = 1.0f + 3.0f / 128;   // a small second term
        A[2] = std::ldexp(1.0f, -130); B[64] = std::ldexp(1.0f, 20);              // a bf16 DENORMAL times 2^20 = 2^-110
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        const double e0 = (255.0 / 128) * (253.0 / 128), e1 = std::ldexp(129.0 / 128, -20) * (131.0 / 128);
        printf("   a1 b1 + a2 b2: device %.9g, fp32(exact sum) %.9g, fp32(fp32(a1 b1) + a2 b2) %.9g\n", D[0], (float) (e0 + e1), (float) e0 + (float) e1);
        std::fill(A.begin(), A.end(), 0.0f); std::fill(B.begin(), B.end(), 0.0f);
        A[0] = std::ldexp(1.0f, -130); B[0] = std::ldexp(1.0f, 20);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        printf("   bf16 denormal 2^-130 x 2^20: device %.6g (exact %.6g): denormal inputs are %s\n", D[0], std::ldexp(1.0, -110), D[0] != 0.0f ? "READ" : "FLUSHED");
        A[0] = std::ldexp(1.0f, -70); B[0] = std::ldexp(1.0f, -70);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        printf("   2^-70 x 2^-70 = 2^-140 (an fp32 denormal): device %.6g (exact %.6g)\n", D[0], std::ldexp(1.0, -140));
        // HOW does it round?  element (0,0) = sum_k A[0][k] B[k][0]; every product below is exact in fp32 on its own
        auto run = [&](std::vector<std::pair<float, float>> terms) {
            std::fill(A.begin(), A.end(), 0.0f); std::fill(B.begin(), B.end(), 0.0f);
            for (size_t k = 0; k < terms.size(); ++k) { A[k] = terms[k].first; B[k * 32] = terms[k].second; }
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            return D[0];
        };
        const float u = std::ldexp(1.0f, -23);       // ulp of 1
        printf("   rounding of a sum of products (ulp = 2^-23; RNE / toward zero / device), results as (x - 1) / ulp:\n");
        struct { const char* what; std::vector<std::pair<float, float>> t; double exact; } probes[] = {
            {"1 + 0.75 ulp                         ", {{1.0f, 1.0f}, {0.75f, u}}, 1.0 + 0.75 * u},
            {"1 + 0.25 ulp                         ", {{1.0f, 1.0f}, {0.25f, u}}, 1.0 + 0.25 * u},
            {"1 + 1.75 ulp                         ", {{1.0f, 1.0f}, {1.75f, u}}, 1.0 + 1.75 * u},
            {"1 + 8 x 0.125 ulp (= 1 ulp)          ", {{1.0f, 1.0f}, {0.125f, u}, {0.125f, u}, {0.125f, u}, {0.125f, u}, {0.125f, u}, {0.125f, u}, {0.125f, u}, {0.125f, u}}, 1.0 + u},
            {"1 + 12 x 0.0625 ulp (= 0.75 ulp)     ", {{1.0f, 1.0f}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}}, 1.0 + 0.75 * u},
            {"1 + 15 x 2^-10 ulp x 64 (= 0.9375ulp)", {{1.0f, 1.0f}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}, {0.0625f, u}}, 1.0 + 0.9375 * u},
            {"1.5 + 1.5 x 2^-24 .. (1.5 + 0.75 ulp)", {{1.5f, 1.0f}, {0.75f, u}}, 1.5 + 0.75 * u},
        };
        for (auto& pr : probes) {
            const float dev = run(pr.t);
            const float rne = (float) pr.exact;
            float rtz = rne; if ((double) rtz > pr.exact) rtz = std::nextafterf(rtz, 0.0f);
            const double base = pr.t[0].first;
            printf("     %s exact %.4f   RNE %.0f   toward zero %.0f   device %.0f\n", pr.what, (pr.exact - base) / u, (rne - base) / u, (rtz - base) / u, (dev - base) / u);
        }
    }
    printf("2. numerics: D[64][32] = A[64][64] B[64][32], relative error of an element against fp64 (relative to the column's largest element)\n");
    {
        std::vector<float> A(64 * 64), B(64 * 32), D(64 * 32);
        std::vector<double> R(64 * 32);
        for (int shape = 0; shape < 3; ++shape) {
            double sums[4][2] = {};   // max, sum of squares
            long count = 0;
            for (int rep = 0; rep < 50; ++rep) {
                // A: a transition matrix (rows sum to 1): exp(-t) on the diagonal-ish, the rest spread unevenly.  61 states, 3 padded with 0.
                std::uniform_real_distribution<double> U(0, 1);
                const double t = shape == 0 ? 0.05 : (shape == 1 ? 0.5 : 3.0);
                for (int i = 0; i < 64; ++i) {
                    double row[64], s = 0;
                    for (int j = 0; j < 64; ++j) { row[j] = (i < 61 && j < 61) ? std::pow(U(g), 6.0) : 0.0; s += (j != i) ? row[j] : 0; }
                    const double stay = std::exp(-t);
                    for (int j = 0; j < 64; ++j) A[i * 64 + j] = (float) (i < 61 && j < 61 ? (j == i ? stay : row[j] / s * (1 - stay)) : 0.0);
                }
                // B: conditional likelihoods, columns scaled so that the largest is in [0.5, 1); wide dynamic range below it
                for (int p = 0; p < 32; ++p) {
                    double col[64], mx = 0;
                    for (int j = 0; j < 64; ++j) { col[j] = j < 61 ? std::exp(-30.0 * std::pow(U(g), 2.0)) : 0.0; mx = std::max(mx, col[j]); }
                    int e; std::frexp(mx, &e);
                    for (int j = 0; j < 64; ++j) B[j * 32 + p] = (float) std::ldexp(col[j], -e);
                }
                for (int i = 0; i < 64; ++i) for (int p = 0; p < 32; ++p) {
                    double s = 0;
                    for (int j = 0; j < 64; ++j) s += (double) A[i * 64 + j] * B[j * 32 + p];
                    R[i * 32 + p] = s;
                }
                hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
                const int modes[4][2] = {{0, 0}, {1, 0}, {1, 1}, {2, 0}};
                for (int c = 0; c < 4; ++c) {
                    hipLaunchKernelGGL(k_numerics, dim3(1), dim3(64), 0, 0, dA, dB, dD, modes[c][0], modes[c][1]);
                    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
                    for (int i = 0; i < 61; ++i) for (int p = 0; p < 32; ++p) {
                        const double err = std::fabs(D[i * 32 + p] - R[i * 32 + p]) / R[i * 32 + p];
                        sums[c][0] = std::max(sums[c][0], err); sums[c][1] += err * err;
                    }
                }
                count += 61 * 32;
            }
            const char* names[4] = {"fp32 MFMA chain (today)", "bf16 x 3, six products, small first", "bf16 x 3, six products, large first", "bf16 x 3, nine products"};
            for (int c = 0; c < 4; ++c) printf("   branch %-5g %-38s max %.3g  rms %.3g   (fp32 epsilon 5.96e-8)\n", shape == 0 ? 0.05 : (shape == 1 ? 0.5 : 3.0), names[c], sums[c][0], std::sqrt(sums[c][1] / count));
        }
    }
    printf("2b. numerics, variants: signed relative error of an element against fp64: mean (bias), rms, max; units of 1e-8\n");
    for (int S : {61, 20}) {
        const int nkb = S > 32 ? 4 : 2, ntile = S > 32 ? 2 : 1, KW = 16 * nkb, MR = 32 * ntile;
        std::vector<float> A(MR * KW), B(KW * 32), D(MR * 32);
        std::vector<double> R(MR * 32);
        struct Var { const char* name; int split, nprod, order; };
        const Var vars[] = {{"fp32 MFMA chain (today)", -1, 0, 0}, {"truncated pieces, 6, per K-block", 0, 6, 0}, {"truncated pieces, 6, whole K", 0, 6, 2},
                            {"rounded pieces, 6, per K-block", 1, 6, 0}, {"rounded pieces, 6, per half of K", 1, 6, 1}, {"rounded pieces, 6, whole K", 1, 6, 2},
                            {"rounded pieces, 9, per half of K", 1, 9, 1}, {"rounded pieces, 9, whole K", 1, 9, 2},
                            {"rounded, 6, per K-block, 2 accumulators + VALU", -2, 0, 0}, {"rounded, 6, per K-block, 4 accumulators + VALU", -4, 0, 0}};
        const int NV = sizeof(vars) / sizeof(vars[0]);
        for (double t : {0.05, 1.0}) {
            double mean[NV] = {}, sq[NV] = {}, mx[NV] = {};
            long count = 0;
            for (int rep = 0; rep < 60; ++rep) {
                std::uniform_real_distribution<double> U(0, 1);
                for (int i = 0; i < MR; ++i) {
                    double row[64], sum = 0;
                    for (int j = 0; j < KW; ++j) { row[j] = (i < S && j < S) ? std::pow(U(g), 6.0) : 0.0; sum += (j != i) ? row[j] : 0; }
                    const double stay = std::exp(-t);
                    for (int j = 0; j < KW; ++j) A[i * KW + j] = (float) (i < S && j < S ? (j == i ? stay : row[j] / sum * (1 - stay)) : 0.0);
                }
                for (int p = 0; p < 32; ++p) {
                    double col[64], m = 0;
                    for (int j = 0; j < KW; ++j) { col[j] = j < S ? std::exp(-30.0 * std::pow(U(g), 2.0)) : 0.0; m = std::max(m, col[j]); }
                    int e; std::frexp(m, &e);
                    for (int j = 0; j < KW; ++j) B[j * 32 + p] = (float) std::ldexp(col[j], -e);
                }
                for (int i = 0; i < MR; ++i) for (int p = 0; p < 32; ++p) {
                    double sum = 0;
                    for (int j = 0; j < KW; ++j) sum += (double) A[i * KW + j] * B[j * 32 + p];
                    R[i * 32 + p] = sum;
                }
                hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
                for (int c = 0; c < NV; ++c) {
                    if (vars[c].split == -2) hipLaunchKernelGGL(k_numerics3<2>, dim3(1), dim3(64), 0, 0, dA, dB, dD, nkb, ntile);
                    else if (vars[c].split == -4) hipLaunchKernelGGL(k_numerics3<4>, dim3(1), dim3(64), 0, 0, dA, dB, dD, nkb, ntile);
                    else if (vars[c].split < 0) hipLaunchKernelGGL(k_numerics_f32, dim3(1), dim3(64), 0, 0, dA, dB, dD, nkb, ntile);
                    else hipLaunchKernelGGL(k_numerics2, dim3(1), dim3(64), 0, 0, dA, dB, dD, vars[c].split, vars[c].nprod, vars[c].order, nkb, ntile);
                    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
                    for (int i = 0; i < S; ++i) for (int p = 0; p < 32; ++p) {
                        const double err = (D[i * 32 + p] - R[i * 32 + p]) / R[i * 32 + p];
                        mean[c] += err; sq[c] += err * err; mx[c] = std::max(mx[c], std::fabs(err));
                    }
                }
                count += S * 32;
            }
            for (int c = 0; c < NV; ++c) printf("   %d states, branch %-5g %-46s bias %+7.3f  rms %7.3f  max %7.2f\n", S, t, vars[c].name, mean[c] / count * 1e8, std::sqrt(sq[c] / count) * 1e8, mx[c] * 1e8);
        }
    }
    printf("3. one child factor at 61 states, cycles per factor and wave (s_memrealtime 100 MHz ticks x clock / 100 MHz), 256 workgroups\n");
    {
        float* sink; long long* out;
        hipMalloc(&sink, 4096); hipMalloc(&out, 256 * 8 * 8);
        std::vector<long long> h(256 * 8);
        const int n = 2000;
        const char* names[5] = {"fp32: 62 x 32x32x2_f32", "split, 48 x 32x32x16_bf16 + B pieces", "split, 72 x 32x32x16_bf16 + B pieces", "48 x 32x32x16_bf16, B pieces constant", "the B pieces alone (VALU)"};
        for (int waves = 4; waves <= 8; waves += 4)
            for (int mode = 0; mode < 5; ++mode) {
                for (int rep = 0; rep < 2; ++rep) {
                    hipMemset(out, 0, 256 * 8 * 8);
                    switch (mode) {
                    case 0: hipLaunchKernelGGL(k_time<0>, dim3(256), dim3(512), 0, 0, n, waves, sink, out); break;
                    case 1: hipLaunchKernelGGL(k_time<1>, dim3(256), dim3(512), 0, 0, n, waves, sink, out); break;
                    case 2: hipLaunchKernelGGL(k_time<2>, dim3(256), dim3(512), 0, 0, n, waves, sink, out); break;
                    case 3: hipLaunchKernelGGL(k_time<3>, dim3(256), dim3(512), 0, 0, n, waves, sink, out); break;
                    default: hipLaunchKernelGGL(k_time<4>, dim3(256), dim3(512), 0, 0, n, waves, sink, out); break;
                    }
                    if (hipDeviceSynchronize() != hipSuccess) { printf("error\n"); return 1; }
                }
                hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
                std::vector<double> d;
                for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) d.push_back(h[b * 8 + w] / 100.0);
                std::sort(d.begin(), d.end());
                printf("   %d waves per SIMD  %-42s %8.1f us per wave for %d factors = %7.3f us per factor\n", waves / 4, names[mode], d[d.size() / 2], n, d[d.size() / 2] / n);
                fflush(stdout);
            }
    }
    return 0;
}
