// mbamd_kernels.h -- hand-written HIP kernels (gfx950 / CDNA4) of the conditional-likelihood engine.
//
// Replaces, on the device, the reference's native loops in src/likelihood.c:
//   k_walk4              CondLikeDown/Root_NUC4* + CondLikeScaler_NUC4* + RemoveNodeScalers (4-state, mbamd_walk4.h)
//   k_partials_gen       CondLikeDown_Gen / _NY98 (general state count), k_rescale_gen = CondLikeScaler_Gen
//   k_transition_matrices TiProbs_Gen / TiProbs_GenCov (src/likelihood.c:9424-9700)
//   k_integrate_lnl      Likelihood_Gen / _NY98 / _NUC4 root integration (src/likelihood.c:5764, 6975, 6238)
//   k_scale_*            Copy/Reset/RemoveNodeScalers bookkeeping (src/likelihood.c:7981-8131)
//
// Data layout in HBM (all fp32 unless noted), P_pad = patterns rounded up to 64:
//   4-state partials   : f4 [buffer][P_pad/64][K][64]  buffer-major arena: one f4 = the 4 states of (category, pattern);
//                        1 KiB contiguous per (node update, block, category).  The waves of a launch run one program at one
//                        pace: at any moment they write into one node's few MB -- a moving window like a fill (block-major
//                        until round 3: 1 KiB pieces into regions 12 MB apart, and a box-dependent kernel time; see
//                        profiles/r03_exp_walk4_linear.txt).  4-state tips are four 64-bit STATE BITPLANES per
//                        (pattern block, tip): uint64 [P_pad/64][buffer][4], plane i bit l = state i compatible with
//                        pattern l of the block (missing = all four); 4-state node exponents are int8
//                        [P_pad/64][scale buffer][K][64] (one per pattern AND category), cumulative exponents
//                        int32 [K][P_pad] per scale buffer, allocated when a buffer is first used that way.
//   general partials   : float  [P_pad/32][K][S][32]   tile-major (gen_index): lanes = consecutive patterns of a
//                        32-pattern tile -> coalesced; a tile of one buffer is one contiguous K*S*128-byte run
//   matrices           : float  [K][SP][SP] transposed (mT[k][j][i] = P_k(i->j)), zero padded to SP
//                        (SP = 4 on the 4-state path); MFMA path: + A-operand copy, see mbamd_kernels_mfma.h
//   tip states         : uint8  [P_pad]               value >= S = missing
//   scale buffers      : int32  [P_pad]               binary exponents (see below), node and cumulative alike
//   pattern weights    : double [P_pad]; site lnL double [P_pad]
//
// Scaling: instead of the reference's "divide by the max, store logf(max)" (CondLikeScaler_*,
// src/likelihood.c:4939-4988) a node is rescaled by the power of two 2^-e, e = frexp-exponent of
// the per-pattern max over (category,state).  Multiplication by 2^-e is exact in fp32, the factor
// is stored as the integer e, and accumulate/remove on cumulative buffers is exact integer
// arithmetic; ln(scale) = e*ln2 is formed in fp64 only at the root.  One v_frexp_exp + v_ldexp
// per value, no logf, no division.
//
// 4-state path: the exponent is per (pattern, category) -- columns of different categories never meet before the
// root, where k_integrate_lnl_s4 recombines them exactly (see mbamd_walk4.h).
//
// What is specific to gfx950 -- address spaces, intrinsics, cross-lane sums, LDS-DMA -- is reached through the small primitive
// headers <mbamd_dev_*.h> (csrc/device/); the TEST-ONLY host emulation (tests/hostemu/) supplies plain-C++ headers of the same
// names on its include path, so the kernel bodies below are the code the CPU CI of the host logic runs.
#ifndef MBAMD_KERNELS_H_
#define MBAMD_KERNELS_H_

#include <stdint.h>

#include <mbamd_dev_base.h>      // MBAMD_AS_GLOBAL / MBAMD_AS_CONST, f4, mbd_* primitives (csrc/device/)

namespace mbamd {

template <class T> __device__ __forceinline__ MBAMD_AS_GLOBAL T* as_global(T* p)
{
    return (MBAMD_AS_GLOBAL T*) (uintptr_t) p;
}
template <class T> __device__ __forceinline__ const MBAMD_AS_CONST T* as_const(const T* p)
{
    return (const MBAMD_AS_CONST T*) (uintptr_t) p;
}

// Block-major addressing (4-state path): consecutive 64-pattern blocks of one buffer are `*_stride`
// elements apart; within a block a partials buffer is [K][64] f4, tips / scale buffers are [64].
// The general-state path keeps linear [P_pad] arrays, which is the same formula with stride 64.
struct BlockGeom {
    unsigned long pstride;     // f4 elements between the blocks of a partials buffer
    unsigned tstride;          // uint64 elements between the blocks of the tip bitplanes
    unsigned sstride;          // int32 elements between the blocks of a scale buffer
};
__host__ __device__ inline size_t blk_index(int c, size_t stride) { return (size_t) (c >> 6) * stride + (size_t) (c & 63); }

// General-state partials (any S other than the 4-state arenas) are stored TILE-MAJOR:
//     [P_pad / 32 tiles][K categories][S states][32 patterns]   fp32
// so the K*S*32 values of one 32-pattern tile -- what one workgroup of the MFMA kernels reads per child
// and writes per result -- are one contiguous run (10 KiB for 20 states x 4 categories, 7.6 KiB for 61
// states) instead of K*S separate 128-byte segments 4*P_pad bytes apart (one DRAM page each).
// gen_base(c) is the offset of element (k = 0, i = 0) of pattern c; element (k, i) is (k*S + i) * 32 further.
__host__ __device__ inline size_t gen_base(int K, int S, int c) { return (size_t) (c >> 5) * K * S * 32 + (c & 31); }
__host__ __device__ inline size_t gen_index(int K, int S, int k, int i, int c) { return gen_base(K, S, c) + ((size_t) k * S + i) * 32; }

enum ChildKind : uint8_t {
    CHILD_PARTIALS = 0,   // dense partials in HBM, not written by this launch
    CHILD_STATES   = 1,   // compact tip: uint8 state codes
    CHILD_LDS      = 2,   // produced earlier in this launch and still resident in the LDS stack
    CHILD_RELOAD   = 3    // produced earlier in this launch by the same lane, evicted from LDS
};
enum ScaleMode : uint8_t { SCALE_NONE = 0, SCALE_WRITE = 1, SCALE_READ = 2 };

// One partial-likelihood operation, resolved to device pointers by the host.  Wave-uniform:
// fetched with scalar loads.
struct alignas(16) PartialsOp {
    float*       dst;
    const void*  c1;
    const void*  c2;
    const float* m1;
    const float* m2;
    int32_t*     scale;        // node scale buffer written (SCALE_WRITE) or read (SCALE_READ)
    uint8_t      c1_kind, c2_kind;
    uint8_t      c1_slot, c2_slot;
    uint8_t      dst_slot;     // 0xFF: do not keep in LDS
    uint8_t      scale_mode;
    uint8_t      flags;        // MBAMD_OP_* (tree-walk kernel)
    uint8_t      pad_[1];
    int32_t      pad2_[2];
};
static_assert(sizeof(PartialsOp) == 64, "PartialsOp must be 64 bytes");

#define MBAMD_NO_SLOT 0xFF
#define MBAMD_MAX_TABLES 4      // operation tables (lists / independent sub-lists) one general-state launch can take

// ---------------------------------------------------------------------------------------------
// exact power-of-two rescaling helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int scale_exponent(float mx)
{
    int e = (mx > 0.0f && mx < 3.0e38f) ? mbd_frexp_exp(mx) : 0;
    e = e < -126 ? -126 : e;      // keep 2^-e a normal float even for denormal maxima
    e = e > 126 ? 126 : e;
    return e;
}
__device__ __forceinline__ float scale_pow2(float v, int neg_e)
{
    return mbd_ldexp(v, neg_e);
}

__device__ __forceinline__ float max4(f4 v) { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); }

// the block sums an integration kernel wrote (pinned host memory, read through its device address) -> one double in device memory,
// in a fixed order: thread t adds the sums t, t + 256, ..., then a tree over the 256 partial sums (mbamdReduceLogLikelihood)
__global__ void __launch_bounds__(256)
k_sum_block_sums(const double* __restrict__ sums, int n, double* __restrict__ out)
{
    double* part = mbd_dyn_lds<double>();        // 256 doubles of dynamic LDS
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += sums[i];
    part[threadIdx.x] = s;
    MBAMD_SYNC();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int) threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        MBAMD_SYNC();
    }
    if (threadIdx.x == 0) *out = part[0];
}

}  // namespace mbamd
#include "mbamd_walk4.h"
#include "mbamd_walkg.h"
namespace mbamd {

// ---------------------------------------------------------------------------------------------
// General state count: level-synchronous kernel.  grid = (P_pad/64, ops in this dependency level),
// one thread per pattern, loop over categories.  The tile-major layout makes every load/store two
// coalesced 128-byte half-wave accesses; the (transposed, zero-padded) transition matrix column
// mT[k][j][0..SP) is wave-uniform and is consumed as scalar operands of v_fmac.
//   FUSED_K > 0 : K == FUSED_K, all K*SP outputs stay in registers and the per-pattern rescale is
//                 fused (power-of-two, see top of file);
//   FUSED_K == 0: any K; writes unscaled partials, k_rescale_gen does the scaling pass.
// ---------------------------------------------------------------------------------------------
template <int SP>
__device__ __forceinline__ void gen_child_factor(const void* ptr, int kind, const float* mT_, int S, int K, int k,
                                                 int c, float (&f)[SP])
{
    if (kind == CHILD_STATES) {
        const unsigned s = as_global(reinterpret_cast<const uint8_t*>(ptr))[c];
        if (s >= (unsigned) S) {
#pragma unroll
            for (int i = 0; i < SP; ++i) f[i] = 1.0f;
        } else {
            const MBAMD_AS_GLOBAL float* col = as_global(mT_) + (size_t) s * SP;   // P(i -> s), all i: contiguous
#pragma unroll
            for (int i = 0; i < SP; ++i) f[i] = col[i];
        }
    } else {
        const MBAMD_AS_CONST float* __restrict__ mT = as_const(mT_);
        const MBAMD_AS_GLOBAL float* __restrict__ cl =
            as_global(reinterpret_cast<const float*>(ptr)) + gen_index(K, S, k, 0, c);
#pragma unroll
        for (int i = 0; i < SP; ++i) f[i] = 0.0f;
#pragma unroll 2
        for (int j = 0; j < S; ++j) {
            const float vj = cl[(size_t) j * 32];
            const MBAMD_AS_CONST float* __restrict__ col = mT + (size_t) j * SP;
#pragma unroll
            for (int i = 0; i < SP; ++i) f[i] = fmaf(col[i], vj, f[i]);
        }
    }
}

template <int SP, int FUSED_K>
__global__ void __launch_bounds__(64)
k_partials_gen(const PartialsOp* __restrict__ ops, int S, int K, int Ppad, int32_t* __restrict__ cumulative)
{
    const MBAMD_AS_CONST PartialsOp* __restrict__ op = as_const(ops) + blockIdx.y;
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int k1 = op->c1_kind, k2 = op->c2_kind;
    const void* c1 = op->c1;
    const void* c2 = op->c2;
    const float* m1 = op->m1;
    const float* m2 = op->m2;
    MBAMD_AS_GLOBAL float* __restrict__ dst = as_global(op->dst) + gen_base(K, S, c);
    MBAMD_AS_GLOBAL int32_t* sc = as_global(op->scale);
    const int mode = op->scale_mode;

    if constexpr (FUSED_K > 0) {
        constexpr int KK = FUSED_K > 0 ? FUSED_K : 1;
        float out[KK][SP];
        float mx = 0.0f;
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            float f2[SP];
            gen_child_factor<SP>(c1, k1, m1 + (size_t) k * SP * SP, S, K, k, c, out[k]);
            gen_child_factor<SP>(c2, k2, m2 + (size_t) k * SP * SP, S, K, k, c, f2);
#pragma unroll
            for (int i = 0; i < SP; ++i) {
                out[k][i] *= f2[i];
                if (i < S) mx = fmaxf(mx, out[k][i]);
            }
        }
        int e = 0;
        if (mode == SCALE_WRITE) {
            e = scale_exponent(mx);
            sc[c] = e;
            if (cumulative != nullptr && e != 0) atomicAdd(cumulative + c, e);
        } else if (mode == SCALE_READ) {
            e = sc[c];
        }
#pragma unroll
        for (int k = 0; k < KK; ++k) {
#pragma unroll
            for (int i = 0; i < SP; ++i)
                if (i < S) dst[((size_t) k * S + i) * 32] = (mode != SCALE_NONE) ? scale_pow2(out[k][i], -e) : out[k][i];
        }
    } else {
        for (int k = 0; k < K; ++k) {
            float f1[SP], f2[SP];
            gen_child_factor<SP>(c1, k1, m1 + (size_t) k * SP * SP, S, K, k, c, f1);
            gen_child_factor<SP>(c2, k2, m2 + (size_t) k * SP * SP, S, K, k, c, f2);
#pragma unroll
            for (int i = 0; i < SP; ++i)
                if (i < S) dst[((size_t) k * S + i) * 32] = f1[i] * f2[i];
        }
    }
}

// scaling pass for the unfused general path (CondLikeScaler_Gen, src/likelihood.c:4939-4988)
__global__ void __launch_bounds__(64)
k_rescale_gen(const PartialsOp* __restrict__ ops, int S, int K, int Ppad, int32_t* __restrict__ cumulative)
{
    const MBAMD_AS_CONST PartialsOp* __restrict__ op = as_const(ops) + blockIdx.y;
    const int mode = op->scale_mode;
    if (mode == SCALE_NONE) return;
    const int c = blockIdx.x * 64 + threadIdx.x;
    MBAMD_AS_GLOBAL float* __restrict__ dst = as_global(op->dst) + gen_base(K, S, c);
    MBAMD_AS_GLOBAL int32_t* sc = as_global(op->scale);
    const int n = K * S;
    int e;
    if (mode == SCALE_WRITE) {
        float mx = 0.0f;
        for (int r = 0; r < n; ++r) mx = fmaxf(mx, dst[(size_t) r * 32]);
        e = scale_exponent(mx);
        sc[c] = e;
        if (cumulative != nullptr && e != 0) atomicAdd(cumulative + c, e);
    } else {
        e = sc[c];
    }
    if (e != 0)
        for (int r = 0; r < n; ++r) dst[(size_t) r * 32] = scale_pow2(dst[(size_t) r * 32], -e);
}

// ---------------------------------------------------------------------------------------------
// Transition probabilities: P_k = U diag(exp(lambda * t * rate_k)) U^-1, fp64 math -> fp32 store,
// negatives clamped to 0 (TiProbs_Gen, src/likelihood.c:9528-9545).  grid = count*K workgroups.
// eig = [U (S*S) | U^-1 (S*S) | lambda (S)] doubles.  Output layout per matrix buffer:
//   transposed == 0: out[k][i][j] (4-state path), stride SP = S
//   transposed == 1: out[k][j][i] zero-padded rows of SP floats (general path)
// ---------------------------------------------------------------------------------------------
struct MatrixJob {
    float* out;          // matrix buffer (K matrices)
    double length;       // branch length
    const double* eig;   // its eigen-system [U | U^-1 | lambda]: jobs of several systems share one launch
    double pad_;
};

// category rates travel as a kernel argument (no host-to-device copy per beagleSetCategoryRates)
#define MBAMD_MAX_RATES 16
struct RatesArg { double r[MBAMD_MAX_RATES]; };

// exp(lambda*t) hoisted: one thread per (job, k, s) fills ev[(job*K+k)*S + s]; the matrix kernel
// below then reads it (second launch on the same stream, so no barrier is needed).
// `jobs` may live in pinned host memory (read once, directly over the host link).
__global__ void __launch_bounds__(256)
k_eigen_exponentials(const MatrixJob* __restrict__ jobs, RatesArg rates, int S, int K, int total, double* __restrict__ ev)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int s = g % S, bk = g / S;
    const int b = bk / K, k = bk % K;
    const double* __restrict__ lam = jobs[b].eig + (size_t) 2 * S * S;
    ev[g] = exp(lam[s] * jobs[b].length * rates.r[k]);
}

// 4-state path: one thread per (branch, category) does the whole 4x4 matrix, exps included
// (TiProbs_Gen for S = 4, src/likelihood.c:9498-9545); output transposed mT[j][i] = P(i->j).
__global__ void __launch_bounds__(256)
k_transition_matrices_s4(const MatrixJob* __restrict__ jobs, RatesArg rates, int K, int total)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int b = g / K, k = g % K;
    const MatrixJob job = jobs[b];
    const double* __restrict__ U = job.eig;
    const double* __restrict__ Ui = job.eig + 16;
    const double* __restrict__ lam = job.eig + 32;
    double e[4];
    for (int s = 0; s < 4; ++s) e[s] = exp(lam[s] * job.length * rates.r[k]);
    float* __restrict__ out = job.out + (size_t) k * 16;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double sum = 0.0;
            for (int s = 0; s < 4; ++s) sum += U[i * 4 + s] * e[s] * Ui[s * 4 + j];
            out[j * 4 + i] = (sum < 0.0) ? 0.0f : (float) sum;
        }
}

// a compiled program from the pinned ring into its device buffer (round 6: a launch of ours instead of hipMemcpyAsync, whose call and
// blit kernel were 10-15 us of the 30-40 us between the matrix kernel and the walk of a chain's full-tree evaluation)
typedef unsigned copy16_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
k_copy_from_ring(const copy16_t* __restrict__ src, copy16_t* __restrict__ dst, unsigned n16)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

__global__ void __launch_bounds__(256)
k_copy_from_ring4(const unsigned* __restrict__ src, unsigned* __restrict__ dst, unsigned n4)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

// The same for the one or two matrices of a branch move (round 6): the jobs travel in the kernel arguments -- read from the pinned ring,
// as the kernel above does, the first thing every thread did was a round trip over the host link (2 of the kernel's 5 us, and the
// path kernel waits behind it in 86 % of a chain's generations).
#define MBAMD_S4_INLINE_JOBS 8
struct MatrixJobs4 { MatrixJob j[MBAMD_S4_INLINE_JOBS]; };
__global__ void __launch_bounds__(64)
k_transition_matrices_s4_inline(MatrixJobs4 jobs, RatesArg rates, int K, int total)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int b = g / K, k = g % K;
    const MatrixJob job = jobs.j[b];
    const double* __restrict__ U = job.eig;
    const double* __restrict__ Ui = job.eig + 16;
    const double* __restrict__ lam = job.eig + 32;
    double e[4];
    for (int s = 0; s < 4; ++s) e[s] = exp(lam[s] * job.length * rates.r[k]);
    float* __restrict__ out = job.out + (size_t) k * 16;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double sum = 0.0;
            for (int s = 0; s < 4; ++s) sum += U[i * 4 + s] * e[s] * Ui[s * 4 + j];
            out[j * 4 + i] = (sum < 0.0) ? 0.0f : (float) sum;
        }
}

// packedT > 0: additionally write the MFMA A-operand copy behind the K transposed matrices:
//   packed[((k*NT + i/32)*T + j/2)*64 + (i%32) + 32*(j%2)] = P_k(i->j),  NT = ceil(S/32), T = packedT = ceil(S/2)
// wgTab > 0: additionally scatter into the tree-walk tables of category k, wgTab floats into the buffer (mbamd_walkg.h)
__global__ void __launch_bounds__(256)
k_transition_matrices_ev(const MatrixJob* __restrict__ jobs, const double* __restrict__ ev, RatesArg rates, int S, int SP, int K,
                         int transposed, int packedT, size_t wgTab)
{
    // ev == nullptr (up to 64 states): the block forms its own S exponentials -- the expression of k_eigen_exponentials, one launch
    // less in front of every small-state evaluation (round 5: a standard-data class is a few hundred patterns, all launch latency)
    __shared__ double own[64];
    const int b = blockIdx.x / K, k = blockIdx.x % K;
    const double* __restrict__ U = jobs[b].eig;
    const double* __restrict__ Ui = jobs[b].eig + (size_t) S * S;
    if (ev == nullptr) {
        const double* __restrict__ lam = jobs[b].eig + (size_t) 2 * S * S;
        if ((int) threadIdx.x < S) own[threadIdx.x] = exp(lam[threadIdx.x] * jobs[b].length * rates.r[k]);
        MBAMD_SYNC();
    }
    const double* __restrict__ e = ev ? ev + (size_t) blockIdx.x * S : own;
    float* __restrict__ out = jobs[b].out + (size_t) k * SP * SP;
    for (int idx = threadIdx.x; idx < S * S; idx += blockDim.x) {
        // consecutive threads take consecutive j so that U^-1 reads are coalesced
        const int i = idx / S, j = idx % S;
        double sum = 0.0;
        for (int s = 0; s < S; ++s) sum += U[i * S + s] * e[s] * Ui[s * S + j];
        const float v = (sum < 0.0) ? 0.0f : (float) sum;
        if (transposed) out[(size_t) j * SP + i] = v;
        else            out[(size_t) i * SP + j] = v;
        if (packedT > 0) {
            const int NT = (S + 31) / 32;
            float* __restrict__ packed = jobs[b].out + (size_t) K * SP * SP;
            packed[((size_t) (k * NT + i / 32) * packedT + j / 2) * 64 + (i % 32) + 32 * (j % 2)] = v;
        }
        if (wgTab > 0) wg_table_put(jobs[b].out + wgTab + (size_t) k * wg_table_floats(S), S, i, j, v);
    }
}

// ---------------------------------------------------------------------------------------------
// Eigen-systems of reversible rate matrices ON THE DEVICE (replaces the host step UpDateCijk ->
// GetEigens -> EigensForRealMatrix: balance, Hessenberg, shifted QR, reference src/likelihood.c:10476-10804,
// src/utils.c:10251-11339, for the models whose Q is reversible -- GTR, every amino-acid matrix, the NY98 codon categories).
// With d = sqrt(pi), B = D Q D^-1 is symmetric: a cyclic Jacobi iteration in fp64 (round-robin pairing: n/2 disjoint
// rotations per step, rows then columns, all 256 threads busy) gives B = V L V^T, and
//     U = D^-1 V,   U^-1 = V^T D,   lambda = diag(L)
// is what beagleSetEigenDecomposition receives -- written straight into the instance's eigen buffers
// [U | U^-1 | lambda] (k_transition_matrices_* read them from there).  mode 1: `q` holds exchangeabilities r_ij instead of
// rates; Q_ij = r_ij pi_j, rows sum to zero, scaled to one expected substitution per unit time (SetProteinQMatrix,
// src/likelihood.c:8765-8880) -- the Q build on the device as well.
// One workgroup of 256 (up to 32 states) or 1 024 threads per matrix, S <= 64; dynamic LDS: eigen_lds_doubles(S) doubles (68 KiB at 61 states).
// ---------------------------------------------------------------------------------------------
// out = [U | U^-1 | lambda | V]: V (S x S) are the orthonormal eigenvectors of the symmetrised matrix -- what a later call
// starts from (`warm` = the V of an eigen-system of a NEARBY rate matrix, e.g. the chain's current state when a move proposes
// new rates: two or three sweeps instead of nine), or nullptr for a cold start from the identity.
struct EigenJob { const double* q; const double* pi; double* out; const double* warm; int mode; int pad_; };
// n = S rounded up to even (an odd S gets a dummy index whose row and column stay zero: its rotations are identities)
__host__ __device__ inline size_t eigen_lds_doubles(int S) { const size_t n = (size_t) ((S + 1) & ~1); return 3 * n * (n + 1) + 4 * (n / 2 + 1) + 1032; }
// TY: rows of the 32-wide thread grid -- 8 (256 threads: up to 32 states) or 32 (1 024 threads beyond: a Jacobi step is a chain of LDS
// round trips and two barriers, and with one wave per SIMD nothing hides them; four waves per SIMD do: 538 -> 366 us for three warm-started 61-state systems)
template <int TY>
__global__ void __launch_bounds__(32 * TY)
k_eigen_reversible(const EigenJob* __restrict__ jobs, int S, int sweeps)
{
    constexpr int NTHREADS = 32 * TY, NUA = 32 / TY, NUV = 64 / TY;
    double* lds = mbd_dyn_lds<double>();
    const EigenJob job = jobs[blockIdx.x];
    const int n = (S + 1) & ~1, m = n / 2, LD = n + 1;
    double* A = lds;                         // [n][LD]  the symmetrised matrix, diagonalised in place
    double* V = A + (size_t) n * LD;         // [n][LD]  accumulated rotations
    double* Wt = V + (size_t) n * LD;        // [n][LD]  warm start: B V0
    double* rc = Wt + (size_t) n * LD;       // [m] cosines
    double* rs = rc + (m + 1);               // [m] sines
    int* rp = reinterpret_cast<int*>(rs + (m + 1));   // [m] pair (p, q), p < q
    int* rq = rp + 2 * (m + 1);
    double* red = rs + (m + 1) + 2 * (m + 1);         // [NTHREADS + 8] reduction scratch
    const int tid = threadIdx.x;
    const int ty = tid >> 5, tx = tid & 31;  // a TY x 32 thread grid over (pair | row, pair): no divisions in the loops
    // ---- B = D Q D^-1, symmetrised (mode 1: Q from exchangeabilities first) ---------------------------------------------
    double scale = 1.0;
    if (job.mode == 1) {
        double part = 0.0;
        for (int i = ty; i < S; i += TY) {
            double row = 0.0;
            for (int j2 = tx; j2 < S; j2 += 32) if (j2 != i) row += job.q[(size_t) i * S + j2] * job.pi[j2];
            part += job.pi[i] * row;
        }
        red[tid] = part;
        MBAMD_SYNC();
        if (tid == 0) { double tot = 0.0; for (int t = 0; t < NTHREADS; ++t) tot += red[t]; red[NTHREADS] = 1.0 / tot; }
        MBAMD_SYNC();
        scale = red[NTHREADS];
        MBAMD_SYNC();
    }
    for (int i = ty; i < n; i += TY)
        for (int j2 = tx; j2 < n; j2 += 32) {
            double b = 0.0;
            if (i != j2 && i < S && j2 < S) {
                const double di = sqrt(job.pi[i]), dj = sqrt(job.pi[j2]);
                const double qij = job.mode == 1 ? job.q[(size_t) i * S + j2] * job.pi[j2] * scale : job.q[(size_t) i * S + j2];
                const double qji = job.mode == 1 ? job.q[(size_t) j2 * S + i] * job.pi[i] * scale : job.q[(size_t) j2 * S + i];
                b = 0.5 * (di * qij / dj + dj * qji / di);
            }
            A[i * LD + j2] = b;
            V[i * LD + j2] = (i == j2) ? 1.0 : 0.0;
        }
    MBAMD_SYNC();
    for (int i = tid; i < S; i += NTHREADS) {     // the diagonal: minus the row sum of Q (what makes the rows of Q sum to zero)
        double row = 0.0;
        for (int j2 = 0; j2 < S; ++j2)
            if (j2 != i) row += job.mode == 1 ? job.q[(size_t) i * S + j2] * job.pi[j2] * scale : job.q[(size_t) i * S + j2];
        A[i * LD + i] = -row;
    }
    if (tid == 0) red[NTHREADS + 3] = 0.0;
    MBAMD_SYNC();
    if (job.warm != nullptr) {
        // A <- V0^T B V0, V <- V0: the rotations then only have to undo what the rate matrix changed since V0 was computed
        for (int i = ty; i < n; i += TY)
            for (int j2 = tx; j2 < n; j2 += 32) V[i * LD + j2] = (i < S && j2 < S) ? job.warm[(size_t) i * S + j2] : (i == j2 ? 1.0 : 0.0);
        MBAMD_SYNC();
        for (int i = ty; i < n; i += TY)
            for (int j2 = tx; j2 < n; j2 += 32) {
                double acc = 0.0;
                for (int k2 = 0; k2 < n; ++k2) acc += A[i * LD + k2] * V[k2 * LD + j2];
                Wt[i * LD + j2] = acc;
            }
        MBAMD_SYNC();
        for (int i = ty; i < n; i += TY)
            for (int j2 = tx; j2 < n; j2 += 32) {
                double acc = 0.0;
                for (int k2 = 0; k2 < n; ++k2) acc += V[k2 * LD + i] * Wt[k2 * LD + j2];
                A[i * LD + j2] = acc;
            }
        MBAMD_SYNC();
    }
    const double enough = job.warm != nullptr ? 1.0 : 2.0;      // converged checks in a row that end the iteration
    for (int sweep = 0; sweep < sweeps; ++sweep) {
        for (int r = 0; r < n - 1; ++r) {
            // round-robin: pair 0 = (n-1, r), pair i = ((r + i) mod (n-1), (r - i) mod (n-1)): n/2 disjoint rotations
            if (tid < m) {
                int p = r + tid, q = r - tid;                     // (mod n-1 without a division: r, tid < n-1)
                if (p >= n - 1) p -= n - 1;
                if (q < 0) q += n - 1;
                if (tid == 0) { p = n - 1; q = r; }
                if (p > q) { const int t = p; p = q; q = t; }
                double c = 1.0, sn = 0.0;
                const double apq = A[p * LD + q];
                if (fabs(apq) > 1e-300) {
                    double t;
                    mbd_jacobi_rotation(A[p * LD + p], A[q * LD + q], apq, c, t);
                    sn = t * c;
                }
                rc[tid] = c; rs[tid] = sn; rp[tid] = p; rq[tid] = q;
            }
            MBAMD_SYNC();
            // A <- J^T A J, one 2 x 2 block per (row pair a, column pair b): every element belongs to exactly one block.
            // All loads of a thread first, then the arithmetic, then all stores: one LDS round trip per step instead of one
            // per block (the compiler cannot reorder loads over stores to the same array).
            if (tx < m) {
                const int p2 = rp[tx], q2 = rq[tx];
                const double c2 = rc[tx], s2 = rs[tx];
                double x[NUA][4], v[NUV][2];
                int pa[NUA], qa[NUA];
                double ca[NUA], sa[NUA];
#pragma unroll
                for (int u = 0; u < NUA; ++u) {
                    const int a = (ty + TY * u < m) ? ty + TY * u : m - 1;
                    pa[u] = rp[a]; qa[u] = rq[a]; ca[u] = rc[a]; sa[u] = rs[a];
                    x[u][0] = A[pa[u] * LD + p2]; x[u][1] = A[pa[u] * LD + q2];
                    x[u][2] = A[qa[u] * LD + p2]; x[u][3] = A[qa[u] * LD + q2];
                }
#pragma unroll
                for (int u = 0; u < NUV; ++u) {
                    const int k = (ty + TY * u < n) ? ty + TY * u : n - 1;
                    v[u][0] = V[k * LD + p2]; v[u][1] = V[k * LD + q2];
                }
#pragma unroll
                for (int u = 0; u < NUA; ++u)
                    if (ty + TY * u < m) {
                        const double y11 = ca[u] * x[u][0] - sa[u] * x[u][2], y12 = ca[u] * x[u][1] - sa[u] * x[u][3];
                        const double y21 = sa[u] * x[u][0] + ca[u] * x[u][2], y22 = sa[u] * x[u][1] + ca[u] * x[u][3];
                        A[pa[u] * LD + p2] = c2 * y11 - s2 * y12; A[pa[u] * LD + q2] = s2 * y11 + c2 * y12;
                        A[qa[u] * LD + p2] = c2 * y21 - s2 * y22; A[qa[u] * LD + q2] = s2 * y21 + c2 * y22;
                    }
#pragma unroll
                for (int u = 0; u < NUV; ++u)
                    if (ty + TY * u < n) {                         // V <- V J
                        const int k = ty + TY * u;
                        V[k * LD + p2] = c2 * v[u][0] - s2 * v[u][1];
                        V[k * LD + q2] = s2 * v[u][0] + c2 * v[u][1];
                    }
            }
            MBAMD_SYNC();
        }
        // converged?  (sum of squared off-diagonal elements against the squared diagonal)
        double off = 0.0, diag = 0.0;
        for (int i = ty; i < n; i += TY)
            for (int j2 = tx; j2 < n; j2 += 32) { const double v = A[i * LD + j2]; if (i == j2) diag += v * v; else off += v * v; }
        double o4, d4;
        mbd_block_sum2<NTHREADS>(off, diag, red, tid, o4, d4);
        if (tid == 0) {
            // (quadratic convergence: once the off-diagonal mass is below 1e-20 of the diagonal's, one more sweep takes it to rounding)
            red[NTHREADS + 2] = (o4 <= 1e-20 * d4) ? red[NTHREADS + 3] + 1.0 : 0.0;
            red[NTHREADS + 3] = red[NTHREADS + 2];
        }
        MBAMD_SYNC();
        if (red[NTHREADS + 2] >= enough) break;
    }
    // ---- U = D^-1 V, U^-1 = V^T D, lambda ------------------------------------------------------------------------------
    double* U = job.out;
    double* Ui = job.out + (size_t) S * S;
    double* lam = job.out + (size_t) 2 * S * S;
    for (int i = ty; i < S; i += TY)
        for (int s2 = tx; s2 < S; s2 += 32) {
            const double di = sqrt(job.pi[i]);
            U[(size_t) i * S + s2] = V[i * LD + s2] / di;
            Ui[(size_t) s2 * S + i] = V[i * LD + s2] * di;
        }
    for (int i = tid; i < S; i += NTHREADS) lam[i] = A[i * LD + i];
    double* Vout = job.out + (size_t) 2 * S * S + S;
    for (int i = ty; i < S; i += TY)
        for (int s2 = tx; s2 < S; s2 += 32) Vout[(size_t) i * S + s2] = V[i * LD + s2];
}

// ---------------------------------------------------------------------------------------------
// Root / edge integration (Likelihood_*; BEAGLE calculateRoot/EdgeLogLikelihoods semantics,
// SURVEY Appendix B).  One thread per pattern:
//   L_n = sum_k w_nk sum_i pi_ni parent_n[k,c,i] * (sum_j P_nk[i,j] child_n[k,c,j])    (edge)
//   lnL_c = log( sum_n L_n 2^(e_n - emax) ) + emax ln2,  e_n = cumulative exponent of subset n
//   site[c] = lnL_c ; wsite[block] = sum over the block's 64 patterns of weight_c * lnL_c (host adds the blocks)
// ---------------------------------------------------------------------------------------------
#define MBAMD_MAX_SUBSETS 8
struct IntegrateArgs {
    const float*   parent[MBAMD_MAX_SUBSETS];
    const void*    child[MBAMD_MAX_SUBSETS];      // nullptr: root integration (no child / matrix)
    const float*   matrix[MBAMD_MAX_SUBSETS];
    const double*  weights[MBAMD_MAX_SUBSETS];    // K category weights
    const double*  freqs[MBAMD_MAX_SUBSETS];      // S state frequencies
    const int32_t* cum[MBAMD_MAX_SUBSETS];        // cumulative exponents or nullptr
    uint8_t        child_kind[MBAMD_MAX_SUBSETS];
    int            count;
};

__device__ __forceinline__ float part_at(const float* p, int S, int K, int k, int c, int i) { return p[gen_index(K, S, k, i, c)]; }
__device__ __forceinline__ float mat_at(const float* m, int SP, int k, int i, int j) { return m[(size_t) k * SP * SP + (size_t) j * SP + i]; }

__global__ void __launch_bounds__(64)
k_integrate_lnl(IntegrateArgs a, int S, int SP, int K, int P, int Ppad,
                const double* __restrict__ pattern_weights, double* __restrict__ site, double* __restrict__ wsite)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    double wl = 0.0;
    if (c < P) {
    int emax = -2147483647;
    for (int n = 0; n < a.count; ++n) {
        const int e = a.cum[n] ? a.cum[n][c] : 0;
        emax = e > emax ? e : emax;
    }
    double total = 0.0;
    for (int n = 0; n < a.count; ++n) {
        double like = 0.0;
        for (int k = 0; k < K; ++k) {
            double cat = 0.0;
            if (a.child[n] == nullptr) {
                for (int i = 0; i < S; ++i) cat += (double) part_at(a.parent[n], S, K, k, c, i) * a.freqs[n][i];
            } else if (a.child_kind[n] == CHILD_STATES) {
                const unsigned s = reinterpret_cast<const uint8_t*>(a.child[n])[c];
                for (int i = 0; i < S; ++i) {
                    const float pc = (s >= (unsigned) S) ? 1.0f : mat_at(a.matrix[n], SP, k, i, (int) s);
                    cat += (double) (part_at(a.parent[n], S, K, k, c, i) * pc) * a.freqs[n][i];
                }
            } else {
                const float* ch = reinterpret_cast<const float*>(a.child[n]);
                for (int i = 0; i < S; ++i) {
                    float acc = 0.0f;
                    for (int j = 0; j < S; ++j)
                        acc = fmaf(mat_at(a.matrix[n], SP, k, i, j), part_at(ch, S, K, k, c, j), acc);
                    cat += (double) (part_at(a.parent[n], S, K, k, c, i) * acc) * a.freqs[n][i];
                }
            }
            like += cat * a.weights[n][k];
        }
        const int e = a.cum[n] ? a.cum[n][c] : 0;
        total += ldexp(like, e - emax);
    }
    const double lnl = log(total) + (double) emax * 0.69314718055994530942;
    site[c] = lnl;
    wl = lnl * pattern_weights[c];
    } else if (c < Ppad) {
        site[c] = 0.0;
    }
    // one partial sum per 64-pattern block, reduced in a fixed order (deterministic); `wsite` is an
    // array of P_pad/64 doubles that may live in pinned host memory: the host adds them up
    mbd_wave_sum_store(wl, wsite + blockIdx.x);
}

// 4-state root / edge integration (Likelihood_NUC4, src/likelihood.c:6238-6366; edge form as above).  One thread per
// pattern; the cumulative exponent is per (pattern, category):
//   lnL_c = log( sum_n sum_k w_nk 2^(E_nkc - Emax_c) sum_i pi_ni parent_n[k,c,i] f_n[k,c,i] ) + Emax_c ln2
// with f = 1 (root form), the matrix column of the tip's state mask (compact tip: sum over compatible states), or the
// matrix-vector product with the child's partials.
struct IntegrateArgs4 {
    const f4*      parent[MBAMD_MAX_SUBSETS];
    const void*    child[MBAMD_MAX_SUBSETS];      // nullptr: root integration
    const float*   matrix[MBAMD_MAX_SUBSETS];     // [K][4][4] transposed
    const double*  weights[MBAMD_MAX_SUBSETS];
    const double*  freqs[MBAMD_MAX_SUBSETS];
    const int32_t* cum[MBAMD_MAX_SUBSETS];        // wide cumulative exponents [K][Ppad] or nullptr
    uint8_t        child_kind[MBAMD_MAX_SUBSETS];
    int            count;
};
__global__ void __launch_bounds__(64)
k_integrate_lnl_s4(IntegrateArgs4 a, int K, int P, int Ppad, BlockGeom g,
                   const double* __restrict__ pattern_weights, double* __restrict__ site, double* __restrict__ wsite)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    const size_t pb = (size_t) blockIdx.x * g.pstride + threadIdx.x;           // f4 index of (block, category 0, lane)
    double wl = 0.0;
    if (c < P) {
        // (the per-thread loads of up to eight categories -- the parent's values and the cumulative exponents -- are issued together:
        //  as loops over k they were 2 K dependent round trips in a kernel that is nothing but latency; the sums keep their order)
        int emax = -2147483647;
        for (int n = 0; n < a.count; ++n)
            for (int k0 = 0; k0 < K; k0 += 8) {
                int e8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) e8[q] = (k0 + q < K && a.cum[n]) ? a.cum[n][(size_t) (k0 + q) * Ppad + c] : (k0 + q < K ? 0 : -2147483647);
#pragma unroll
                for (int q = 0; q < 8; ++q) emax = e8[q] > emax ? e8[q] : emax;
            }
        double total = 0.0;
        for (int n = 0; n < a.count; ++n) {
            const double* __restrict__ pi = a.freqs[n];
            unsigned mask = 0;
            if (a.child[n] != nullptr && a.child_kind[n] == CHILD_STATES) {     // state bitplanes of this block
                const uint64_t* planes = reinterpret_cast<const uint64_t*>(a.child[n]) + (size_t) blockIdx.x * g.tstride;
                for (int i = 0; i < 4; ++i) mask |= (unsigned) (planes[i] >> threadIdx.x & 1u) << i;
            }
            f4 p8[8];
            int ec8[8];
            for (int k = 0; k < K; ++k) {
                if ((k & 7) == 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (k + q >= K) break;
                        p8[q] = a.parent[n][pb + (size_t) (k + q) * 64];
                        ec8[q] = a.cum[n] ? a.cum[n][(size_t) (k + q) * Ppad + c] : 0;
                    }
                }
                f4 p = p8[0];
                int e = ec8[0];
#pragma unroll
                for (int q = 1; q < 8; ++q) if ((k & 7) == q) { p = p8[q]; e = ec8[q]; }
                float f[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (a.child[n] != nullptr) {
                    const float* __restrict__ mT = a.matrix[n] + k * 16;                   // mT[j][i] = P(i->j)
                    float v[4];
                    if (a.child_kind[n] == CHILD_STATES) {
                        for (int j = 0; j < 4; ++j) v[j] = (mask >> j & 1u) ? 1.0f : 0.0f;
                    } else {
                        const f4 q = reinterpret_cast<const f4*>(a.child[n])[pb + (size_t) k * 64];
                        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                    }
                    for (int i = 0; i < 4; ++i)
                        f[i] = fmaf(mT[12 + i], v[3], fmaf(mT[8 + i], v[2], fmaf(mT[4 + i], v[1], mT[i] * v[0])));
                }
                const double cat = (double) (p.x * f[0]) * pi[0] + (double) (p.y * f[1]) * pi[1] + (double) (p.z * f[2]) * pi[2] +
                                   (double) (p.w * f[3]) * pi[3];
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

// 20/61-state tree-walk layout (mbamd_walkg.h): partials float [tile][buffer][K][T][64], tip states uint8 [tile][buffer][32],
// cumulative exponents per (pattern, category) like the 4-state path.  Same arithmetic as k_integrate_lnl, the categories
// recombined as in k_integrate_lnl_s4: k_integrate_lnl_wg_wide (mbamd_integrate_wg.h).
struct WgGeom { unsigned long tileFloats; unsigned tipTileBytes; int TP; };
#include "mbamd_integrate_wg.h"       // k_integrate_lnl_wg_wide (eight threads per pattern)
// ---------------------------------------------------------------------------------------------
// cumulative scale-factor bookkeeping (exact integer arithmetic)
// ---------------------------------------------------------------------------------------------
// (round 5: the list of sources sits in pinned host memory -- read element by element inside every thread's loop it was a round
//  trip over the host link per source, 33 us for the 99 interior nodes of a 100-taxon tree; a block now copies it into LDS in one
//  parallel read and loops over the copy)
#define MBAMD_ACC_LIST 256
__global__ void __launch_bounds__(256)
k_scale_accumulate(const int32_t* const* __restrict__ src, int count, int sign, int n, int32_t* __restrict__ cum, int fresh)
{
    __shared__ const int32_t* list[MBAMD_ACC_LIST];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = c < n;
    int acc = 0;
    for (int base = 0; base < count; base += MBAMD_ACC_LIST) {
        const int m = count - base < MBAMD_ACC_LIST ? count - base : MBAMD_ACC_LIST;
        if (base > 0) MBAMD_SYNC();
        if ((int) threadIdx.x < m) list[threadIdx.x] = src[base + (int) threadIdx.x];
        MBAMD_SYNC();
        if (live)
            for (int i = 0; i < m; ++i) acc += list[i][c];
    }
    if (live) cum[c] = (fresh ? 0 : cum[c]) + sign * acc;      // fresh: the buffer was reset just before (beagleResetScaleFactors + Accumulate = one launch)
}

// dst[c] = src ? src[c] : 0 over one scale buffer
__global__ void __launch_bounds__(256)
k_scale_copy(const int32_t* __restrict__ src, int n, int32_t* __restrict__ dst)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    dst[c] = src ? src[c] : 0;
}

// ---- 4-state path: node exponents int8 [block][scale buffer][K][64] (arena), cumulative int32 [K][Ppad] -------------
// element (k, c) of node-exponent buffer `idx`
__host__ __device__ inline size_t exp_index(unsigned estride, int K, int idx, int k, int c)
{
    return (size_t) (c >> 6) * estride + ((size_t) idx * K + k) * 64 + (c & 63);
}
struct ExpSource { const int32_t* wide; int narrow; int pad_; };      // a cumulative (wide) buffer, or arena buffer index `narrow`
// (the host lists the arena buffers -- `narrow` -- in front of the wide ones: the first nNarrow entries are summed in a loop without a
//  branch, eight loads in flight)
__global__ void __launch_bounds__(256)
k_exp_accumulate(const ExpSource* __restrict__ src, int count, int nNarrow, int sign, int K, int Ppad, const int8_t* __restrict__ arena,
                 unsigned estride, int32_t* __restrict__ cum, int fresh)
{
    __shared__ ExpSource list[MBAMD_ACC_LIST];       // (the list through LDS: see k_scale_accumulate)
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g < K * Ppad;
    const int k = live ? g / Ppad : 0, c = live ? g % Ppad : 0;
    const int8_t* const mine = arena + exp_index(estride, K, 0, k, c);      // this (pattern, category) of buffer 0
    const size_t perBuffer = exp_index(estride, K, 1, k, c) - exp_index(estride, K, 0, k, c);
    int acc = 0;
    for (int base = 0; base < count; base += MBAMD_ACC_LIST) {
        const int m = count - base < MBAMD_ACC_LIST ? count - base : MBAMD_ACC_LIST;
        if (base > 0) MBAMD_SYNC();
        if ((int) threadIdx.x < m) list[threadIdx.x] = src[base + (int) threadIdx.x];
        MBAMD_SYNC();
        if (!live) continue;
        const int nn = nNarrow - base < 0 ? 0 : (nNarrow - base < m ? nNarrow - base : m);      // narrow entries of this piece
        int i = 0;
        for (; i + 8 <= nn; i += 8) {
            int v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = mine[(size_t) list[i + u].narrow * perBuffer];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; i < nn; ++i) acc += mine[(size_t) list[i].narrow * perBuffer];
        for (; i < m; ++i) acc += list[i].wide[g];
    }
    if (live) cum[g] = (fresh ? 0 : cum[g]) + sign * acc;
}
// narrow -> wide (a node buffer that is then used as a cumulative one) and narrow -> narrow copies (src < 0: zero fill)
__global__ void __launch_bounds__(256)
k_exp_widen(const int8_t* __restrict__ arena, unsigned estride, int idx, int K, int Ppad, int32_t* __restrict__ wide)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= K * Ppad) return;
    wide[g] = arena[exp_index(estride, K, idx, g / Ppad, g % Ppad)];
}
__global__ void __launch_bounds__(256)
k_exp_copy(int8_t* __restrict__ arena, unsigned estride, int src, int dst, int K, int Ppad)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= K * Ppad) return;
    arena[exp_index(estride, K, dst, g / Ppad, g % Ppad)] = src >= 0 ? arena[exp_index(estride, K, src, g / Ppad, g % Ppad)] : (int8_t) 0;
}

// ---------------------------------------------------------------------------------------------
// layout conversion between the boundary's [category][pattern][state] doubles and device layouts
// ---------------------------------------------------------------------------------------------
// LAYOUT: 0 general tile-major buffer, 1 4-state arena (pstride = f4 elements between blocks), 2 20/61-state tree-walk
// arena (pstride = floats between tiles; `out` / `in` = the buffer's block in tile 0)
__host__ __device__ inline size_t wg_index(int S, size_t tileFloats, int k, int i, int c)
{
    return (size_t) (c / MBAMD_WG_TW) * tileFloats + (size_t) k * wg_pairs_padded(S) * 64 + wg_elem(S, i, c % MBAMD_WG_TW);
}
template <int LAYOUT>
__global__ void __launch_bounds__(256)
k_import_partials(const double* __restrict__ in, int in_has_categories, int S, int K, int P, int Ppad, size_t pstride,
                  float* __restrict__ out)
{
    const size_t total = (size_t) K * P * S;
    const size_t g = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int i = (int) (g % S);
    const int c = (int) ((g / S) % P);
    const int k = (int) (g / ((size_t) S * P));
    const double v = in_has_categories ? in[g] : in[(size_t) c * S + i];
    if (LAYOUT == 1) out[(blk_index(c, pstride) + (size_t) k * 64) * 4 + i] = (float) v;
    else if (LAYOUT == 2) out[wg_index(S, pstride, k, i, c)] = (float) v;
    else out[gen_index(K, S, k, i, c)] = (float) v;
}

template <int LAYOUT>
__global__ void __launch_bounds__(256)
k_export_partials(const float* __restrict__ in, int S, int K, int P, int Ppad, size_t pstride, double* __restrict__ out)
{
    const size_t total = (size_t) K * P * S;
    const size_t g = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int i = (int) (g % S);
    const int c = (int) ((g / S) % P);
    const int k = (int) (g / ((size_t) S * P));
    out[g] = LAYOUT == 1 ? (double) in[(blk_index(c, pstride) + (size_t) k * 64) * 4 + i]
           : LAYOUT == 2 ? (double) in[wg_index(S, pstride, k, i, c)] : (double) in[gen_index(K, S, k, i, c)];
}

}  // namespace mbamd
#include "mbamd_matrices_mfma.h"      // k_transition_matrices_mfma (9 .. 64 states: the fp64 matrix cores)
#endif
