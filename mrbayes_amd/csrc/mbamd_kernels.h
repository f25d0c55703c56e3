// mbamd_kernels.h -- hand-written HIP kernels (gfx950 / CDNA4) of the conditional-likelihood engine.
//
// Replaces, on the device, the reference's native loops in src/likelihood.c:
//   k_walk_s4            CondLikeDown/Root_NUC4* + CondLikeScaler_NUC4* + RemoveNodeScalers (4-state)
//   k_partials_gen       CondLikeDown_Gen / _NY98 (general state count), k_rescale_gen = CondLikeScaler_Gen
//   k_transition_matrices TiProbs_Gen / TiProbs_GenCov (src/likelihood.c:9424-9700)
//   k_integrate_lnl      Likelihood_Gen / _NY98 / _NUC4 root integration (src/likelihood.c:5764, 6975, 6238)
//   k_scale_*            Copy/Reset/RemoveNodeScalers bookkeeping (src/likelihood.c:7981-8131)
//
// Data layout in HBM (all fp32 unless noted), P_pad = patterns rounded up to 64:
//   4-state partials   : f4 [K][P_pad]            one f4 = the 4 states of (category, pattern)
//   general partials   : float  [K][S][P_pad]         state-major: lanes = consecutive patterns -> coalesced
//   4-state matrices   : float  [K][4][4]             row = from-state (same as the reference ti[k][i][j])
//   general matrices   : float  [K][SP][SP] transposed (mT[k][j][i] = P_k(i->j)), zero padded to SP
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
// The kernels use no cross-lane operations and no __syncthreads, so the same source also builds
// against tests/hostemu/hip_emu.h for CPU-only CI of the host logic (never part of the product).
#ifndef MBAMD_KERNELS_H_
#define MBAMD_KERNELS_H_

#include <stdint.h>

// Pointers that reach a kernel through the operation table are generic ("flat") as far as the
// compiler knows.  Casting them to the global address space turns flat_load/flat_store into
// global_load/global_store, and casting wave-uniform read-only data (operation table, transition
// matrices) to the constant address space lets the compiler fetch it with scalar loads (s_load_*)
// into SGPRs, where it feeds v_fma as a scalar operand for all 64 lanes at once.
#if defined(MBAMD_HOST_EMU)
#define MBAMD_AS_GLOBAL
#define MBAMD_AS_CONST
namespace mbamd { typedef float4 f4; }
#else
#define MBAMD_AS_GLOBAL __attribute__((address_space(1)))
#define MBAMD_AS_CONST __attribute__((address_space(4)))
// a native clang vector (not HIP's f4 class) so that it can be loaded/stored through
// address-space qualified pointers as one dwordx4 access
namespace mbamd { typedef float f4 __attribute__((ext_vector_type(4))); }
#endif

namespace mbamd {

template <class T> __device__ __forceinline__ MBAMD_AS_GLOBAL T* as_global(T* p)
{
    return (MBAMD_AS_GLOBAL T*) (uintptr_t) p;
}
template <class T> __device__ __forceinline__ const MBAMD_AS_CONST T* as_const(const T* p)
{
    return (const MBAMD_AS_CONST T*) (uintptr_t) p;
}

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
    uint8_t      pad_[2];
    int32_t      pad2_[2];
};
static_assert(sizeof(PartialsOp) == 64, "PartialsOp must be 64 bytes");

#define MBAMD_NO_SLOT 0xFF

// ---------------------------------------------------------------------------------------------
// exact power-of-two rescaling helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int scale_exponent(float mx)
{
#if defined(MBAMD_HOST_EMU)
    int e = 0;
    if (mx > 0.0f && mx < 3.0e38f) (void) frexpf(mx, &e);
#else
    int e = (mx > 0.0f && mx < 3.0e38f) ? __builtin_amdgcn_frexp_expf(mx) : 0;
#endif
    e = e < -126 ? -126 : e;      // keep 2^-e a normal float even for denormal maxima
    e = e > 126 ? 126 : e;
    return e;
}
__device__ __forceinline__ float scale_pow2(float v, int neg_e)
{
#if defined(MBAMD_HOST_EMU)
    return ldexpf(v, neg_e);
#else
    return __builtin_amdgcn_ldexpf(v, neg_e);
#endif
}

__device__ __forceinline__ float max4(f4 v) { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); }

// 4x4 matrix (row-major, uniform -> scalar registers) times f4
__device__ __forceinline__ f4 mat4_mul(const MBAMD_AS_CONST float* __restrict__ m, f4 v)
{
    f4 r;
    r.x = fmaf(m[3], v.w, fmaf(m[2], v.z, fmaf(m[1], v.y, m[0] * v.x)));
    r.y = fmaf(m[7], v.w, fmaf(m[6], v.z, fmaf(m[5], v.y, m[4] * v.x)));
    r.z = fmaf(m[11], v.w, fmaf(m[10], v.z, fmaf(m[9], v.y, m[8] * v.x)));
    r.w = fmaf(m[15], v.w, fmaf(m[14], v.z, fmaf(m[13], v.y, m[12] * v.x)));
    return r;
}

// a f4 this lane itself stored earlier in the same launch: bypass the (non-coherent) vector L1
__device__ __forceinline__ f4 load_own_store(const f4* p)
{
#if defined(MBAMD_HOST_EMU)
    return *p;
#else
    const float* f = reinterpret_cast<const float*>(p);
    f4 r;
    r.x = __hip_atomic_load(f + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.y = __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.z = __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.w = __hip_atomic_load(f + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return r;
#endif
}

// ---------------------------------------------------------------------------------------------
// 4-state tree-walk kernel.
//
// One wave (= one 64-thread workgroup) owns 64 site patterns for the whole operation list: lane =
// pattern, all K categories in registers.  Site patterns are independent through the entire
// pruning recursion, so the wave walks the post-order list by itself -- no inter-workgroup
// dependency, one launch per beagleUpdatePartials instead of one per tree level.  A freshly
// computed node is written to HBM (it must persist for later partial updates) AND pushed into
// a per-wave LDS stack slot chosen by the host, so its parent reads it back from LDS: HBM sees
// each interior partial exactly once, as a streaming 1 KiB-per-instruction store.  Transition
// matrices are wave-uniform and arrive through scalar loads; per-pattern max-rescaling (the
// reference's separate CondLikeScaler pass) and the cumulative-scaler update are fused in.
// ---------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void walk_load_child(const void* ptr, int kind, int slot, int Ppad, int c, int lane,
                                                const f4* lds, f4 (&v)[K])
{
    if (kind == CHILD_LDS) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = lds[(slot * K + k) * 64 + lane];
    } else if (kind == CHILD_STATES) {
        const unsigned s = as_global(reinterpret_cast<const uint8_t*>(ptr))[c];
        f4 one;
        one.x = (s == 0u || s >= 4u) ? 1.0f : 0.0f;
        one.y = (s == 1u || s >= 4u) ? 1.0f : 0.0f;
        one.z = (s == 2u || s >= 4u) ? 1.0f : 0.0f;
        one.w = (s == 3u || s >= 4u) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = one;
    } else if (kind == CHILD_PARTIALS) {
        const MBAMD_AS_GLOBAL f4* p = as_global(reinterpret_cast<const f4*>(ptr));
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = p[(size_t) k * Ppad + c];
    } else {
        const f4* p = reinterpret_cast<const f4*>(ptr);
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = load_own_store(p + (size_t) k * Ppad + c);
    }
}

template <int K>
__global__ void __launch_bounds__(64)
k_walk_s4(const PartialsOp* __restrict__ ops, int nops, int Ppad, int32_t* __restrict__ cumulative)
{
#if defined(MBAMD_HOST_EMU)
    f4* lds = reinterpret_cast<f4*>(mbamd_emu_dyn_lds());
#else
    extern __shared__ f4 lds[];
#endif
    const int lane = threadIdx.x;
    const int c = blockIdx.x * 64 + lane;
    int cum_e = 0;

    const MBAMD_AS_CONST PartialsOp* __restrict__ cops = as_const(ops);
    for (int o = 0; o < nops; ++o) {
        const MBAMD_AS_CONST PartialsOp* __restrict__ op = cops + o;
        const int k1 = op->c1_kind, k2 = op->c2_kind;
        if (k1 == CHILD_RELOAD || k2 == CHILD_RELOAD) {
            // our own earlier stores must have reached L2 before we read them back
#if !defined(MBAMD_HOST_EMU)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        }
        f4 a[K], b[K];
        walk_load_child<K>(op->c1, k1, op->c1_slot, Ppad, c, lane, lds, a);
        walk_load_child<K>(op->c2, k2, op->c2_slot, Ppad, c, lane, lds, b);

        const MBAMD_AS_CONST float* __restrict__ m1 = as_const(op->m1);
        const MBAMD_AS_CONST float* __restrict__ m2 = as_const(op->m2);
        f4 out[K];
        float mx = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const f4 f1 = mat4_mul(m1 + 16 * k, a[k]);
            const f4 f2 = mat4_mul(m2 + 16 * k, b[k]);
            out[k].x = f1.x * f2.x;
            out[k].y = f1.y * f2.y;
            out[k].z = f1.z * f2.z;
            out[k].w = f1.w * f2.w;
            mx = fmaxf(mx, max4(out[k]));
        }

        const int mode = op->scale_mode;
        if (mode != SCALE_NONE) {
            int e;
            MBAMD_AS_GLOBAL int32_t* sc = as_global(op->scale);
            if (mode == SCALE_WRITE) {
                e = scale_exponent(mx);
                sc[c] = e;
                cum_e += e;
            } else {
                e = sc[c];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                out[k].x = scale_pow2(out[k].x, -e);
                out[k].y = scale_pow2(out[k].y, -e);
                out[k].z = scale_pow2(out[k].z, -e);
                out[k].w = scale_pow2(out[k].w, -e);
            }
        }

        MBAMD_AS_GLOBAL f4* __restrict__ dst = as_global(reinterpret_cast<f4*>(op->dst));
#pragma unroll
        for (int k = 0; k < K; ++k) dst[(size_t) k * Ppad + c] = out[k];
        const int ds = op->dst_slot;
        if (ds != MBAMD_NO_SLOT) {
#pragma unroll
            for (int k = 0; k < K; ++k) lds[(ds * K + k) * 64 + lane] = out[k];
        }
    }
    if (cumulative != nullptr && cum_e != 0) cumulative[c] += cum_e;
}

// ---------------------------------------------------------------------------------------------
// General state count: level-synchronous kernel.  grid = (P_pad/64, ops in this dependency level),
// one thread per pattern, loop over categories.  State-major layout makes every load/store a
// coalesced 256-byte wave access; the (transposed, zero-padded) transition matrix column
// mT[k][j][0..SP) is wave-uniform and is consumed as scalar operands of v_fmac.
//   FUSED_K > 0 : K == FUSED_K, all K*SP outputs stay in registers and the per-pattern rescale is
//                 fused (power-of-two, see top of file);
//   FUSED_K == 0: any K; writes unscaled partials, k_rescale_gen does the scaling pass.
// ---------------------------------------------------------------------------------------------
template <int SP>
__device__ __forceinline__ void gen_child_factor(const void* ptr, int kind, const float* mT_, int S, int k,
                                                 int Ppad, int c, float (&f)[SP])
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
            as_global(reinterpret_cast<const float*>(ptr)) + (size_t) k * S * Ppad + c;
#pragma unroll
        for (int i = 0; i < SP; ++i) f[i] = 0.0f;
#pragma unroll 2
        for (int j = 0; j < S; ++j) {
            const float vj = cl[(size_t) j * Ppad];
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
    MBAMD_AS_GLOBAL float* __restrict__ dst = as_global(op->dst) + c;
    MBAMD_AS_GLOBAL int32_t* sc = as_global(op->scale);
    const int mode = op->scale_mode;

    if constexpr (FUSED_K > 0) {
        constexpr int KK = FUSED_K > 0 ? FUSED_K : 1;
        float out[KK][SP];
        float mx = 0.0f;
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            float f2[SP];
            gen_child_factor<SP>(c1, k1, m1 + (size_t) k * SP * SP, S, k, Ppad, c, out[k]);
            gen_child_factor<SP>(c2, k2, m2 + (size_t) k * SP * SP, S, k, Ppad, c, f2);
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
                if (i < S) dst[((size_t) k * S + i) * Ppad] = (mode != SCALE_NONE) ? scale_pow2(out[k][i], -e) : out[k][i];
        }
    } else {
        for (int k = 0; k < K; ++k) {
            float f1[SP], f2[SP];
            gen_child_factor<SP>(c1, k1, m1 + (size_t) k * SP * SP, S, k, Ppad, c, f1);
            gen_child_factor<SP>(c2, k2, m2 + (size_t) k * SP * SP, S, k, Ppad, c, f2);
#pragma unroll
            for (int i = 0; i < SP; ++i)
                if (i < S) dst[((size_t) k * S + i) * Ppad] = f1[i] * f2[i];
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
    MBAMD_AS_GLOBAL float* __restrict__ dst = as_global(op->dst) + c;
    MBAMD_AS_GLOBAL int32_t* sc = as_global(op->scale);
    const int n = K * S;
    int e;
    if (mode == SCALE_WRITE) {
        float mx = 0.0f;
        for (int r = 0; r < n; ++r) mx = fmaxf(mx, dst[(size_t) r * Ppad]);
        e = scale_exponent(mx);
        sc[c] = e;
        if (cumulative != nullptr && e != 0) atomicAdd(cumulative + c, e);
    } else {
        e = sc[c];
    }
    if (e != 0)
        for (int r = 0; r < n; ++r) dst[(size_t) r * Ppad] = scale_pow2(dst[(size_t) r * Ppad], -e);
}

// ---------------------------------------------------------------------------------------------
// Transition probabilities: P_k = U diag(exp(lambda * t * rate_k)) U^-1, fp64 math -> fp32 store,
// negatives clamped to 0 (TiProbs_Gen, src/likelihood.c:9528-9545).  grid = count*K workgroups.
// eig = [U (S*S) | U^-1 (S*S) | lambda (S)] doubles.  Output layout per matrix buffer:
//   transposed == 0: out[k][i][j] (4-state path), stride SP = S
//   transposed == 1: out[k][j][i] zero-padded rows of SP floats (general path)
// ---------------------------------------------------------------------------------------------
struct MatrixJob {
    float* out;        // matrix buffer (K matrices)
    double length;     // branch length
};

// exp(lambda*t) hoisted: one thread per (job, k, s) fills ev[(job*K+k)*S + s]; the matrix kernel
// below then reads it (second launch on the same stream, so no barrier is needed).
__global__ void __launch_bounds__(256)
k_eigen_exponentials(const MatrixJob* __restrict__ jobs, const double* __restrict__ eig,
                     const double* __restrict__ rates, int S, int K, int total, double* __restrict__ ev)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int s = g % S, bk = g / S;
    const int b = bk / K, k = bk % K;
    const double* __restrict__ lam = eig + (size_t) 2 * S * S;
    ev[g] = exp(lam[s] * jobs[b].length * rates[k]);
}

__global__ void __launch_bounds__(256)
k_transition_matrices_ev(const MatrixJob* __restrict__ jobs, const double* __restrict__ eig,
                         const double* __restrict__ ev, int S, int SP, int K, int transposed)
{
    const int b = blockIdx.x / K, k = blockIdx.x % K;
    const double* __restrict__ U = eig;
    const double* __restrict__ Ui = eig + (size_t) S * S;
    const double* __restrict__ e = ev + (size_t) blockIdx.x * S;
    float* __restrict__ out = jobs[b].out + (size_t) k * SP * SP;
    for (int idx = threadIdx.x; idx < S * S; idx += blockDim.x) {
        // consecutive threads take consecutive j so that U^-1 reads are coalesced
        const int i = idx / S, j = idx % S;
        double sum = 0.0;
        for (int s = 0; s < S; ++s) sum += U[i * S + s] * e[s] * Ui[s * S + j];
        const float v = (sum < 0.0) ? 0.0f : (float) sum;
        if (transposed) out[(size_t) j * SP + i] = v;
        else            out[(size_t) i * SP + j] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Root / edge integration (Likelihood_*; BEAGLE calculateRoot/EdgeLogLikelihoods semantics,
// SURVEY Appendix B).  One thread per pattern:
//   L_n = sum_k w_nk sum_i pi_ni parent_n[k,c,i] * (sum_j P_nk[i,j] child_n[k,c,j])    (edge)
//   lnL_c = log( sum_n L_n 2^(e_n - emax) ) + emax ln2,  e_n = cumulative exponent of subset n
//   site[c] = lnL_c ; wsite[c] = weight_c * lnL_c   (summed by k_chunk_sums + host)
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

template <bool S4>
__device__ __forceinline__ float part_at(const float* p, int S, int Ppad, int k, int c, int i)
{
    return S4 ? p[((size_t) k * Ppad + c) * 4 + i] : p[((size_t) k * S + i) * Ppad + c];
}
template <bool S4>
__device__ __forceinline__ float mat_at(const float* m, int SP, int k, int i, int j)
{
    return S4 ? m[k * 16 + i * 4 + j] : m[(size_t) k * SP * SP + (size_t) j * SP + i];
}

template <bool S4>
__global__ void __launch_bounds__(64)
k_integrate_lnl(IntegrateArgs a, int S, int SP, int K, int P, int Ppad,
                const double* __restrict__ pattern_weights, double* __restrict__ site, double* __restrict__ wsite)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= P) {
        if (c < Ppad) { site[c] = 0.0; wsite[c] = 0.0; }
        return;
    }
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
                for (int i = 0; i < S; ++i) cat += (double) part_at<S4>(a.parent[n], S, Ppad, k, c, i) * a.freqs[n][i];
            } else if (a.child_kind[n] == CHILD_STATES) {
                const unsigned s = reinterpret_cast<const uint8_t*>(a.child[n])[c];
                for (int i = 0; i < S; ++i) {
                    const float pc = (s >= (unsigned) S) ? 1.0f : mat_at<S4>(a.matrix[n], SP, k, i, (int) s);
                    cat += (double) (part_at<S4>(a.parent[n], S, Ppad, k, c, i) * pc) * a.freqs[n][i];
                }
            } else {
                const float* ch = reinterpret_cast<const float*>(a.child[n]);
                for (int i = 0; i < S; ++i) {
                    float acc = 0.0f;
                    for (int j = 0; j < S; ++j)
                        acc = fmaf(mat_at<S4>(a.matrix[n], SP, k, i, j), part_at<S4>(ch, S, Ppad, k, c, j), acc);
                    cat += (double) (part_at<S4>(a.parent[n], S, Ppad, k, c, i) * acc) * a.freqs[n][i];
                }
            }
            like += cat * a.weights[n][k];
        }
        const int e = a.cum[n] ? a.cum[n][c] : 0;
        total += ldexp(like, e - emax);
    }
    const double lnl = log(total) + (double) emax * 0.69314718055994530942;
    site[c] = lnl;
    wsite[c] = lnl * pattern_weights[c];
}

// sums[t] = sum of wsite[t*chunk .. (t+1)*chunk): the few hundred partial sums go to the host,
// which adds them in a fixed order (deterministic, fp64).
__global__ void __launch_bounds__(64)
k_chunk_sums(const double* __restrict__ wsite, int n, int chunk, int nchunks, double* __restrict__ sums)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= nchunks) return;
    const int lo = t * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    double s = 0.0;
    for (int i = lo; i < hi; ++i) s += wsite[i];
    sums[t] = s;
}

// ---------------------------------------------------------------------------------------------
// cumulative scale-factor bookkeeping (exact integer arithmetic)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_scale_accumulate(const int32_t* const* __restrict__ src, int count, int sign, int n, int32_t* __restrict__ cum)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    int acc = 0;
    for (int i = 0; i < count; ++i) acc += src[i][c];
    cum[c] += sign * acc;
}

// ---------------------------------------------------------------------------------------------
// layout conversion between the boundary's [category][pattern][state] doubles and device layouts
// ---------------------------------------------------------------------------------------------
template <bool S4>
__global__ void __launch_bounds__(256)
k_import_partials(const double* __restrict__ in, int in_has_categories, int S, int K, int P, int Ppad,
                  float* __restrict__ out)
{
    const size_t total = (size_t) K * P * S;
    const size_t g = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int i = (int) (g % S);
    const int c = (int) ((g / S) % P);
    const int k = (int) (g / ((size_t) S * P));
    const double v = in_has_categories ? in[g] : in[(size_t) c * S + i];
    if (S4) out[((size_t) k * Ppad + c) * 4 + i] = (float) v;
    else    out[((size_t) k * S + i) * Ppad + c] = (float) v;
}

template <bool S4>
__global__ void __launch_bounds__(256)
k_export_partials(const float* __restrict__ in, int S, int K, int P, int Ppad, double* __restrict__ out)
{
    const size_t total = (size_t) K * P * S;
    const size_t g = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int i = (int) (g % S);
    const int c = (int) ((g / S) % P);
    const int k = (int) (g / ((size_t) S * P));
    out[g] = S4 ? (double) in[((size_t) k * Ppad + c) * 4 + i] : (double) in[((size_t) k * S + i) * Ppad + c];
}

}  // namespace mbamd
#endif
