// mbamd_dev_base.h (gfx950) -- the handful of device primitives the kernels are written against: address spaces, the native
// f4, exponent intrinsics, dynamic LDS, wave / block sums.  Included as <mbamd_dev_base.h>: the product is compiled with
// -I csrc/device; the TEST-ONLY host emulation puts tests/hostemu in front, whose header of the same name implements the same
// primitives in plain C++ -- so the kernel sources themselves carry no second implementation and no preprocessor fork.
#ifndef MBAMD_DEV_BASE_H_
#define MBAMD_DEV_BASE_H_

// Pointers that reach a kernel through the operation table are generic ("flat") as far as the
// compiler knows.  Casting them to the global address space turns flat_load/flat_store into
// global_load/global_store, and casting wave-uniform read-only data (operation table, transition
// matrices) to the constant address space lets the compiler fetch it with scalar loads (s_load_*)
// into SGPRs, where it feeds v_fma as a scalar operand for all 64 lanes at once.
#define MBAMD_AS_GLOBAL __attribute__((address_space(1)))
#define MBAMD_AS_CONST __attribute__((address_space(4)))
#define MBAMD_SYNC() __syncthreads()
// a store that does not allocate in L2 (a result stream written once: with write-allocate it evicts matrices and programs)
#define MBAMD_STORE_NT(value, pointer) __builtin_nontemporal_store(value, pointer)
// the lanes of a wave exchange data through LDS: LDS instructions of a wave execute in order, so the hardware needs nothing --
// this only keeps the compiler from moving memory accesses across the point
#define MBAMD_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// a use the compiler cannot move or drop: whatever wait the values need is placed here (a load consumed in the branch that issued it)
#define MBAMD_CONSUME1(a) asm volatile("" : "+v"(a))
#define MBAMD_CONSUME4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define MBAMD_IMPL_NAME "mbamd HIP gfx950"
namespace mbamd {
// a native clang vector (not HIP's f4 class) so that it can be loaded/stored through
// address-space qualified pointers as one dwordx4 access
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int mbd_frexp_exp(float v) { return __builtin_amdgcn_frexp_expf(v); }
__device__ __forceinline__ float mbd_ldexp(float v, int e) { return __builtin_amdgcn_ldexpf(v, e); }
// 2^e as a float for -126 <= e <= 127 (the exponent field written directly): multiplying by it is the exact scaling ldexp does,
// one v_pk_mul_f32 per two values instead of two v_ldexp_f32
__device__ __forceinline__ float mbd_pow2(int e) { return __builtin_bit_cast(float, (unsigned) (127 + e) << 23); }
__device__ __forceinline__ int mbd_wave_index() { return __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)); }
// the workgroup's dynamic LDS (one symbol per translation unit: every kernel sees the same array)
// maximum over the lanes l, l ^ PW, l ^ 2 PW, ... of a wave (lane = group * PW + member: the same member of every group)
template <int PW> __device__ __forceinline__ double mbd_max_across_groups(double v)
{
#pragma unroll
    for (int d = PW; d < 64; d <<= 1) v = fmax(v, __shfl_xor(v, d));
    return v;
}
template <class T> __device__ __forceinline__ T* mbd_dyn_lds()
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mbd_lds_bytes[];
    return reinterpret_cast<T*>(mbd_lds_bytes);
}
// D (16 x 16) += A (16 x 4) B (4 x 16) in fp64: lane l gives A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds column l & 15 of D,
// register r = row (l >> 4) + 4 r
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f64x4 mbd_mfma_f64_16x16x4(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
#define MBD_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ float mbd_shfl_xor(float v, int d) { return __shfl_xor(v, d); }         // the value lane l ^ d holds
__device__ __forceinline__ double mbd_shfl_xor(double v, int d) { return __shfl_xor(v, d); }
__device__ __forceinline__ long long mbd_clock() { return (long long) __builtin_amdgcn_s_memtime(); }   // (timing experiments)
// the value lane l + off of this lane's group of 32 holds (its own beyond the group's end)
__device__ __forceinline__ double mbd_shfl_down_32(double v, int off) { return __shfl_down(v, off, 32); }
// sum of `v` over the 64 lanes of the (only) wave of a 64-thread block, stored by lane 0: fixed order, deterministic
__device__ __forceinline__ void mbd_wave_sum_store(double v, double* slot)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (threadIdx.x == 0) *slot = v;
}
// sums of `off` and `diag` over the 256 threads of a block (valid in thread 0); `red`: 264 doubles of LDS scratch
__device__ __forceinline__ void mbd_block_sum2_256(double off, double diag, double* red, int tid, double& o4, double& d4)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_down(off, o); diag += __shfl_down(diag, o); }
    if ((tid & 63) == 0) { red[tid >> 6] = off; red[8 + (tid >> 6)] = diag; }
    __syncthreads();
    o4 = (red[0] + red[1]) + (red[2] + red[3]);
    d4 = (red[8] + red[9]) + (red[10] + red[11]);
}
// the same for a block of NT threads (NT / 64 waves); `red`: NT + 8 doubles of LDS scratch, of which the first NT / 32 are used here
template <int NT> __device__ __forceinline__ void mbd_block_sum2(double off, double diag, double* red, int tid, double& o4, double& d4)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_down(off, o); diag += __shfl_down(diag, o); }
    constexpr int NW = NT / 64;
    if ((tid & 63) == 0) { red[tid >> 6] = off; red[NW + (tid >> 6)] = diag; }
    __syncthreads();
    o4 = d4 = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { o4 += red[w]; d4 += red[NW + w]; }
}
// Jacobi rotation (cosine c, tangent t) that annihilates a_pq.  The ANGLE may be approximate (a Jacobi iteration corrects
// itself), the rotation must be orthogonal: hardware reciprocal / square root for tau and t, Newton steps on the reciprocal
// square root that normalises (c, s).  (The correctly rounded divisions and roots were the longest part of a step, on one
// wave, before a barrier.)
__device__ __forceinline__ void mbd_jacobi_rotation(double app, double aqq, double apq, double& c, double& t)
{
    const double tau = (aqq - app) * 0.5 * __builtin_amdgcn_rcp(apq);
    const double at = fabs(tau);
    t = at < 1e150 ? __builtin_amdgcn_rcp(at + __builtin_amdgcn_sqrt(1.0 + at * at)) : 0.0;
    t = tau >= 0.0 ? t : -t;
    const double w = 1.0 + t * t;
    const double c0 = __builtin_amdgcn_rsq(w);
    c = c0 * (1.5 - 0.5 * w * c0 * c0);
    c = c * (1.5 - 0.5 * w * c * c);
}
}  // namespace mbamd
#endif
