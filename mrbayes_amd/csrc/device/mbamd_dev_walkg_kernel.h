// mbamd_dev_walkg_kernel.h (gfx950) -- the device primitives of the 20/61-state tree-walk kernels (csrc/mbamd_walkg_kernel.h,
// csrc/mbamd_pathg_kernel.h): the matrix-core instruction, the lane swap, waits, barriers, where an inline program sits.  The
// TEST-ONLY host emulation has a header of the same name in front on its include path (tests/hostemu/) that implements the same
// primitives on fibers -- an MFMA there is a wave-wide exchange and 32 multiply-adds per lane --, so the kernels themselves, with
// their tile loops, table layouts, register sets and LDS slots, are the code the CPU CI runs.
#ifndef MBAMD_DEV_WALKG_KERNEL_H_
#define MBAMD_DEV_WALKG_KERNEL_H_
namespace mbamd {
__device__ __forceinline__ const Walk4Entry* wg_program(const WalkGArgsInline&)
{
    return reinterpret_cast<const Walk4Entry*>((uintptr_t) __builtin_amdgcn_kernarg_segment_ptr() + offsetof(WalkGArgsInline, inl));
}
typedef float mbd_acc16 __attribute__((ext_vector_type(16)));
// D (32 x 32) += A (32 x 2) B (2 x 32): lane l gives A[l & 31][l >> 5] and B[l >> 5][l & 31] and holds column l & 31 of D, register r =
// row (r & 3) + 8 (r >> 2) + 4 (l >> 5)
__device__ __forceinline__ mbd_acc16 mbd_mfma_f32_32x32x2(float a, float b, mbd_acc16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// D (32 x 32) += A (32 x 16) B (16 x 32) with bf16 operands, exact products, fp32 accumulation (v_mfma_f32_32x32x16_bf16, 32 cycles on
// a pipe of its own: the vector ALU keeps issuing beside it, profiles/r06_bf16x3.txt).  Lane l = 32 h + m gives the eight bf16
// A[m][8 h + j] and B[8 h + j][m], j = 0..7 (element j in bits 16 (j & 1) of dword j >> 1); D as above.
typedef float mbd_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mbd_acc16 mbd_mfma_bf16_32x32x16(mbd_f4 a, mbd_f4 b, mbd_acc16 c)
{
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the bf16 nearest to lo (bits 0..15) and to hi (bits 16..31), ties to even
__device__ __forceinline__ unsigned mbd_cvt_pk_bf16(float lo, float hi)
{
    // (v_cvt_pk_bf16_f32 through the compiler, not inline assembly: it pads the VALU -> MFMA-operand wait states only for
    //  instructions it knows; the first version, an asm statement, fed stale pieces to the matrix core now and then)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// max(v, the value lane l ^ 32 holds) (v_permlane32_swap: in the VALU, no LDS round trip)
__device__ __forceinline__ float mbd_max_lane_xor32(float v)
{
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}
// LDS as the compiler must see it for a value another wave changes (address-space inference leaves volatile accesses flat)
#define MBAMD_AS_LDS __attribute__((address_space(3)))
#define MBD_DRAIN_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")      // this wave's stores and loads have completed
#define MBD_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MBD_WG_BARRIER() __builtin_amdgcn_s_barrier()
#define MBD_COMPILER_FENCE() asm volatile("" ::: "memory")
#define MBD_PIN_VGPR(x) asm volatile("" :: "v"(x) : "memory")                            // the value is in its register HERE
#define MBD_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))                                       // the value comes out of HERE: not a load the optimiser may move or merge
#define MBD_SPIN_PAUSE() __builtin_amdgcn_s_sleep(1)
// A bounded window on memory (a raw buffer resource): loads through it take a 32-bit per-lane offset and a wave-uniform offset, need
// no 64-bit address arithmetic in the VALU, and a lane whose offset lies OUTSIDE the window reads zeros without touching memory --
// how k_walkb keeps the operand loads a chunk does not need in its instruction sequence for free.
typedef __amdgpu_buffer_rsrc_t mbd_buf;
__device__ __forceinline__ mbd_buf mbd_make_buffer(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int) bytes, 0x00020000); }
__device__ __forceinline__ mbd_f4 mbd_buffer_load_f4(mbd_buf b, unsigned lane_offset, unsigned wave_offset)
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(mbd_f4, (u32x4) __builtin_amdgcn_raw_buffer_load_b128(b, (int) lane_offset, (int) wave_offset, 0));
}
#define MBD_OUTSIDE 0x80000000u                      // a lane offset outside every window
__device__ __forceinline__ int mbd_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
}  // namespace mbamd
#endif
