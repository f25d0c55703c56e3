// mbamd_dev_walkg_kernel.h -- TEST ONLY (tests/hostemu): the device primitives of the 20/61-state tree-walk kernels
// (mrbayes_amd/csrc/device/mbamd_dev_walkg_kernel.h) on the fibers of hip_emu.h, so that the host-emulation build compiles and runs
// the PRODUCT's kernels (csrc/mbamd_walkg_kernel.h, csrc/mbamd_walkg2_kernel.h) as they are: tile loops, table layouts, register
// sets, LDS slots, the pair protocol.  A matrix-core instruction is a wave-wide exchange of the operands and the 32 x 32 x 2
// arithmetic per lane; waits are no-ops (memory is coherent between fibers), a spin gives the other fibers a turn.
// Never part of the product.
#ifndef MBAMD_DEV_WALKG_KERNEL_H_
#define MBAMD_DEV_WALKG_KERNEL_H_
namespace mbamd {
inline const Walk4Entry* wg_program(const WalkGArgsInline& a) { return a.inl; }
typedef float mbd_acc16 __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x2_f32: lane l gives A[l & 31][l >> 5] and B[l >> 5][l & 31]; it holds column l & 31 of D, register r = row
// (r & 3) + 8 (r >> 2) + 4 (l >> 5).  (The two products of an element are added one after the other, k = 0 first; the hardware's
// internal order is not documented -- the tests compare with a tolerance.)
inline mbd_acc16 mbd_mfma_f32_32x32x2(float a, float b, mbd_acc16 c)
{
    const EmuExchange x = mbamd_emu_exchange(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b));
    const unsigned lane = threadIdx.x & 63u, n = lane & 31u, h = lane >> 5;
    const float b0 = __builtin_bit_cast(float, (uint32_t) x.b[n]), b1 = __builtin_bit_cast(float, (uint32_t) x.b[32u + n]);
    for (unsigned r = 0; r < 16; ++r) {
        const unsigned i = (r & 3u) + 8u * (r >> 2) + 4u * h;
        c[r] = fmaf(__builtin_bit_cast(float, (uint32_t) x.a[32u + i]), b1, fmaf(__builtin_bit_cast(float, (uint32_t) x.a[i]), b0, c[r]));
    }
    return c;
}
inline float mbd_max_lane_xor32(float v)
{
    const EmuExchange x = mbamd_emu_exchange(__builtin_bit_cast(uint32_t, v), 0u);
    return fmaxf(v, __builtin_bit_cast(float, (uint32_t) x.a[(threadIdx.x & 63u) ^ 32u]));
}
#define MBAMD_AS_LDS
#define MBD_DRAIN_ALL() ((void) 0)
#define MBD_DRAIN_VMEM() ((void) 0)
#define MBD_WG_BARRIER() mbamd_emu_barrier()
#define MBD_COMPILER_FENCE() ((void) 0)
#define MBD_PIN_VGPR(x) ((void) (x))
#define MBD_SPIN_PAUSE() mbamd_emu_yield()
inline int mbd_uniform(int v) { return v; }
}  // namespace mbamd
#endif
