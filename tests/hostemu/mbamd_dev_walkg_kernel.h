// mbamd_dev_walkg_kernel.h -- TEST ONLY (tests/hostemu): the device primitives of the 20/61-state tree-walk kernels
// (mrbayes_amd/csrc/device/mbamd_dev_walkg_kernel.h) on the fibers of hip_emu.h, so that the host-emulation build compiles and runs
// the PRODUCT's kernels (csrc/mbamd_walkg_kernel.h, csrc/mbamd_pathg_kernel.h) as they are: tile loops, table layouts, register
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
// v_mfma_f32_32x32x16_bf16: lane l = 32 h + m gives A[m][8 h + j] and B[8 h + j][m] (eight bf16 each); products of two bf16 are exact
// in fp32; the sixteen of an element are added in double and the sum to the accumulator with ONE rounding (the hardware's internal
// order and width are not documented -- the tests compare with a tolerance; kernels that must agree bit for bit use this same primitive
// in the same order).
typedef float mbd_f4 __attribute__((ext_vector_type(4)));
inline mbd_acc16 mbd_mfma_bf16_32x32x16(mbd_f4 a, mbd_f4 b, mbd_acc16 c)
{
    uint32_t A[64][4], B[64][4];
    for (int half = 0; half < 2; ++half) {           // (an exchange carries 64 bits of each operand per lane; its areas are reused: copy)
        // (through float temporaries: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this clang)
        const float a0 = a[2 * half], a1 = a[2 * half + 1], b0 = b[2 * half], b1 = b[2 * half + 1];
        const uint64_t ua = (uint64_t) __builtin_bit_cast(uint32_t, a0) | (uint64_t) __builtin_bit_cast(uint32_t, a1) << 32;
        const uint64_t ub = (uint64_t) __builtin_bit_cast(uint32_t, b0) | (uint64_t) __builtin_bit_cast(uint32_t, b1) << 32;
        const EmuExchange x = mbamd_emu_exchange(ua, ub);
        for (unsigned l = 0; l < 64; ++l) {
            A[l][2 * half] = (uint32_t) x.a[l]; A[l][2 * half + 1] = (uint32_t) (x.a[l] >> 32);
            B[l][2 * half] = (uint32_t) x.b[l]; B[l][2 * half + 1] = (uint32_t) (x.b[l] >> 32);
        }
    }
    auto bf = [](const uint32_t (&v)[4], unsigned j) { return (double) __builtin_bit_cast(float, (uint32_t) ((v[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) << 16); };
    const unsigned lane = threadIdx.x & 63u, n = lane & 31u, h = lane >> 5;
    for (unsigned r = 0; r < 16; ++r) {
        const unsigned i = (r & 3u) + 8u * (r >> 2) + 4u * h;
        double sum = 0.0;
        for (unsigned hh = 0; hh < 2; ++hh)
            for (unsigned j = 0; j < 8; ++j) sum += bf(A[32u * hh + i], j) * bf(B[32u * hh + n], j);
        c[r] = (float) ((double) c[r] + sum);
    }
    return c;
}
inline unsigned mbd_cvt_pk_bf16(float lo, float hi) { return (unsigned) wg_bf16_rne(lo) | (unsigned) wg_bf16_rne(hi) << 16; }
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
#define MBD_OPAQUE_VGPR(x) ((void) 0)
#define MBD_SPIN_PAUSE() mbamd_emu_yield()
struct mbd_buf { const char* p; unsigned bytes; };
inline mbd_buf mbd_make_buffer(const void* p, unsigned bytes) { return mbd_buf{static_cast<const char*>(p), bytes}; }
inline mbd_f4 mbd_buffer_load_f4(mbd_buf b, unsigned lane_offset, unsigned wave_offset)
{
    const unsigned long long off = (unsigned long long) lane_offset + wave_offset;
    mbd_f4 v = (mbd_f4) (0.0f);
    if (off + 16 <= b.bytes) std::memcpy(&v, b.p + off, 16);
    return v;
}
#define MBD_OUTSIDE 0x80000000u
inline int mbd_uniform(int v) { return v; }
}  // namespace mbamd
#endif
