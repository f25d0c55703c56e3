// mbamd_dev_base.h -- TEST ONLY (tests/hostemu): plain-C++ twins of the device primitives of
// mrbayes_amd/csrc/device/mbamd_dev_base.h, found first on the include path of the host-emulation build.  Never part of the product.
#ifndef MBAMD_DEV_BASE_H_
#define MBAMD_DEV_BASE_H_
#include <cmath>
#define MBAMD_AS_GLOBAL
#define MBAMD_AS_CONST
#define MBAMD_SYNC() mbamd_emu_barrier()
#define MBAMD_STORE_NT(value, pointer) (*(pointer) = (value))
#define MBAMD_WAVE_SYNC() mbamd_emu_wave_sync()    // (threads are fibers here: all lanes of the wave have arrived; wave-uniform control flow only)
#define MBAMD_CONSUME1(a) ((void) (a))
#define MBAMD_CONSUME4(a, b, c, d) ((void) (a), (void) (b), (void) (c), (void) (d))
#define MBAMD_IMPL_NAME "mbamd HOST EMULATION (test only)"
namespace mbamd {
typedef float f4 __attribute__((ext_vector_type(4)));     // (the host compiler is clang: the product's own vector type)
inline int mbd_frexp_exp(float v) { int e = 0; (void) frexpf(v, &e); return e; }
inline float mbd_ldexp(float v, int e) { return ldexpf(v, e); }
inline float mbd_pow2(int e) { return ldexpf(1.0f, e); }
inline int mbd_wave_index() { return (int) (threadIdx.x >> 6); }
template <int PW> inline double mbd_max_across_groups(double v)      // (threads are fibers: through a buffer, between two barriers)
{
    static double buf[1024];
    const int t = (int) threadIdx.x, base = t & ~63;
    buf[t] = v;
    mbamd_emu_barrier();
    double m = v;
    for (int l = (t & 63) % PW; l < 64; l += PW) m = fmax(m, buf[base + l]);
    mbamd_emu_barrier();
    return m;
}
typedef double f64x4 __attribute__((ext_vector_type(4)));
// v_mfma_f64_16x16x4_f64 (a wave-wide exchange of the operands, then this lane's four elements; k = 0 first)
inline f64x4 mbd_mfma_f64_16x16x4(double a, double b, f64x4 c)
{
    const EmuExchange x = mbamd_emu_exchange(__builtin_bit_cast(uint64_t, a), __builtin_bit_cast(uint64_t, b));
    const unsigned lane = threadIdx.x & 63u, j = lane & 15u;
    for (unsigned r = 0; r < 4; ++r) {
        const unsigned i = (lane >> 4) + 4u * r;
        for (unsigned k = 0; k < 4; ++k) c[r] = fma(__builtin_bit_cast(double, x.a[i + 16u * k]), __builtin_bit_cast(double, x.b[j + 16u * k]), c[r]);
    }
    return c;
}
#define MBD_SCHED_BARRIER() ((void) 0)
inline float mbd_shfl_xor(float v, int d)
{
    const EmuExchange x = mbamd_emu_exchange(__builtin_bit_cast(uint32_t, v), 0u);
    return __builtin_bit_cast(float, (uint32_t) x.a[(threadIdx.x & 63u) ^ (unsigned) d]);
}
inline double mbd_shfl_xor(double v, int d)
{
    const EmuExchange x = mbamd_emu_exchange(__builtin_bit_cast(uint64_t, v), 0u);
    return __builtin_bit_cast(double, x.a[(threadIdx.x & 63u) ^ (unsigned) d]);
}
inline long long mbd_clock() { return 0; }
inline double mbd_shfl_down_32(double v, int off)                  // (a wave-wide exchange: every live lane of the wave must call it)
{
    const EmuExchange x = mbamd_emu_exchange(__builtin_bit_cast(uint64_t, v), 0u);
    const unsigned lane = threadIdx.x & 63u, src = lane + (unsigned) off;
    return (src >> 5) == (lane >> 5) ? __builtin_bit_cast(double, x.a[src]) : v;
}
template <class T> inline T* mbd_dyn_lds() { return reinterpret_cast<T*>(mbamd_emu_dyn_lds()); }
// (threads of a block run one after the other between barriers: thread 0 comes first)
// (as fibers the lanes of a wave reach this point in any order -- whoever completed the last collective runs on first: the lanes'
//  values meet in a wave-wide exchange and the first thread adds them up in lane order)
inline void mbd_wave_sum_store(double v, double* slot)
{
    if (emu_fibers().current < 0) {
        if (threadIdx.x == 0) *slot = 0.0;
        *slot += v;
        return;
    }
    uint64_t bits;
    std::memcpy(&bits, &v, sizeof bits);
    const EmuExchange x = mbamd_emu_exchange(bits, 0);
    if (threadIdx.x != 0) return;
    double total = 0.0;
    const unsigned n = blockDim.x < 64u ? blockDim.x : 64u;
    for (unsigned l = 0; l < n; ++l) { double t; std::memcpy(&t, &x.a[l], sizeof t); total += t; }
    *slot = total;
}
inline void mbd_block_sum2_256(double off, double diag, double* red, int tid, double& o4, double& d4)
{
    red[tid] = off;
    mbamd_emu_barrier();
    if (tid == 0) { double t = 0.0; for (int u = 0; u < 256; ++u) t += red[u]; red[257] = t; }
    mbamd_emu_barrier();
    red[tid] = diag;
    mbamd_emu_barrier();
    o4 = d4 = 0.0;
    if (tid == 0) {
        for (int u = 0; u < 256; ++u) d4 += red[u];
        o4 = red[257];
    }
}
template <int NT> inline void mbd_block_sum2(double off, double diag, double* red, int tid, double& o4, double& d4)
{
    static double a[2048], b[2048];                  // (not the kernel's scratch: thread 0 sums when every thread has written)
    a[tid] = off; b[tid] = diag;
    mbamd_emu_barrier();
    o4 = d4 = 0.0;
    if (tid == 0) for (int u = 0; u < NT; ++u) { o4 += a[u]; d4 += b[u]; }
    mbamd_emu_barrier();
    (void) red;
}
inline void mbd_jacobi_rotation(double app, double aqq, double apq, double& c, double& t)
{
    const double tau = (aqq - app) / (2.0 * apq);
    t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    c = 1.0 / sqrt(1.0 + t * t);
}
}  // namespace mbamd
#endif
