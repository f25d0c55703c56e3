// Store-pattern microbenchmark for the fp64 general-state partials (61 states x 5056 patterns x 99 buffers = 244 MB per pass):
//   rows16   : a wave writes 61 rows x 128 B, rows P*8 bytes apart (the [state][pattern] layout, 16-pattern MFMA tiles)  -- today's kernels
//   rows64   : a wave writes 61 rows x 512 B (lane = pattern)
//   tile16   : a wave writes 61 x 128 B contiguous (tile-major layout [tile][state][16])
//   tile16x4 : the same with 16-byte stores
// plus reads of the same shapes.  hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int S = 61, P = 5056, NB = 99;

__global__ void __launch_bounds__(64) w_rows16(double* base, int rd)
{
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    double* dst = base + (size_t) blockIdx.y * S * P + (size_t) blockIdx.x * 16 + n;
    double acc = 0.0;
    for (int q = 0; q < 16; ++q) {
        const int i = g + 4 * q;
        if (i < S) { if (rd) acc += dst[(size_t) i * P]; else dst[(size_t) i * P] = (double) (i + lane); }
    }
    if (rd && acc == 12345.678) dst[0] = acc;
}
__global__ void __launch_bounds__(64) w_rows64(double* base, int rd)
{
    const int lane = threadIdx.x;
    double* dst = base + (size_t) blockIdx.y * S * P + (size_t) blockIdx.x * 64 + lane;
    double acc = 0.0;
    for (int i = 0; i < S; ++i) { if (rd) acc += dst[(size_t) i * P]; else dst[(size_t) i * P] = (double) (i + lane); }
    if (rd && acc == 12345.678) dst[0] = acc;
}
__global__ void __launch_bounds__(64) w_tile16(double* base, int rd)
{
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    double* dst = base + (size_t) blockIdx.y * S * P + (size_t) blockIdx.x * 16 * S + n;
    double acc = 0.0;
    for (int q = 0; q < 16; ++q) {
        const int i = g + 4 * q;
        if (i < S) { if (rd) acc += dst[(size_t) i * 16]; else dst[(size_t) i * 16] = (double) (i + lane); }
    }
    if (rd && acc == 12345.678) dst[0] = acc;
}
__global__ void __launch_bounds__(64) w_tile16x4(double* base, int rd)
{
    const int lane = threadIdx.x;
    double2* dst = reinterpret_cast<double2*>(base + (size_t) blockIdx.y * S * P + (size_t) blockIdx.x * 16 * S) + lane;
    double acc = 0.0;
    for (int q = 0; q < 8; ++q) {                       // 61 x 16 doubles = 488 double2 = 7.6 wave-wide stores
        const int e = q * 64 + lane;
        if (e < S * 8) { if (rd) { double2 v = dst[q * 64]; acc += v.x + v.y; } else dst[q * 64] = make_double2((double) e, (double) lane); }
    }
    if (rd && acc == 12345.678) dst[0].x = acc;
}
template <class K> void run(const char* name, K kern, dim3 grid, double* d, int rd)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, d, rd);
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, d, rd);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double) NB * S * P * 8;
    printf("%-10s %-5s %8.1f us per pass  %6.2f TB/s\n", name, rd ? "read" : "write", ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}
int main()
{
    double* d; const size_t bytes = (size_t) NB * S * P * 8;
    // four buffers so that successive passes do not hit the same lines in the last-level cache? no: one pass is 244 MB, about the cache's size
    CHECK(hipMalloc(&d, bytes + 4096)); CHECK(hipMemset(d, 0, bytes));
    for (int rd = 0; rd < 2; ++rd) {
        run("rows16", w_rows16, dim3(P / 16, NB), d, rd);
        run("rows64", w_rows64, dim3(P / 64, NB), d, rd);
        run("tile16", w_tile16, dim3(P / 16, NB), d, rd);
        run("tile16x4", w_tile16x4, dim3(P / 16, NB), d, rd);
    }
    return 0;
}
