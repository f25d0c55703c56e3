// Microbenchmark (gfx950): what can a second wave of a SIMD do while the first runs a chain of v_mfma_f32_32x32x2_f32?
// Eight waves per workgroup: waves 0-3 (role A) and 4-7 (role B) land pairwise on the four SIMDs of a CU.  Each role runs one
// of: 0 nothing, 1 dependent MFMA chain, 2 two interleaved MFMA chains, 3 dependent VALU chain (v_fma), 4 SALU loop, 5 LDS
// read loop, 6 mixed VALU+SALU (address-arithmetic like).  Prints the median duration of each role in microseconds.
//   hipcc --offload-arch=gfx950 -O3 mfma_shadow.hip -o mfma_shadow && ./mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void work(int role, int n, float* sink, float* lds)
{
    const unsigned lane = threadIdx.x & 63;
    if (role == 1) {
        v16 acc = {};
        float a = lane * 1e-3f, b = 1.0f;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (acc[0] == 123.456f) sink[lane] = acc[3];
    } else if (role == 2) {
        v16 acc = {}, acc2 = {};
        float a = lane * 1e-3f, b = 1.0f;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc2, 0, 0, 0);
            }
        }
        if (acc[0] + acc2[1] == 123.456f) sink[lane] = acc[3];
    } else if (role == 3) {
        float x = lane * 1e-3f;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 256; ++u) x = __builtin_fmaf(x, 1.000001f, 1e-7f);      // 16 MFMA x 64 cycles = 1024 cycles = 256 VALU x 4
        }
        if (x == 123.456f) sink[lane] = x;
    } else if (role == 4) {
        unsigned s = __builtin_amdgcn_readfirstlane(n);
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 256; ++u) asm volatile("s_add_u32 %0, %0, 3\n\ts_xor_b32 %0, %0, 5" : "+s"(s) : : "scc");
        }
        if (s == 12345u) sink[lane] = 1.0f;
    } else if (role == 5) {
        float x = 0.0f;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) x += lds[(lane + u * 64 + (int) x) & 4095];
        }
        if (x == 123.456f) sink[lane] = x;
    } else if (role == 7 || role == 8) {
        v16 acc = {};
        float a = lane * 1e-3f, b = 1.0f;
        float x0 = lane * 1e-3f, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (role == 7) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                x0 = __builtin_fmaf(x0, 1.000001f, 1e-7f); x1 = __builtin_fmaf(x1, 1.000001f, 1e-7f);
                x2 = __builtin_fmaf(x2, 1.000001f, 1e-7f); x3 = __builtin_fmaf(x3, 1.000001f, 1e-7f);
                x0 = __builtin_fmaf(x0, 1.000001f, 1e-7f); x1 = __builtin_fmaf(x1, 1.000001f, 1e-7f);
                x2 = __builtin_fmaf(x2, 1.000001f, 1e-7f); x3 = __builtin_fmaf(x3, 1.000001f, 1e-7f);
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            }
        }
        if (acc[0] + x0 + x1 + x2 + x3 == 123.456f) sink[lane] = acc[3];
    } else if (role == 9) {
        // round 6: the same question for the 16-bit matrix core -- 32 v_mfma_f32_32x32x16_bf16 (32 cycles each) = the 1024 cycles of role 1
        v16 acc = {};
        const bf16x8 a = __builtin_bit_cast(bf16x8, u4{0x3c003c00u + lane, 0x3c003c10u, 0x3c003c20u, 0x3c003c30u}), b = __builtin_bit_cast(bf16x8, u4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        if (acc[0] == 123.456f) sink[lane] = acc[3];
    } else if (role == 10 || role == 11) {
        // 32 x (one bf16 MFMA + 4 independent v_fma_f32 behind it): do the vector instructions run in the MFMA's shadow?
        v16 acc = {};
        const bf16x8 a = __builtin_bit_cast(bf16x8, u4{0x3c003c00u + lane, 0x3c003c10u, 0x3c003c20u, 0x3c003c30u}), b = __builtin_bit_cast(bf16x8, u4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
        float x0 = lane * 1e-3f, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                if (role == 10) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                x0 = __builtin_fmaf(x0, 1.000001f, 1e-7f); x1 = __builtin_fmaf(x1, 1.000001f, 1e-7f);
                x2 = __builtin_fmaf(x2, 1.000001f, 1e-7f); x3 = __builtin_fmaf(x3, 1.000001f, 1e-7f);
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            }
        }
        if (acc[0] + x0 + x1 + x2 + x3 == 123.456f) sink[lane] = acc[3];
    } else if (role == 6) {
        float x = lane * 1e-3f;
        unsigned s = __builtin_amdgcn_readfirstlane(n);
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 128; ++u) {
                x = __builtin_fmaf(x, 1.000001f, 1e-7f);
                asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");
            }
        }
        if (x == 123.456f || s == 12345u) sink[lane] = x;
    }
}
__global__ void __launch_bounds__(512) k(int roleA, int roleB, int n, float* sink, long long* out, int prioA, int prioB)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = 0.0f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const int pr = wave < 4 ? prioA : prioB;
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 3) __builtin_amdgcn_s_setprio(3);
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    work(wave < 4 ? roleA : roleB, n, sink, lds);
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        long long* o = out + ((size_t) blockIdx.x * 8 + wave) * 2;
        o[0] = t1 - t0;
        o[1] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
    }
}
int main()
{
    const int blocks = 256, n = 200;
    float* sink; long long* out;
    hipMalloc(&sink, 4096); hipMalloc(&out, blocks * 8 * 2 * sizeof(long long));
    std::vector<long long> h(blocks * 8 * 2);
    const char* names[] = {"idle", "mfma dependent chain", "mfma two chains", "valu chain", "salu loop", "lds reads", "valu+salu", "mfma + 8 valu between", "the 8 valu alone", "bf16 mfma chain", "bf16 mfma + 4 valu between", "the 4 valu alone"};
    const int cases[][4] = {{1, 0, 0, 0}, {3, 0, 0, 0}, {6, 0, 0, 0}, {1, 3, 0, 0},                       // round 5's baseline rows again (same box)
                            {9, 0, 0, 0}, {9, 3, 0, 0}, {9, 6, 0, 0}, {9, 5, 0, 0}, {9, 9, 0, 0}, {9, 1, 0, 0},     // round 6: the 16-bit matrix core
                            {11, 0, 0, 0}, {10, 0, 0, 0}, {10, 10, 0, 0}, {10, 11, 0, 0}};
    printf("start\n"); fflush(stdout);
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, c[0], c[1], n, sink, out, c[2], c[3]);
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); fflush(stdout); return 1; }
        }
        hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        std::vector<double> a, b;
        int sameSimd = 0;
        for (int bl = 0; bl < blocks; ++bl) {
            for (int w = 0; w < 8; ++w) (w < 4 ? a : b).push_back(h[((size_t) bl * 8 + w) * 2] / 100.0);
            for (int w = 0; w < 4; ++w) sameSimd += ((h[((size_t) bl * 8 + w) * 2 + 1] >> 4) & 3) == ((h[((size_t) bl * 8 + w + 4) * 2 + 1] >> 4) & 3);
        }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        fflush(stdout); printf("prio %d/%d A %-22s B %-22s : A %8.1f us   B %8.1f us   (waves w, w+4 on one SIMD: %d of %d)\n", c[2], c[3], names[c[0]], names[c[1]], a[a.size() / 2], b[b.size() / 2], sameSimd, blocks * 4);
    }
    fflush(stdout);
    return 0;
}
