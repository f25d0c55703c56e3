// hip_emu.h -- TEST INFRASTRUCTURE ONLY.  A few dozen lines of "HIP on the host" so that the engine's
// host logic (buffer bookkeeping, operation scheduling, LDS-slot allocation, layout conversion, the
// BEAGLE call protocol) can be exercised by `pytest -m "not gpu"` in a container without a GPU.
// Kernels are run one thread after another; kernels that use workgroup barriers are launched with
// MBAMD_LAUNCH_BARRIER, which runs the threads of a block as fibers (ucontext) that yield at every barrier.
//
// It is compiled only into tests/hostemu/_build/libmbamd_hostemu_TESTONLY.so.  The product library
// (mrbayes_amd/libhmsbeagle.so) is always built by hipcc for gfx950 against the real HIP runtime and has
// no CPU path; nothing under mrbayes_amd/ references this file.
#ifndef MBAMD_HIP_EMU_H_
#define MBAMD_HIP_EMU_H_

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
#include <ucontext.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static              // (the blocks of a launch run one after the other)
#ifndef __restrict__
#define __restrict__
#endif

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
inline int min(int a, int b) { return a < b ? a : b; }      // (HIP declares these at global scope)
inline int max(int a, int b) { return a > b ? a : b; }

inline dim3& emu_threadIdx() { static thread_local dim3 v; return v; }
inline dim3& emu_blockIdx()  { static thread_local dim3 v; return v; }
inline dim3& emu_blockDim()  { static thread_local dim3 v; return v; }
inline dim3& emu_gridDim()   { static thread_local dim3 v; return v; }
#define threadIdx emu_threadIdx()
#define blockIdx  emu_blockIdx()
#define blockDim  emu_blockDim()
#define gridDim   emu_gridDim()

inline void*& emu_lds_ptr() { static thread_local void* p = nullptr; return p; }
inline void* mbamd_emu_dyn_lds() { return emu_lds_ptr(); }

inline int atomicAdd(int* p, int v) { int o = *p; *p += v; return o; }

typedef int hipError_t;
typedef int hipStream_t;
struct EmuEvent { std::chrono::steady_clock::time_point t; };
typedef EmuEvent* hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
#define hipStreamNonBlocking 0
#define hipHostMallocDefault 0
struct hipDeviceProp_t { char name[256]; size_t totalGlobalMem; int multiProcessorCount; char gcnArchName[256]; };

inline const char* hipGetErrorString(hipError_t) { return "host-emulation error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    std::snprintf(p->name, sizeof p->name, "host emulation (TEST ONLY)");
    std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "hostemu");
    p->totalGlobalMem = (size_t) 8 << 30; p->multiProcessorCount = 256; return hipSuccess; }   // the geometry heuristics see an MI355X
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**) p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**) p, n, f); }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2D(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
    for (size_t r = 0; r < height; ++r) std::memcpy((char*) d + r * dpitch, (const char*) s + r * spitch, width);
    return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = 0; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new EmuEvent; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
#define hipEventDisableTiming 2
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new EmuEvent; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipStreamWriteValue32(hipStream_t, void* ptr, uint32_t value, unsigned) { *static_cast<uint32_t*>(ptr) = value; return hipSuccess; }   // (kernels run at launch)
inline hipError_t hipDeviceGetPCIBusId(char* out, int len, int dev) { std::snprintf(out, (size_t) len, "emu:%02x:00.0", dev); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }

template <class Kernel, class... Args>
inline void emu_launch(Kernel kernel, dim3 grid, dim3 block, size_t lds_bytes, Args... args)
{
    std::vector<unsigned char> lds(lds_bytes + 64);
    emu_gridDim() = grid;
    emu_blockDim() = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                emu_blockIdx() = dim3(bx, by, bz);
                void* base = lds.data();
                size_t space = lds.size();
                emu_lds_ptr() = std::align(16, lds_bytes, base, space);
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx) {
                            emu_threadIdx() = dim3(tx, ty, tz);
                            kernel(args...);
                        }
            }
}
#define MBAMD_LAUNCH(kernel, grid, block, lds, stream, ...) emu_launch(kernel, dim3(grid), dim3(block), lds, __VA_ARGS__)

// ---- workgroup barriers and wave collectives: the threads of one block as fibers ------------------------------
// A fiber runs until it has to wait for others (a workgroup barrier, a wave-wide exchange such as an emulated MFMA, a spin on a
// value another wave writes) and yields; the scheduler resumes the fibers of the block in thread order, round after round.
// Barriers COUNT arrivals -- waves of one workgroup may pass different numbers of wave-level exchanges between two workgroup
// barriers -- and threads that have left the kernel do not count (like s_barrier).  After a barrier the threads resume in thread
// order (thread 0 first: the plain-C++ reductions of mbamd_dev_base.h rely on it).
#if defined(__x86_64__)
// switching a fiber = six callee-saved registers and the stack pointer (swapcontext adds two system calls for the signal mask;
// an emulated 61-state operation is ~8 000 switches)
extern "C" void mbamd_emu_switch(void** save_sp, void* load_sp);
asm(".text\n.globl mbamd_emu_switch\n.hidden mbamd_emu_switch\n.type mbamd_emu_switch,@function\nmbamd_emu_switch:\n"
    "pushq %rbp\npushq %rbx\npushq %r12\npushq %r13\npushq %r14\npushq %r15\n"
    "movq %rsp, (%rdi)\nmovq %rsi, %rsp\n"
    "popq %r15\npopq %r14\npopq %r13\npopq %r12\npopq %rbx\npopq %rbp\nret\n"
    ".size mbamd_emu_switch, .-mbamd_emu_switch\n");
struct EmuContext { void* sp = nullptr; };
#else
struct EmuContext { ucontext_t uc; };
#endif
struct EmuWave {                                     // a wave's exchange area: what every lane contributed to the current collective
    unsigned arrived = 0, generation = 0, live = 0;
    uint64_t x[2][2][64];                            // [parity of the collective][operand][lane]
    unsigned parity = 0;
};
struct EmuFibers {
    EmuContext scheduler;
    std::vector<EmuContext> ctx;
    std::vector<std::vector<unsigned char>> stacks;
    std::vector<char> done;
    std::vector<EmuWave> waves;
    int current = -1;
    unsigned live = 0, arrived = 0, generation = 0;  // the workgroup barrier
    void (*body)(void*) = nullptr;
    void* arg = nullptr;
};
inline EmuFibers& emu_fibers() { static thread_local EmuFibers f; return f; }
inline void emu_fiber_entry();
inline void emu_to_scheduler(EmuFibers& f)
{
#if defined(__x86_64__)
    mbamd_emu_switch(&f.ctx[f.current].sp, f.scheduler.sp);
#else
    swapcontext(&f.ctx[f.current].uc, &f.scheduler.uc);
#endif
}
inline void emu_to_fiber(EmuFibers& f, unsigned t)
{
#if defined(__x86_64__)
    mbamd_emu_switch(&f.scheduler.sp, f.ctx[t].sp);
#else
    swapcontext(&f.scheduler.uc, &f.ctx[t].uc);
#endif
}
inline void emu_fiber_prepare(EmuFibers& f, unsigned t)
{
#if defined(__x86_64__)
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stacks[t].data()) + f.stacks[t].size()) & ~(uintptr_t) 15;
    void** sp = reinterpret_cast<void**>(top) - 8;   // r15 r14 r13 r12 rbx rbp | return address = the entry | a slot a caller would own
    for (int i = 0; i < 8; ++i) sp[i] = nullptr;
    sp[6] = reinterpret_cast<void*>(&emu_fiber_entry);
    f.ctx[t].sp = sp;
#else
    getcontext(&f.ctx[t].uc);
    f.ctx[t].uc.uc_stack.ss_sp = f.stacks[t].data();
    f.ctx[t].uc.uc_stack.ss_size = f.stacks[t].size();
    f.ctx[t].uc.uc_link = &f.scheduler.uc;
    makecontext(&f.ctx[t].uc, (void (*)()) emu_fiber_entry, 0);
#endif
}
// give the other threads of the block a turn (a spin on something another wave writes)
inline void mbamd_emu_yield()
{
    EmuFibers& f = emu_fibers();
    if (f.current >= 0) emu_to_scheduler(f);
}
inline bool mbamd_emu_fibers_active() { return emu_fibers().current >= 0; }
inline void mbamd_emu_barrier()
{
    EmuFibers& f = emu_fibers();
    if (f.current < 0) return;                       // plain launch: threads run to completion one after another
    const unsigned gen = f.generation;
    if (++f.arrived >= f.live) { f.arrived = 0; ++f.generation; }
    do emu_to_scheduler(f); while (f.generation == gen);   // (the last to arrive yields too: everybody resumes in thread order)
}
// all lanes of the calling thread's wave have arrived (only wave-uniform control flow may reach it)
inline EmuWave& mbamd_emu_wave()
{
    EmuFibers& f = emu_fibers();
    return f.waves[(unsigned) f.current >> 6];
}
inline void mbamd_emu_wave_sync()
{
    EmuFibers& f = emu_fibers();
    if (f.current < 0) return;
    EmuWave& w = f.waves[(unsigned) f.current >> 6];
    const unsigned gen = w.generation;
    if (++w.arrived >= w.live) { w.arrived = 0; ++w.generation; return; }
    do emu_to_scheduler(f); while (w.generation == gen);
}
// every lane contributes two (up to) 64-bit values and sees what all lanes contributed (one synchronisation per collective: the areas of
// consecutive collectives alternate, and nobody can be two collectives ahead of a lane that has not read yet)
struct EmuExchange { const uint64_t* a; const uint64_t* b; };
inline EmuExchange mbamd_emu_exchange(uint64_t a, uint64_t b)
{
    EmuFibers& f = emu_fibers();
    if (f.current < 0) { std::fprintf(stderr, "host emulation: a wave-wide exchange in a kernel launched without fibers\n"); std::abort(); }
    EmuWave& w = f.waves[(unsigned) f.current >> 6];
    const unsigned lane = (unsigned) f.current & 63u;
    const unsigned par = w.parity;                   // (uniform: flipped by the lane that completes the collective)
    w.x[par][0][lane] = a;
    w.x[par][1][lane] = b;
    const unsigned gen = w.generation;
    if (++w.arrived >= w.live) { w.arrived = 0; w.parity ^= 1u; ++w.generation; }
    else do emu_to_scheduler(f); while (w.generation == gen);
    return EmuExchange{w.x[par][0], w.x[par][1]};
}
inline void emu_fiber_entry()
{
    EmuFibers& f = emu_fibers();
    f.body(f.arg);
    f.done[f.current] = 1;
    // a thread that leaves no longer takes part in barriers: release whoever was only waiting for it
    EmuWave& w = f.waves[(unsigned) f.current >> 6];
    --f.live; --w.live;
    if (f.live > 0 && f.arrived >= f.live) { f.arrived = 0; ++f.generation; }
    if (w.live > 0 && w.arrived >= w.live) { w.arrived = 0; w.parity ^= 1u; ++w.generation; }
    for (;;) emu_to_scheduler(f);
}
template <class Kernel, class... Args>
inline void emu_launch_barrier(Kernel kernel, dim3 grid, dim3 block, size_t lds_bytes, Args... args)
{
    std::vector<unsigned char> lds(lds_bytes + 64);
    emu_gridDim() = grid;
    emu_blockDim() = block;
    EmuFibers& f = emu_fibers();
    const unsigned T = block.x;
    if (f.ctx.size() < T) { f.ctx.resize(T); f.stacks.resize(T); }
    for (unsigned t = 0; t < T; ++t) if (f.stacks[t].empty()) f.stacks[t].resize(256 * 1024);
    f.done.assign(T, 0);
    f.waves.resize((T + 63) / 64);
    auto call = [&]() { kernel(args...); };
    f.body = [](void* p) { (*static_cast<decltype(call)*>(p))(); };
    f.arg = &call;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
            emu_blockIdx() = dim3(bx, by, bz);
            void* base = lds.data();
            size_t space = lds.size();
            emu_lds_ptr() = std::align(16, lds_bytes, base, space);
            std::fill(f.done.begin(), f.done.end(), 0);
            f.live = T; f.arrived = 0; f.generation = 0;
            for (unsigned w = 0; w < f.waves.size(); ++w) {
                f.waves[w].arrived = 0; f.waves[w].generation = 0; f.waves[w].parity = 0;
                f.waves[w].live = std::min(64u, T - 64u * w);
            }
            for (unsigned t = 0; t < T; ++t) emu_fiber_prepare(f, t);
            unsigned live = T;
            while (live > 0) {                       // one pass = every live thread runs until it has to wait (or to its end)
                live = 0;
                for (unsigned t = 0; t < T; ++t) {
                    if (f.done[t]) continue;
                    emu_threadIdx() = dim3(t, 0, 0);
                    f.current = (int) t;
                    emu_to_fiber(f, t);
                    if (!f.done[t]) ++live;
                }
            }
            f.current = -1;
        }
}
#define MBAMD_LAUNCH_BARRIER(kernel, grid, block, lds, stream, ...) emu_launch_barrier(kernel, dim3(grid), dim3(block), lds, __VA_ARGS__)

#endif
