// mbamd_dev_runtime.h (gfx950) -- the HIP runtime and the launch forms the engine's host side uses.  The TEST-ONLY host emulation
// has a header of the same name in front on its include path (tests/hostemu/: kernels run as plain loops / fibers on the CPU).
#ifndef MBAMD_DEV_RUNTIME_H_
#define MBAMD_DEV_RUNTIME_H_
#include <hip/hip_runtime.h>
#define MBAMD_LAUNCH(kernel, grid, block, lds, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, stream, __VA_ARGS__)
#define MBAMD_LAUNCH_BARRIER MBAMD_LAUNCH          // (the host emulation runs kernels with workgroup barriers as fibers)
#endif
