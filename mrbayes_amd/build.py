"""Build the engine's shared library in-tree with hipcc for gfx950.

    python -m mrbayes_amd.build            -> mrbayes_amd/libhmsbeagle.so   (the product, HIP only)

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU
box with the repo snapshot.  The library name is what MrBayes links (`-lhmsbeagle`,
reference configure.ac:179-182).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "mbamd_engine.cpp")
DEPS = [SRC, os.path.join(HERE, "csrc", "mbamd_kernels.h"), os.path.join(ROOT, "include", "libhmsbeagle", "beagle.h")]
LIB = os.path.join(HERE, "libhmsbeagle.so")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the engine can only be built with the ROCm toolchain")


def build_library(force=False, verbose=False):
    deps = list(DEPS)
    for d in (os.path.join(HERE, "csrc"), os.path.join(HERE, "csrc", "device"), os.path.join(ROOT, "include", "libhmsbeagle")):
        deps += [os.path.join(d, f) for f in os.listdir(d) if os.path.isfile(os.path.join(d, f))]
    if not force and not _stale(LIB, deps):
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-I", os.path.join(HERE, "csrc", "device"), SRC, "-o", LIB]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    for d in os.environ.get("MBAMD_BUILD_DEFINES", "").split():      # experiments: extra -D switches
        cmd.insert(1, "-D" + d)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
