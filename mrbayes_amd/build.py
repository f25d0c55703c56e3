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


# A second build of the same sources with every general-state contraction on v_mfma_f32_32x32x2_f32 (round 5's arithmetic): the
# measured A/B partner of the bf16 x 3 contraction (bench.py reports both for the codon workload; MBAMD_LIBRARY selects it).
FP32_CHAIN_LIB = os.path.join(HERE, "libhmsbeagle_fp32chain.so")


def build_fp32_chain_variant(force=False):
    return build_library(force=force, target=FP32_CHAIN_LIB, defines=["MBAMD_WG_BF_MIN=999"])


def build_library(force=False, verbose=False, target=None, defines=()):
    global LIB
    if target is not None:
        saved, LIB = LIB, target
        try:
            return _build(force, verbose, defines)
        finally:
            LIB = saved
    return _build(force, verbose, defines)


def _build(force, verbose, defines):
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
    for d in list(defines) + os.environ.get("MBAMD_BUILD_DEFINES", "").split():      # variants / experiments: extra -D switches
        cmd.insert(1, "-D" + d)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
