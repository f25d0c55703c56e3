"""Build the TEST-ONLY host-emulation variant of the engine (see hip_emu.h).  Never used by the product."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "mrbayes_amd", "csrc", "mbamd_engine.cpp")
OUT = os.path.join(HERE, "_build", "libmbamd_hostemu_TESTONLY.so")


def host_compiler():
    """clang++ of the ROCm toolchain, as a HOST compiler: the emulation compiles the product's own matrix-core kernel bodies
    (clang vector types, __builtin_nontemporal_store) with the gfx950 intrinsics replaced by helpers of tests/hostemu/."""
    cands = [os.environ.get("MBAMD_EMU_CXX")]
    for hip in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if hip:
            cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hip))), "lib", "llvm", "bin", "clang++"))
    cands += ["/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")]
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise RuntimeError("the host emulation needs clang++ (the ROCm toolchain's host compiler)")


def build():
    csrc = os.path.join(ROOT, "mrbayes_amd", "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc) if not os.path.isdir(os.path.join(csrc, f))]
    deps += [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    deps += [os.path.join(ROOT, "include", "libhmsbeagle", f) for f in os.listdir(os.path.join(ROOT, "include", "libhmsbeagle"))]
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # no preprocessor switch: tests/hostemu comes FIRST on the include path, so the engine's <mbamd_dev_*.h> device-primitive
    # headers resolve to the plain-C++ twins here instead of the gfx950 ones in mrbayes_amd/csrc/device/
    subprocess.check_call([host_compiler(), "-O2", "-std=c++17", "-fPIC", "-shared", "-w",
                           "-I", HERE, "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "mrbayes_amd", "csrc"), SRC, "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build())
