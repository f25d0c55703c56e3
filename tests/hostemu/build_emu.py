"""Build the TEST-ONLY host-emulation variant of the engine (see hip_emu.h).  Never used by the product."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "mrbayes_amd", "csrc", "mbamd_engine.cpp")
OUT = os.path.join(HERE, "_build", "libmbamd_hostemu_TESTONLY.so")


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
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w",
                           "-I", HERE, "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "mrbayes_amd", "csrc"), SRC, "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build())
