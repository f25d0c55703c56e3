"""Build the TEST-ONLY host-emulation variant of the engine (see hip_emu.h).  Never used by the product."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "mrbayes_amd", "csrc", "mbamd_engine.cpp")
OUT = os.path.join(HERE, "_build", "libmbamd_hostemu_TESTONLY.so")


def build():
    csrc = os.path.join(ROOT, "mrbayes_amd", "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(HERE, "hip_emu.h"), os.path.join(HERE, "mbamd_walkg_emu.h"), os.path.join(HERE, "mbamd_integrate_wg_emu.h"),
                                                               os.path.join(ROOT, "include", "libhmsbeagle", "beagle.h")]
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DMBAMD_HOST_EMU", "-fPIC", "-shared", "-w",
                           "-I", HERE, "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "mrbayes_amd", "csrc"), SRC, "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build())
