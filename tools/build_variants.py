#!/usr/bin/env python3
"""Build experiment variants of the engine next to the product library: build_x/libhmsbeagle_<name>.so, each with extra -D
switches (ablations / alternatives measured on the GPU with MBAMD_LIBRARY=...).  build_x/ is git-ignored but travels with gpurun.

    python tools/build_variants.py name1=DEF_A,DEF_B name2=DEF_C ...
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrbayes_amd import build as mbbuild  # noqa: E402


def one(spec):
    name, _, defs = spec.partition("=")
    out = os.path.join(ROOT, "build_x", "libhmsbeagle_%s.so" % name)
    cmd = [mbbuild.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "-fvisibility=hidden",
           "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "mrbayes_amd", "csrc"), "-I", os.path.join(ROOT, "mrbayes_amd", "csrc", "device")]
    cmd += ["-D" + d for d in defs.split(",") if d]
    cmd += [mbbuild.SRC, "-o", out]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "build_x"), exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        for o in ex.map(one, sys.argv[1:]):
            print(o)
