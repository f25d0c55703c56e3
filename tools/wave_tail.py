"""Does the fullest SIMD set k_walk4_t's time?  DNA 1000 taxa, GTR+G4, pattern counts around 50 000: 768 blocks x 4 categories = 3 072 waves
(three per SIMD exactly), 782 blocks = 3 128 (the BASELINE shape: 56 SIMDs carry a fourth wave), 1 024 blocks = 4 096 (four exactly).
Prints the partials kernel's time per evaluation and per pattern block.  usage: wave_tail.py [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mrbayes_amd import beagle as bg, likelihood as lk
from mrbayes_amd.division import synthetic_division
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lib = bg.BeagleLibrary()
for blocks in (704, 768, 782, 832, 896, 1024):
    div = synthetic_division("gtr", 1000, blocks * 64, seed=7, tree_seed=3)
    bd = lk.BeagleDivision(div, lib, scaling=lk.MB_BEAGLE_SCALE_ALWAYS)
    try:
        bd.LogLike(0); bd.AcceptMove(0)
        bd.inst.kernel_timing(True)
        for warm in range(2):
            for i in range(steps):
                bd.TouchAllTreeNodes(0); bd.LogLike(0); bd.AcceptMove(0)
            kms, kn = bd.inst.get_kernel_timing()
        print("%5d blocks (%6d patterns, %5d waves = %.3f per SIMD): partials kernel %.4f ms, %.4f us per block" %
              (blocks, blocks * 64, blocks * 4, blocks * 4 / 1024.0, kms / kn, kms / kn * 1e3 / blocks), flush=True)
    finally:
        bd.finalize()
