#!/usr/bin/env python3
"""Apply the device eigen-solver binding (integration/mrbayes/mbamd_eigen_glue.h) to a TEMPORARY copy of the reference's
src/likelihood.c -- test infrastructure: oracle/Makefile calls this for _ref/mb_amd_full / _ref/mb_emu_full.

    patch_eigen.py <reference src/likelihood.c> <output likelihood.c>

Exact-text replacements of code fragments with asserted counts (see patch_reports.py): an upstream change stops the build."""
import sys

from patch_reports import include_glue, replace


def patch(text):
    text = include_glue(text).replace('#include "mbamd_reports_glue.h"', '#include "mbamd_eigen_glue.h"', 1)
    text = replace(text, "isComplex = GetEigens (n, q[0], eigenValues,", "isComplex = MbamdGetEigens (m, whichChain, q, 0, n, q[0], eigenValues,", 2, "GetEigens of one rate matrix")
    text = replace(text, "isComplex = GetEigens (n, q[k], eigenValues,", "isComplex = MbamdGetEigens (m, whichChain, q, k, n, q[k], eigenValues,", 1, "GetEigens of the parts")
    return text


if __name__ == "__main__":
    with open(sys.argv[1]) as f:
        src = f.read()
    with open(sys.argv[2], "w") as f:
        f.write(patch(src))
