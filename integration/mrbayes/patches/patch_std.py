#!/usr/bin/env python3
"""Apply the standard-data binding (integration/mrbayes/mbamd_std_glue.h) to a TEMPORARY copy of the reference's
src/likelihood.c (oracle/Makefile: _ref/mb_amd_std, _ref/mb_emu_std, and -- on top of patch_eigen.py -- _ref/mb_*_full).

    patch_std.py <src/likelihood.c, possibly already patched by patch_eigen.py> <output likelihood.c>

One exact-text replacement of a code fragment with an asserted count (see patch_reports.py): an upstream change stops the build."""
import sys

from patch_reports import replace


def patch(text):
    lines = text.split("\n")
    last_inc = max(i for i, l in enumerate(lines[:200]) if l.startswith("#include"))
    lines.insert(last_inc + 1, '#include "mbamd_std_glue.h"')
    text = "\n".join(lines)
    # LaunchLogLikeForDivision: a standard-data division the engine serves skips the host pass over the tree
    text = replace(text,
                   "    if (m->parsModelId == NO && m->dataType != CONTINUOUS)\n        {\n",
                   "    if (MbamdStdServes (m) == YES)\n        {\n        MbamdStdLogLike (chain, d, lnL);\n        return;\n        }\n"
                   "    if (m->parsModelId == NO && m->dataType != CONTINUOUS)\n        {\n",
                   1, "host pass of LaunchLogLikeForDivision")
    return text


if __name__ == "__main__":
    with open(sys.argv[1]) as f:
        src = f.read()
    with open(sys.argv[2], "w") as f:
        f.write(patch(src))
