#!/usr/bin/env python3
"""Apply the standard-data binding (integration/mrbayes/mbamd_std_glue.h) to a TEMPORARY copy of the reference's
src/likelihood.c (oracle/Makefile: _ref/mb_amd_std, _ref/mb_emu_std, and -- on top of patch_eigen.py -- _ref/mb_*_full).

    patch_std.py <src/likelihood.c, possibly already patched by patch_eigen.py> <output likelihood.c>
    patch_std.py --mcmc <src/mcmc.c, possibly already patched by patch_reports.py> <output mcmc.c>

Exact-text replacements of a code fragment with an asserted count (see patch_reports.py): an upstream change stops the build."""
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


def patch_mcmc(text):
    lines = text.split("\n")
    last_inc = max(i for i, l in enumerate(lines[:200]) if l.startswith("#include"))
    lines.insert(last_inc + 1, '#include "mbamd_std_glue.h"')
    text = "\n".join(lines)
    # PrintStates: a standard-data division on the engine gets its host arrays filled before the reference's final pass reads them
    text = replace(text,
                   "                for (i=j=tree->nIntNodes - 1; i>=0; i--)\n",
                   "                (void) MbamdStdMaterialise (coldId, d);\n"
                   "                for (i=j=tree->nIntNodes - 1; i>=0; i--)\n",
                   1, "final-pass loop of PrintStates")
    # FreeChainMemory: the binding's instances and class tables go with the reference's own chain memory -- a second mcmc / execute
    # of the session (other chains, taxa, characters, categories, priors) looks at the model again
    text = replace(text,
                   "void FreeChainMemory (void)\n{\n    int         i, j, k, nRates;\n    ModelInfo   *m;\n",
                   "void FreeChainMemory (void)\n{\n    int         i, j, k, nRates;\n    ModelInfo   *m;\n\n    MbamdStdFinalize ();\n",
                   1, "top of FreeChainMemory")
    return text


if __name__ == "__main__":
    mcmc = sys.argv[1] == "--mcmc"
    args = sys.argv[2:] if mcmc else sys.argv[1:]
    with open(args[0]) as f:
        src = f.read()
    with open(args[1], "w") as f:
        f.write(patch_mcmc(src) if mcmc else patch(src))
