#!/usr/bin/env python3
"""Apply the device-parsimony binding (integration/mrbayes/mbamd_pars_glue.h) to a TEMPORARY copy of the reference's
src/proposal.c -- test infrastructure: oracle/Makefile calls this for _ref/mb_amd_pars and _ref/mb_emu_pars.

    patch_pars.py <reference src/proposal.c> <output proposal.c>

The edits are located by the reference's own function names and comments (no reference text is stored here):
  * the glue header is included after proposal.c's last #include;
  * Move_ParsSPR1 and Move_ParsTBR1 are bracketed by #define/#undef of GetParsDP / GetParsFP to the glue's versions;
  * inside each, the candidate loop between the comments "cycle through the possibilities and record the parsimony
    length" and "find the min length and the sum for the forward move" is kept, wrapped in `if (MbamdParsHostToo ...)`,
    with the device call in front of it and the comparison (MBAMD_PARS_CHECK=1) behind it.
The output is written outside the repository (the Makefile passes a path under $(OBJ), /tmp by default).
"""
import re
import sys

BEGIN = "cycle through the possibilities and record the parsimony length"
END = "find the min length and the sum for the forward move"

SPR1_CALL = """    if (MbamdParsActive (t) == YES)
        {
        if (MbamdParsLengths (t, chain, moveInRoot == YES ? 0 : (u->anc == NULL ? 1 : 2), pRoot, nRoot, pCrown, nCrown,
                              a, b, u, v, nSitesOfPat, warpFactor, parLength) == ERROR)
            goto errorExit;
        }
"""
TBR1_CALL = """    if (MbamdParsActive (t) == YES)
        {
        if (MbamdParsLengths (t, chain, 3, pRoot, nRoot, pCrown, nCrown, a, b, u, v, nSitesOfPat, warpFactor, parLength) == ERROR)
            goto errorExit;
        }
"""


def patch(src: str) -> str:
    lines = src.split("\n")
    # 1. the include
    last_inc = max(i for i, l in enumerate(lines[:200]) if l.startswith("#include"))
    lines.insert(last_inc + 1, '#include "mbamd_pars_glue.h"')

    def function_range(name):
        start = next(i for i, l in enumerate(lines) if re.match(r"^int %s \(" % name, l))
        end = next(i for i in range(start + 1, len(lines)) if re.match(r"^int Move_\w+ \(", lines[i]))
        # back up over the comment block that precedes the next function
        while end > start and not lines[end - 1].startswith("}"):
            end -= 1
        return start, end

    for name, call in (("Move_ParsTBR1", TBR1_CALL), ("Move_ParsSPR1", SPR1_CALL)):      # later function first: indices stay valid
        start, end = function_range(name)
        body = range(start, end)
        b = [i for i in body if BEGIN in lines[i]]
        e = [i for i in body if END in lines[i]]
        assert len(b) == 1 and len(e) == 1 and b[0] < e[0], (name, b, e)
        assert any("errorExit:" in lines[i] for i in body), name
        new = (lines[:start] + ["#define GetParsDP MbamdGetParsDP", "#define GetParsFP MbamdGetParsFP"] + lines[start:b[0] + 1]
               + call.rstrip("\n").split("\n")
               + ["    if (MbamdParsHostToo (t, parLength, nRoot * nCrown) == YES)", "    {"]
               + lines[b[0] + 1:e[0]]
               + ["    }", "    MbamdParsCompare (parLength, nRoot * nCrown);", ""]
               + lines[e[0]:end] + ["#undef GetParsDP", "#undef GetParsFP"] + lines[end:])
        lines = new
    return "\n".join(lines)


if __name__ == "__main__":
    with open(sys.argv[1]) as f:
        out = patch(f.read())
    with open(sys.argv[2], "w") as f:
        f.write(out)
