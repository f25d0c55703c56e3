#!/usr/bin/env python3
"""Apply the device-parsimony binding (integration/mrbayes/mbamd_pars_glue.h) to a TEMPORARY copy of the reference's
src/proposal.c -- test infrastructure: oracle/Makefile calls this for _ref/mb_amd_pars and _ref/mb_emu_pars.

    patch_pars.py <reference src/proposal.c> <output proposal.c>
    patch_pars.py <reference src/model.c>    <output model.c>         (pattern compression, mbamd_compress_glue.h)

The edits are located by the reference's own function names and comments (no reference text is stored here):
  * the glue header is included after proposal.c's last #include;
  * the seven parsimony-biased moves (Move_ParsSPR, _ParsSPR1, _ParsSPR2, _ParsSPRClock, _ParsSPRClock_Fossil, _ParsTBR1,
    _ParsTBR2) are bracketed by #define/#undef of GetParsDP / GetParsFP to the glue's versions;
  * inside each, the loop between the comments "cycle through the possibilities and record the parsimony length" and
    "find the min length and the sum for the forward move" is kept: the candidate-matrix moves get it wrapped in
    `if (MbamdParsHostToo ...)` with the device call in front and the comparison (MBAMD_PARS_CHECK=1) behind; the
    node-length moves get the per-pattern loop of a marked node wrapped in `if (MbamdParsMarkedLength ...)`.
The output is written outside the repository (the Makefile passes a path under $(OBJ), /tmp by default).
"""
import os
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

    # later functions first: indices stay valid
    for name, call in (("Move_ParsTBR2", TBR1_CALL), ("Move_ParsTBR1", TBR1_CALL), ("Move_ParsSPRClock_Fossil", None), ("Move_ParsSPRClock", None),
                       ("Move_ParsSPR2", SPR1_CALL), ("Move_ParsSPR1", SPR1_CALL), ("Move_ParsSPR", None)):
        start, end = function_range(name)
        body = range(start, end)
        b = [i for i in body if BEGIN in lines[i]]
        e = [i for i in body if END in lines[i]]
        assert len(b) == 1 and len(e) == 1 and b[0] < e[0], (name, b, e)
        if call is not None:
            # candidate-matrix shape: parLength[i + j*nRoot]
            assert any("errorExit:" in lines[i] for i in body), name
            middle = (call.rstrip("\n").split("\n")
                      + ["    if (MbamdParsHostToo (t, parLength, nRoot * nCrown) == YES)", "    {"]
                      + lines[b[0] + 1:e[0]]
                      + ["    }", "    MbamdParsCompare (parLength, nRoot * nCrown);", ""])
        else:
            # node-length shape: `length = 0.0;` followed by the two-branch loop over the site patterns of marked node p
            region = lines[b[0] + 1:e[0]]
            at = [i for i, l in enumerate(region) if l.strip() == "length = 0.0;"]
            assert len(at) == 1 and "nParsIntsPerSite == 1" in region[at[0] + 1], (name, at)
            depth, stop, seen_else = 0, None, False
            for i in range(at[0] + 1, len(region)):        # the if { } else { } statement ends where the else-block closes
                depth += region[i].count("{") - region[i].count("}")
                if region[i].lstrip().startswith("else"):
                    seen_else = True
                if seen_else and depth == 0 and "}" in region[i]:
                    stop = i
                    break
            assert stop is not None, name
            middle = (["    MbamdParsMarkedLengths (t, chain, v, nSitesOfPat);"] + region[:at[0] + 1]
                      + ["            if (MbamdParsMarkedLength (t, n, p, &length) == NO)", "            {"]
                      + region[at[0] + 1:stop + 1]
                      + ["            }", "            MbamdParsMarkedCheck (t, n, p, length);"]
                      + region[stop + 1:])
        lines = (lines[:start] + ["#define GetParsDP MbamdGetParsDP", "#define GetParsFP MbamdGetParsFP"] + lines[start:b[0] + 1]
                 + middle + lines[e[0]:end] + ["#undef GetParsDP", "#undef GetParsFP"] + lines[end:])
    return "\n".join(lines)


COMPRESS_BEGIN = "/* is it unique? */"
COMPRESS_END = "/* if subject to data augmentation, it is always unique */"


def patch_model(src: str) -> str:
    """src/model.c, CompressData: the O(columns^2) search for an identical kept column becomes a hash-table lookup
    (integration/mrbayes/mbamd_compress_glue.h); the reference's search is kept for the switched-off and the check mode."""
    lines = src.split("\n")
    last_inc = max(i for i, l in enumerate(lines[:200]) if l.startswith("#include"))
    lines.insert(last_inc + 1, '#include "mbamd_compress_glue.h"')
    start = next(i for i, l in enumerate(lines) if re.match(r"^int CompressData \(", l))
    b = next(i for i in range(start, len(lines)) if COMPRESS_BEGIN in lines[i])
    e = next(i for i in range(b, len(lines)) if COMPRESS_END in lines[i])
    orig = lines[b + 1:e]
    assert any("isSame = NO;" in l for l in orig[:3]) and any("isSame = YES;" in l for l in orig), "CompressData does not look as expected"
    new = (["            if (mp->dataType != CONTINUOUS && MbamdCompressActive () == YES)",
            "                {",
            "                int mbamdSame, mbamdWhere;",
            "                mbamdSame = MbamdFindSamePattern (tempMatrix, numLocalChar, numLocalTaxa, m->nCharsPerSite, m->compMatrixStart, newColumn, &mbamdWhere);",
            "                if (MbamdCompressCheckWanted () == YES)",
            "                    {"]
           + orig
           + ["                    MbamdCompressCompare (isSame, i, mbamdSame, mbamdWhere);",
              "                    }",
              "                isSame = mbamdSame;",
              "                i = mbamdWhere;",
              "                }",
              "            else",
              "                {"]
           + orig
           + ["                }"])
    return "\n".join(lines[:b + 1] + new + lines[e:])


if __name__ == "__main__":
    with open(sys.argv[1]) as f:
        text = f.read()
    out = patch_model(text) if os.path.basename(sys.argv[1]) == "model.c" else patch(text)
    with open(sys.argv[2], "w") as f:
        f.write(out)
