#!/usr/bin/env python3
"""Apply the reports / covarion binding (integration/mrbayes/mbamd_reports_glue.h) to TEMPORARY copies of the reference's
src/mcmc.c and src/mbbeagle.c (oracle/Makefile calls this for _ref/mb_amd_reports / _ref/mb_emu_reports / _ref/mb_*_full).

    patch_reports.py <reference src/mcmc.c>     <output mcmc.c>
    patch_reports.py <reference src/mbbeagle.c> <output mbbeagle.c>

Every edit is an exact-text replacement of a CODE fragment of the reference (no comment is used as an anchor) that must occur
exactly the stated number of times: an upstream change of any of the sites stops the build here instead of producing a binary
that silently does something else.  No reference text is stored beyond those one-line anchors.
"""
import os
import sys


def replace(text, old, new, count, what):
    n = text.count(old)
    if n != count:
        raise SystemExit("patch_reports.py: %s: expected %d occurrence(s) of %r, found %d -- the reference changed here" % (what, count, old, n))
    return text.replace(old, new)


def include_glue(text):
    lines = text.split("\n")
    last_inc = max(i for i, l in enumerate(lines[:200]) if l.startswith("#include"))
    lines.insert(last_inc + 1, '#include "mbamd_reports_glue.h"')
    return "\n".join(lines)


def patch_mcmc(text):
    text = include_glue(text)
    # 1. InitChainCondLikes: the engine keeps divisions the BEAGLE API cannot serve
    text = replace(text,
                   "                if (m->printAncStates == YES || m->printSiteRates == YES || m->printPosSel ==YES || m->printSiteOmegas==YES)\n",
                   "                if (MbamdEngineServes (m) == YES)\n"
                   "                    m->useBeagle = YES;\n"
                   "                else if (MbamdEngineRefuses (m, d+1) == YES)\n"
                   "                    ;\n"
                   "                else if (m->printAncStates == YES || m->printSiteRates == YES || m->printPosSel ==YES || m->printSiteOmegas==YES)\n",
                   1, "BEAGLE refusal in InitChainCondLikes")
    # 2. the read-outs of the top node's conditional likelihoods (PrintStates: column headers; PrintStatesToFiles: values)
    text = replace(text, "m->PosSelProbs (tree->root->left, d, coldId) == ERROR",
                   "(MbamdReportsRoot (tree->root->left, d, coldId) == ERROR || m->PosSelProbs (tree->root->left, d, coldId) == ERROR)",
                   2, "PosSelProbs calls")
    text = replace(text, "m->SiteOmegas (tree->root->left, d, coldId) == ERROR",
                   "(MbamdReportsRoot (tree->root->left, d, coldId) == ERROR || m->SiteOmegas (tree->root->left, d, coldId) == ERROR)",
                   2, "SiteOmegas calls")
    text = replace(text, "                m->PrintSiteRates (node, d, coldId);\n",
                   "                if (MbamdReportsRoot (node, d, coldId) == ERROR) goto errorExit;\n"
                   "                m->PrintSiteRates (node, d, coldId);\n",
                   1, "PrintSiteRates call")
    # 1b. (BEAGLE v3 builds) a division the binding serves keeps its OWN instance: the binding's buffer indices carry no division
    #     offset (the reference adds numTiProbs * nCijkParts * divisionIndex inside a multi-partition instance) and
    #     mbamdGetScaledPartials reads one partition
    text = replace(text, "        if (beagleResourceNumber != 0 && numCurrentDivisions > 1)\n",
                   "        if (beagleResourceNumber != 0 && numCurrentDivisions > 1 && MbamdEngineServesAny () == NO)\n",
                   1, "multi-partition instance decision")
    # 3. the final pass
    text = replace(text, "                    m->CondLikeUp (node, d, coldId);\n",
                   "                    if (MbamdReportsUp (tree, node, d, coldId) == NO)\n"
                   "                        m->CondLikeUp (node, d, coldId);\n",
                   1, "CondLikeUp loop")
    return text


def patch_mbbeagle(text):
    """Covarion divisions (m->switchRates != NULL): the K rate categories are K eigen-system PARTS with one category each --
    exactly how the file already treats the omega classes of a codon model -- instead of K parts x K categories."""
    text = include_glue(text)
    text = replace(text,
                   "createBeagleInstance(m, m->nCijkParts, m->numRateCats, m->numModelStates,",
                   "createBeagleInstance(m, m->nCijkParts, (m->switchRates != NULL ? 1 : m->numRateCats), m->numModelStates,",
                   1, "instance creation")
    # tips of a covarion division are partials (on- and off-states of the observed nucleotide / amino acid)
    text = replace(text,
                   "#else\n        if (m->isPartAmbig[i] == NO)\n#endif\n            {\n            charBits = m->parsSets[i];",
                   "#else\n        if (m->isPartAmbig[i] == NO && m->numStates == m->numModelStates)\n#endif\n            {\n            charBits = m->parsSets[i];",
                   1, "tip data")
    # category weights per part, every evaluation (the non-v3 TreeLikelihood_Beagle: the 4-space indented `if`)
    text = replace(text,
                   "\n    if (m->numOmegaCats > 1)\n",
                   "\n    if (m->switchRates != NULL)\n        {\n        for (i=0; i<m->nCijkParts; i++)\n            {\n"
                   "            m->inWeights[0] = freq;\n"
                   "            beagleSetCategoryWeights(m->beagleInstance, m->cijkIndex[chain] + i, m->inWeights);\n            }\n        }\n"
                   "    else if (m->numOmegaCats > 1)\n",
                   1, "category weights")
    # the covarion eigen-systems carry their category's rate (reference TiProbs_GenCov, src/likelihood.c:9632: t = length * correctionFactor)
    text = replace(text,
                   "\n        m->inRates[k] = baseRate * catRate[k] * correctionFactor;\n",
                   "\n        m->inRates[k] = baseRate * catRate[k] * correctionFactor;\n"
                   "    if (m->switchRates != NULL)\n        m->inRates[0] = correctionFactor;\n",
                   1, "category rates")
    return text


if __name__ == "__main__":
    with open(sys.argv[1]) as f:
        src = f.read()
    name = os.path.basename(sys.argv[1])
    out = patch_mcmc(src) if name == "mcmc.c" else patch_mbbeagle(src)
    with open(sys.argv[2], "w") as f:
        f.write(out)
