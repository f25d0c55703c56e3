/*
 * mbamd_reports.h -- engine extension: the final ("up") pass over conditional likelihoods and their scaled read-out.
 *
 * What MrBayes needs when a run reports ancestral states, site rates, positively selected sites or site omegas
 * (`report ancstates / siterates / possel / siteomega`): reference CondLikeUp_Bin / _Gen / _NUC4 (src/likelihood.c:4574-4795)
 * and the inputs of PrintAncStates_*, PrintSiteRates_Gen, PosSelProbs, SiteOmegas (src/mcmc.c:10108-11070, 12212).  The
 * reference switches BEAGLE off for such divisions (src/mcmc.c:5760-5765); with this extension and the binding in
 * integration/mrbayes/mbamd_reports_glue.c ("Reports and covarion") they stay on the engine.
 * SURVEY 8(f) row 3.  Same conventions as beagle.h: plain pointers and sizes in, status codes out.
 */
#ifndef MBAMD_REPORTS_ABI_H_
#define MBAMD_REPORTS_ABI_H_

#include "libhmsbeagle/beagle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One step of the final pass: the "final" conditional likelihoods of a node from those of its ancestor
 * (reference CondLikeUp_*: `clFP` from `clFA`, `clDP`, `tiP`).  All indices are partials / matrix buffer indices of the
 * instance; the destination is any partials buffer the client owns (MrBayes: m->condLikeScratchIndex[node]).
 *   ancestorFinal >= 0 : destination = up-pass of (ancestorFinal, downPartials, transitionMatrix)
 *   ancestorFinal <  0 : the top interior node: destination = downPartials, times -- rootTip >= 0, unrooted trees -- the
 *                        factor of the tip across transitionMatrix (what CondLikeRoot_* includes natively,
 *                        src/likelihood.c:2152-4500; BEAGLE mode integrates that branch at the end instead) */
typedef struct {
    int destinationPartials;
    int ancestorFinal;
    int downPartials;
    int transitionMatrix;
    int rootTip;
} MbamdFinalOperation;

/* Operations run in list order (an ancestor's final partials must be listed before its descendants'). */
BEAGLE_DLLEXPORT int mbamdUpdateFinalPartials(int instance, const MbamdFinalOperation* operations, int operationCount);

/* Read a partials buffer the way the reference's read-outs expect it: outPartials[k][pattern][state] (floats, categories
 * slowest: reference src/likelihood.c:860-876) with all categories of a pattern at ONE common scale, and that scale as the
 * natural-log site scaler outLnScale[pattern] (true value = outPartials * exp(outLnScale); reference `lnScaler`).
 * cumulativeScaleIndex: the cumulative scale buffer of the tree the buffer belongs to (BEAGLE_OP_NONE: unscaled).
 * Final partials of every node, and the top node's, carry exactly the tree's cumulative factor (see csrc/mbamd_reports.h). */
BEAGLE_DLLEXPORT int mbamdGetScaledPartials(int instance, int bufferIndex, int cumulativeScaleIndex, float* outPartials,
                                            float* outLnScale);

#ifdef __cplusplus
}
#endif
#endif
