/*
 * mbamd_pars_glue.h -- what a MrBayes maintainer adds to src/proposal.c to run the Fitch-parsimony scoring of the
 * parsimony-biased moves on the GPU (include/libhmsbeagle/mbamd_parsimony.h).  See INTEGRATION.md, "Device parsimony".
 * Our code, written against the reference's public types (src/bayes.h); it contains no reference source.
 *
 *   #include "mbamd_pars_glue.h"                      after proposal.c's own includes
 *   #define GetParsDP MbamdGetParsDP                  around the parsimony-biased moves (Move_ParsSPR1 and Move_ParsTBR1 are in
 *   #define GetParsFP MbamdGetParsFP                  the default mix, src/model.c:22628, 22755), #undef'd after them
 *   the candidate loop ("cycle through the possibilities and record the parsimony length", src/proposal.c:10782-10884,
 *   13429-13474) is bracketed:
 *       if (MbamdParsActive (t) == YES && MbamdParsLengths (...) == ERROR) goto errorExit;
 *       if (MbamdParsHostToo (t, parLength, nRoot*nCrown) == YES) { ...the reference's loop, unchanged... }
 *       MbamdParsCompare (parLength, nRoot*nCrown);
 * integration/mrbayes/patches/patch_pars.py applies exactly this to a temporary copy of proposal.c when oracle/Makefile builds
 * _ref/mb_amd_pars and _ref/mb_emu_pars.
 */
#ifndef MBAMD_PARS_GLUE_H_
#define MBAMD_PARS_GLUE_H_

/* YES when every division of the tree runs on the engine (m->useBeagle) and MBAMD_DEVICE_PARSIMONY is not 0 */
int     MbamdParsActive (Tree *t);
/* GetParsDP / GetParsFP (src/mcmc.c:4849, 4881) on the device sets; fall back to the host functions when not active.
 * MbamdGetParsDP returns 0.0 on the device: both moves discard the value (src/proposal.c:10702-10743, 13396-13410). */
MrBFlt  MbamdGetParsDP (Tree *t, TreeNode *p, int chain);
void    MbamdGetParsFP (Tree *t, TreeNode *p, int chain);
/* parLength[i + j*nRoot] = sum over divisions of warpFactor * length of candidate (pRoot[i], pCrown[j]).
 * kind 0: ParsSPR1 moving in the root part   ((A | B) & P : A = pRoot[i], B = its ancestor, P = v)
 * kind 1: ParsSPR1, u has no ancestor        (P & (C | D) : P = u, C = pCrown[j], D = its ancestor)
 * kind 2: ParsSPR1 moving in the crown part  ((A | B) & (C | D) : A = a, B = b)
 * kind 3: ParsTBR1                           ((A | B) & (C | D) : A = pRoot[i], B = its ancestor) */
int     MbamdParsLengths (Tree *t, int chain, int kind, TreeNode **pRoot, int nRoot, TreeNode **pCrown, int nCrown,
                          TreeNode *a, TreeNode *b, TreeNode *u, TreeNode *v, CLFlt *nSitesOfPat, MrBFlt warpFactor,
                          MrBFlt *parLength);
/* The node-length shape of Move_ParsSPR, Move_ParsSPRClock and Move_ParsSPRClock_Fossil (src/proposal.c:10241-10296,
 * 12179-12234, 12840-12895): for every marked node p the length of (P | A) & V, A the node's ancestor, V = v.
 * MbamdParsMarkedLengths computes them all in one device call before the node loop; MbamdParsMarkedLength hands one out
 * (YES: `*length` is set, skip the host loop; NO: run it -- not active, or check mode); MbamdParsMarkedCheck compares. */
void    MbamdParsMarkedLengths (Tree *t, int chain, TreeNode *v, CLFlt *nSitesOfPat);
int     MbamdParsMarkedLength (Tree *t, int n, TreeNode *p, MrBFlt *length);
void    MbamdParsMarkedCheck (Tree *t, int n, TreeNode *p, MrBFlt length);
/* YES: run the reference's host loop as well (device not active, or MBAMD_PARS_CHECK=1: the device result is kept aside) */
int     MbamdParsHostToo (Tree *t, MrBFlt *parLength, int n);
/* MBAMD_PARS_CHECK=1: compare what the host loop just wrote with the device result; abort on any difference */
void    MbamdParsCompare (MrBFlt *parLength, int n);

#endif
