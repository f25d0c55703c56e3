/*
 * mbamd_std_glue.h -- standard (morphology) data on the engine: the MrBayes side.
 *
 * The reference keeps divisions of datatype=standard on its host kernels (CondLikeDown_Std / CondLikeRoot_Std /
 * CondLikeScaler_Std / Likelihood_Std, src/likelihood.c:1920, 4496, 5547, 7359; selected at src/mcmc.c:18296-18306; never
 * handed to BEAGLE, src/mcmc.c:5741-5771): the characters of one division have DIFFERENT state counts (2 ... 10, ordered or
 * unordered), which the BEAGLE API -- one state count per instance -- cannot express.  This binding groups the compressed
 * characters of a division by their transition-matrix class (state count x ordered / unordered: exactly the classes the
 * reference's TiProbs_Std fills, src/likelihood.c:10066) and gives every class its own engine instance.  Everything the reference
 * computes per evaluation for such a division runs on the device: the transition matrices (a class's rate matrix has a closed-form
 * eigen-system -- unordered: the complete graph, ordered: the path, binary with unequal frequencies: two states -- which the engine
 * exponentiates per touched branch from its length, beagleUpdateTransitionMatrices; TiProbs_Std's term-by-term closed forms are
 * those exponentials), conditional likelihoods, rescaling and the root integration; the correction for unobservable patterns
 * (coding=variable / informative: the dummy characters in front of the division) is the reference's own formula over the site
 * values the engine returns, as for restriction sites (src/mbbeagle.c:1322-1358).
 *
 * Served: equal state frequencies (symdirihyperpr=fixed(infinity), the default: SYMPI_EQUAL), any rate-category count; unequal
 * frequencies (any other symdirihyperpr) of BINARY characters: the beta categories of the symmetric-beta prior are a mixture, one
 * buffer set ("part") per category inside the class's instance, mixed by the integration (src/likelihood.c:1942-1955, 7441-7470).
 * Ancestral states (report ancstates=yes, equal frequencies): the reference's own final pass and read-out run on host arrays that
 * MbamdStdMaterialise fills from the device once per printed sample.
 * Not served (the division stays on the host kernels, with a printed reason): unequal frequencies of characters with more than two
 * states (an eigen-system per CHARACTER, src/likelihood.c:10426-10455), irreversible characters; report siterates.
 *
 * Hooks (applied to temporary copies of src/likelihood.c and src/mcmc.c by integration/mrbayes/patches/patch_std.py):
 *     LaunchLogLikeForDivision:   if (MbamdStdServes (m) == YES) { MbamdStdLogLike (chain, d, lnL); return; }
 *     PrintStates, in front of the final-pass loop of a division that reports ancestral states:   MbamdStdMaterialise (coldId, d);
 *     FreeChainMemory, first statement:   MbamdStdFinalize ();     (instances and class tables are per analysis, like the reference's)
 */
#ifndef MBAMD_STD_GLUE_H_
#define MBAMD_STD_GLUE_H_

#include "bayes.h"

int  MbamdStdServes (ModelInfo *m);                    /* YES: this division's likelihood is computed by MbamdStdLogLike */
void MbamdStdLogLike (int chain, int d, MrBFlt *lnL);  /* replaces the host pass of LaunchLogLikeForDivision for such a division */
void MbamdStdFinalize (void);                          /* frees the engine instances and every table of the binding (FreeChainMemory, atexit) */
int  MbamdStdMaterialise (int chain, int d);           /* report ancstates: fills the division's host arrays from the device (YES) or does nothing (NO) */

#endif
