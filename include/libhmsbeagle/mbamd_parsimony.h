/*
 * mbamd_parsimony.h -- C ABI of the MI355X Fitch-parsimony scorer: the step either side of the
 * likelihood path that bounds MrBayes' default move mix.  Exported by the same libhmsbeagle.so as the BEAGLE ABI.
 *
 * What it replaces in the reference (NBISweden/MrBayes; all of it host loops over `m->parsSets`, one BitsLong word
 * -- two for >= 64 states -- per site pattern and tree node, src/mcmc.c:6840-6895):
 *
 *   mbamdParsDownPass    GetParsDP -> GetFitchPartials                      src/mcmc.c:4849-4876, 4794-4846
 *                        (also the node loop of Likelihood_Pars             src/likelihood.c:7617-7680, and
 *                         GetParsimonySubtreeRootstate's up-pass            src/mcmc.c:5076-5150: the same set operation)
 *   mbamdParsFinalPass   GetParsFP                                          src/mcmc.c:4881-4954
 *   mbamdParsScore       the candidate loops of the parsimony-biased moves  src/proposal.c:10783-10876 (ParsSPR1),
 *                        13430-13472 (ParsTBR1), 10240-10290 (ParsSPR), 11470-11560, 12190-12220, 12850-12880, 13970-14000;
 *                        GetParsimonyBrlens / GetParsimonyLength tails      src/mcmc.c:4978-5010, 5030-5070;
 *                        per-node lengths of Likelihood_Pars                src/likelihood.c:7643-7676
 *
 * One parsimony instance = the state sets of one data division, resident in HBM in the narrowest unsigned type that
 * holds `setBits` bits (u8 for DNA, u32 for amino acids, u64 for codons, 2 x u64 beyond).  Sets keep their values
 * between calls exactly like m->parsSets does (the reference's final pass reads the set of a clipped-out node as
 * whatever an earlier move left there; a mirror that forgot it would propose differently).
 *
 * All functions return BEAGLE_SUCCESS (0) or a negative BEAGLE_ERROR_* code; mbamdGetLastError() has the text.
 * Calls are stream-ordered and return without waiting unless they hand back a value.
 * The binding a MrBayes maintainer adds is integration/mrbayes/mbamd_pars_glue.c.
 */
#ifndef MBAMD_LIBHMSBEAGLE_PARSIMONY_H_
#define MBAMD_LIBHMSBEAGLE_PARSIMONY_H_

#include "libhmsbeagle/beagle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* setCount sets (m->numParsSets) of patternCount site patterns (m->numChars), wordsPerSet 64-bit words each
 * (m->nParsIntsPerSite: 1 or 2), of which the low setBits bits can be set (the division's state count).
 * likelihoodInstance >= 0: live on the same GPU as that BEAGLE instance; -1: device 0.  Returns the handle (>= 0). */
BEAGLE_DLLEXPORT int mbamdParsCreateInstance(int setCount, int patternCount, int wordsPerSet, int setBits, int likelihoodInstance);
BEAGLE_DLLEXPORT int mbamdParsFinalizeInstance(int pars);

/* one set <-> the host's BitsLong array m->parsSets[setIndex]: patternCount * wordsPerSet words (tips: InitParsSets,
 * src/mcmc.c:6897-7040) */
BEAGLE_DLLEXPORT int mbamdParsSetSets(int pars, int setIndex, const unsigned long long* sets);
BEAGLE_DLLEXPORT int mbamdParsGetSets(int pars, int setIndex, unsigned long long* outSets);

/* numSitesOfPat of the chain (CLFlt = float, src/mcmc.c:265): patternCount weights.  Re-sending the same values is free. */
BEAGLE_DLLEXPORT int mbamdParsSetPatternWeights(int pars, const float* weights);

/* Fitch down-pass over `count` operations in the given (post-)order: ops[4*i..] = { destination, source1, source2,
 * unused(-1) };  x = S1 & S2, and where that is empty x = S1 | S2 and the pattern's weight is added to the length.
 * outLength: NULL = do not wait (the parsimony moves ignore GetParsDP's value), else the summed length. */
BEAGLE_DLLEXPORT int mbamdParsDownPass(int pars, const int* ops, int count, double* outLength);

/* Final-pass (GetParsFP) over `count` nodes in pre-order: ops[4*i..] = { node, left, right, ancestor }.  The node's
 * down-pass set is overwritten by its final set, exactly as the reference does in place. */
BEAGLE_DLLEXPORT int mbamdParsFinalPass(int pars, const int* ops, int count);

/* Lengths of `count` candidate positions: tuples[4*i..] = { a, b, c, d }, a set index or -1 (empty set) each:
 *     outLengths[i] = sum over patterns k of weight[k] * [ ((A | B) & (C | D)) == 0 ]
 * (a,b,p,-1) is ParsSPR1's root-side case, (p,-1,c,d) its crown-side case, (a,-1,b,-1) a node length. */
BEAGLE_DLLEXPORT int mbamdParsScore(int pars, const int* tuples, int count, double* outLengths);

#ifdef __cplusplus
}
#endif
#endif
