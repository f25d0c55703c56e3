/*
 * mbamd_reports_glue.h -- what a MrBayes maintainer adds so that divisions which report ancestral states, site rates,
 * positively selected sites or site omegas, and covarion divisions, stay on the engine.  The
 * reference switches BEAGLE off for them (src/mcmc.c:5760-5771): the BEAGLE API has no final pass and its covarion branch
 * in src/mbbeagle.c is unfinished.  include/libhmsbeagle/mbamd_reports.h supplies the final pass (CondLikeUp_*) and a scaled
 * read-out; this file is the MrBayes side.  Our code against the reference's public types; no reference source in it.
 * integration/mrbayes/patches/patch_reports.py applies the edits below to temporary copies of src/mcmc.c and src/mbbeagle.c
 * (oracle/Makefile: ref-amd-reports); INTEGRATION.md, "Reports and covarion".
 *
 *   src/mcmc.c, InitChainCondLikes, in front of the test that refuses BEAGLE (:5761):
 *       if (MbamdEngineServes (m) == YES) m->useBeagle = YES; else if (MbamdEngineRefuses (m, d+1) == YES) ; else
 *   src/mcmc.c, PrintStates / PrintStatesToFiles: every call of m->PrintSiteRates / m->PosSelProbs / m->SiteOmegas is
 *       preceded by MbamdReportsRoot (node, d, coldId), and in the final-pass loop (:13148)
 *       if (MbamdReportsUp (tree, node, d, coldId) == NO) m->CondLikeUp (node, d, coldId);
 *   src/mbbeagle.c, covarion (one eigen-system per rate category = one "part" with ONE category each, like the omega
 *       classes of a codon model): the instance is created with one category, tips are partials, rates and weights per part.
 */
#ifndef MBAMD_REPORTS_GLUE_H_
#define MBAMD_REPORTS_GLUE_H_

/* YES: the division asks for one of the reports (or is a covarion model) and the engine serves it -- keep m->useBeagle.
 * NO: leave the reference's decision alone (not requested, Gibbs-sampled rates, double precision, MBAMD_DEVICE_REPORTS=0). */
int     MbamdEngineServes (ModelInfo *m);
/* YES: the division must NOT go to the BEAGLE path although the reference would send it there, and says why (one printed line):
 * rates=adgamma -- the autocorrelated-gamma HMM (CalcLikeAdgamma, src/mcmc.c:1575) reads per-category site likelihoods
 * (m->rateProbs) that only the host's Likelihood_Adgamma fills; on the BEAGLE path nobody does (src/mcmc.c:7452-7478), and the
 * chain would run on a wrong likelihood without a word. */
int     MbamdEngineRefuses (ModelInfo *m, int divisionNumber);
/* YES if MbamdEngineServes answers YES for any current division (BEAGLE v3 builds: such an analysis keeps one instance per
   division instead of one multi-partition instance, src/mcmc.c:5846-5856) */
int     MbamdEngineServesAny (void);
/* Before m->PrintSiteRates / m->PosSelProbs / m->SiteOmegas (they read the TOP interior node's conditional likelihoods and
 * the site scalers from host arrays): on an engine division, compute the top node's 3-way product on the device and
 * materialise it -- and the site scalers -- in the host arrays those functions read.  NO_ERROR / ERROR. */
int     MbamdReportsRoot (TreeNode *top, int division, int chain);
/* In the final-pass loop of PrintStatesToFiles (called for every interior node, top node first): on an engine division the
 * whole pass runs on the device when the top node comes by, and the final conditional likelihoods of the locked nodes land in
 * the host arrays m->PrintAncStates reads; returns YES (skip m->CondLikeUp).  NO: not an engine division. */
int     MbamdReportsUp (Tree *t, TreeNode *node, int division, int chain);

#endif
