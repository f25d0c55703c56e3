/*
 * mbamd_reports_glue.c -- the MrBayes side of the engine's final pass / scaled read-out (see mbamd_reports_glue.h).
 * Compiled and linked with the reference's own sources; our code, no reference source in it.
 *
 * The reference's read-outs (PrintAncStates_*, PrintSiteRates_Gen, PosSelProbs, SiteOmegas) are kept as they are: they
 * read conditional likelihoods [category][pattern][state] and natural-log site scalers from host arrays that a BEAGLE
 * division never allocates (src/mcmc.c:5970-6040).  This file allocates the few rows they touch, lets the device compute
 * what goes into them, and points the division's scalar read-out functions at them.
 */
#include "bayes.h"
#include "mcmc.h"
#include "model.h"
#include "utils.h"
#include "libhmsbeagle/mbamd_reports.h"
#include "mbamd_reports_glue.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if !defined (BEAGLE_ENABLED)
#error "the reports binding needs the BEAGLE build (m->useBeagle, m->beagleInstance)"
#endif

/* scalar read-outs of the reference (src/mcmc.c:10108, 10267; external linkage, no prototype in its headers) */
int PosSelProbs (TreeNode *p, int division, int chain);
int SiteOmegas (TreeNode *p, int division, int chain);

static float    *partBuf = NULL, *lnBuf = NULL;
static size_t   partCap = 0, lnCap = 0;

static void Die (const char *what)
{
    fprintf (stderr, "mbamd reports: %s (%s)\n", what, mbamdGetLastError());
    exit (1);
}

static int EnvOff (void)
{
    const char *s = getenv("MBAMD_DEVICE_REPORTS");
    return (s != NULL && s[0] == '0') ? YES : NO;
}

int MbamdEngineRefuses (ModelInfo *m, int divisionNumber)
{
    if (m->correlation == NULL)
        return (NO);
    MrBayesPrint ("%s   Non-beagle version of conditional likelihood calculator will be used for division %d: the\n", spacer, divisionNumber);
    MrBayesPrint ("%s   autocorrelated gamma model (rates=adgamma) reads per-category site likelihoods the BEAGLE path does not fill.\n", spacer);
    return (YES);
}

int MbamdEngineServes (ModelInfo *m)
{
    if (EnvOff () == YES || m->gibbsGamma == YES || m->correlation != NULL)
        return (NO);
    if ((beagleFlags & BEAGLE_FLAG_PRECISION_DOUBLE) != 0)
        return (NO);                /* the double-precision engine has no final pass */
    if (m->printAncStates == YES || m->printSiteRates == YES || m->printPosSel == YES || m->printSiteOmegas == YES)
        return (YES);
    if (m->switchRates != NULL)
        return (YES);
    return (NO);
}

int MbamdEngineServesAny (void)
{
    int d;
    for (d=0; d<numCurrentDivisions; d++)
        if (MbamdEngineServes (&modelSettings[d]) == YES)
            return (YES);
    return (NO);
}

/* the host arrays the reference's read-outs index: pointer tables over every buffer index, rows allocated when first needed */
static int Parts (ModelInfo *m)
{
    return (m->nCijkParts > 1 ? m->nCijkParts : 1);
}

static void Tables (ModelInfo *m)
{
    if (m->condLikes == NULL)
        {
        m->condLikes = (CLFlt **) SafeCalloc ((size_t) m->numCondLikes * Parts(m) + 1, sizeof(CLFlt *));
        if (!m->condLikes)
            Die ("out of memory");
        }
    if (m->scalers == NULL)
        {
        m->scalers = (CLFlt **) SafeCalloc ((size_t) m->numScalers * Parts(m) + 1, sizeof(CLFlt *));
        if (!m->scalers)
            Die ("out of memory");
        }
    if (m->clP == NULL)
        {
        m->clP = (CLFlt **) SafeCalloc ((size_t) (m->numTiCats > 0 ? m->numTiCats : m->numRateCats * m->numOmegaCats) + 1, sizeof(CLFlt *));
        if (!m->clP)
            Die ("out of memory");
        }
    /* the vector variants read another layout: a division on the engine uses the scalar read-outs */
    m->useVec = VEC_NONE;
    if (m->PosSelProbs != NULL)
        m->PosSelProbs = &PosSelProbs;
    if (m->SiteOmegas != NULL)
        m->SiteOmegas = &SiteOmegas;
}

static CLFlt *Row (CLFlt **table, int index, size_t n)
{
    if (table[index] == NULL)
        {
        table[index] = (CLFlt *) SafeCalloc (n, sizeof(CLFlt));
        if (!table[index])
            Die ("out of memory");
        }
    return table[index];
}

/* Device buffers `first` .. `first + parts - 1` (one per eigen-system part: omega classes, covarion rate categories; each
 * with `cats` categories on the device) -> the host row [part * cats + category][pattern][state] at ONE scale per pattern,
 * the common natural-log scaler of the pattern -> lnScaler (max over the parts; exact ratios are powers of two). */
static void Download (ModelInfo *m, int first, int firstCum, CLFlt *row, CLFlt *lnScaler)
{
    int         n, c, parts = Parts(m), cats, S = m->numModelStates, P = m->numChars;
    size_t      i, per;
    float       *dst, f;

    cats = (m->switchRates != NULL) ? 1 : m->numRateCats;        /* (covarion: one category per part, see patch_reports.py) */
    per = (size_t) cats * P * S;
    if (per * parts > partCap)
        {
        partCap = per * parts;
        partBuf = (float *) realloc (partBuf, partCap * sizeof(float));
        }
    if ((size_t) P * parts > lnCap)
        {
        lnCap = (size_t) P * parts;
        lnBuf = (float *) realloc (lnBuf, lnCap * sizeof(float));
        }
    if (!partBuf || !lnBuf)
        Die ("out of memory");
    for (n=0; n<parts; n++)
        if (mbamdGetScaledPartials (m->beagleInstance, first + n, firstCum + n, partBuf + per * n, lnBuf + (size_t) P * n) != BEAGLE_SUCCESS)
            Die ("mbamdGetScaledPartials failed");
    for (c=0; c<P; c++)
        {
        f = lnBuf[c];
        for (n=1; n<parts; n++)
            if (lnBuf[(size_t) P * n + c] > f)
                f = lnBuf[(size_t) P * n + c];
        lnScaler[c] = (CLFlt) f;
        }
    for (n=0; n<parts; n++)
        {
        dst = partBuf + per * n;
        for (c=0; c<P; c++)
            {
            f = (float) exp ((double) lnBuf[(size_t) P * n + c] - (double) lnScaler[c]);
            if (f != 1.0f)
                {
                int k;
                for (k=0; k<cats; k++)
                    for (i=0; i<(size_t) S; i++)
                        dst[((size_t) k * P + c) * S + i] *= f;
                }
            }
        for (i=0; i<per; i++)
            row[per * n + i] = (CLFlt) dst[i];
        }
}

/* the final pass from the top node down to every node with want[p->index] set (its ancestors included by the caller) */
static void FinalPass (Tree *t, int division, int chain, const char *want)
{
    int                     i, n, count = 0, parts;
    TreeNode                *p;
    ModelInfo               *m = &modelSettings[division];
    MbamdFinalOperation     *ops;

    parts = Parts(m);
    ops = (MbamdFinalOperation *) SafeCalloc ((size_t) t->nIntNodes * parts + 1, sizeof(MbamdFinalOperation));
    if (!ops)
        Die ("out of memory");
    for (i=t->nIntNodes-1; i>=0; i--)                       /* root-ward nodes first (reference src/mcmc.c:13148) */
        {
        p = t->intDownPass[i];
        if (want[p->index] == 0)
            continue;
        for (n=0; n<parts; n++)
            {
            ops[count].destinationPartials = m->condLikeScratchIndex[p->index] + n;
            ops[count].downPartials = m->condLikeIndex[chain][p->index] + n;
            ops[count].transitionMatrix = m->tiProbsIndex[chain][p->index] + n;
            if (p->anc->anc == NULL)
                {
                ops[count].ancestorFinal = -1;
                ops[count].rootTip = (t->isRooted == NO) ? m->condLikeIndex[chain][p->anc->index] : -1;
                }
            else
                {
                ops[count].ancestorFinal = m->condLikeScratchIndex[p->anc->index] + n;
                ops[count].rootTip = -1;
                }
            count++;
            }
        }
    if (mbamdUpdateFinalPartials (m->beagleInstance, ops, count) != BEAGLE_SUCCESS)
        Die ("mbamdUpdateFinalPartials failed");
    free (ops);
}

/* MBAMD_REPORTS_CHECK=1: the log-likelihood recomputed on the host from what was just materialised for the top node (row,
   site scalers) must be the division's current log-likelihood -- the reference's own formula (Likelihood_Gen / _NY98,
   src/likelihood.c:5764-5917, 6975-7040: category weights, state frequencies, the invariable-sites term, site scalers). */
static long nChecked = 0;

static void ReportChecks (void)
{
    fprintf (stderr, "mbamd reports check: %ld top-node read-outs reproduced the division's log-likelihood\n", nChecked);
}

static void CheckRoot (ModelInfo *m, int division, int chain, const CLFlt *row, const CLFlt *lnScaler)
{
    int         c, k, i, S = m->numModelStates, P = m->numChars, nk;
    MrBFlt      *bs, covBF[64], *swr, probOn, freq, pInvar = 0.0, lnL = 0.0, like, cat, likeI, want, *omegaCatFreq = NULL;
    const CLFlt *clInvar = NULL;
    CLFlt       *nSites = numSitesOfPat + m->compCharStart;      /* (weights of the original data: chain id 0 scheme) */

    bs = GetParamSubVals (m->stateFreq, chain, state[chain]);
    if (m->switchRates != NULL)
        {
        swr = GetParamVals (m->switchRates, chain, state[chain]);
        probOn = swr[0] / (swr[0] + swr[1]);
        for (i=0; i<S/2; i++)
            {
            covBF[i] = bs[i] * probOn;
            covBF[i+S/2] = bs[i] * (1.0 - probOn);
            }
        bs = covBF;
        }
    if (m->pInvar != NULL)
        {
        pInvar = *GetParamVals (m->pInvar, chain, state[chain]);
        clInvar = m->invCondLikes;
        }
    nk = m->numRateCats * m->numOmegaCats;
    freq = (1.0 - pInvar) / m->numRateCats;
    if (m->numOmegaCats > 1)
        omegaCatFreq = GetParamSubVals (m->omega, chain, state[chain]);
    for (c=0; c<P; c++)
        {
        like = 0.0;
        for (k=0; k<nk; k++)
            {
            cat = 0.0;
            for (i=0; i<S; i++)
                cat += (MrBFlt) row[((size_t) k * P + c) * S + i] * bs[i];
            like += cat * (omegaCatFreq != NULL ? omegaCatFreq[k / m->numRateCats] * freq : freq);
            }
        if (clInvar != NULL)
            {
            likeI = 0.0;
            for (i=0; i<S; i++)
                likeI += (MrBFlt) clInvar[(size_t) c * S + i] * bs[i];
            if (lnScaler[c] > -200.0)
                like += likeI * pInvar / exp ((MrBFlt) lnScaler[c]);
            }
        lnL += ((MrBFlt) lnScaler[c] + log (like)) * (MrBFlt) nSites[c];
        }
    want = m->lnLike[2*chain + state[chain]];
    if (!(fabs (lnL - want) <= 2e-5 * fabs (want)))
        {
        fprintf (stderr, "mbamd reports check: division %d: the top node's read-out gives lnL = %.9f, the division's is %.9f\n", division + 1, lnL, want);
        exit (1);
        }
    if (nChecked++ == 0)
        atexit (ReportChecks);
}

int MbamdReportsRoot (TreeNode *top, int division, int chain)
{
    char        *want;
    Tree        *t;
    ModelInfo   *m = &modelSettings[division];

    if (m->useBeagle == NO)
        return (NO_ERROR);
    Tables (m);
    t = GetTree (m->brlens, chain, state[chain]);
    want = (char *) SafeCalloc ((size_t) t->nNodes + 1, 1);
    if (!want)
        Die ("out of memory");
    want[top->index] = 1;
    FinalPass (t, division, chain, want);
    free (want);
    /* PrintSiteRates_Gen / PosSelProbs / SiteOmegas read the top node under its LIVE index: that row gets the 3-way product */
    Download (m, m->condLikeScratchIndex[top->index], m->siteScalerIndex[chain],
              Row (m->condLikes, m->condLikeIndex[chain][top->index], (size_t) m->condLikeLength),
              Row (m->scalers, m->siteScalerIndex[chain], (size_t) m->numChars));
    {
    const char *chk = getenv("MBAMD_REPORTS_CHECK");
    if (chk != NULL && chk[0] != '0' && m->dataType != RESTRICTION)
        CheckRoot (m, division, chain, m->condLikes[m->condLikeIndex[chain][top->index]], m->scalers[m->siteScalerIndex[chain]]);
    }
    /* (PrintSiteRates_Gen uses the top node's scratch row as work space) */
    (void) Row (m->condLikes, m->condLikeScratchIndex[top->index], (size_t) m->condLikeLength);
    return (NO_ERROR);
}

int MbamdReportsUp (Tree *t, TreeNode *node, int division, int chain)
{
    int         i;
    char        *want;
    TreeNode    *p, *q;
    ModelInfo   *m = &modelSettings[division];

    if (m->useBeagle == NO)
        return (NO);
    if (node->anc->anc != NULL)
        return (YES);                               /* everything was done when the top node came by */
    Tables (m);
    want = (char *) SafeCalloc ((size_t) t->nNodes + 1, 1);
    if (!want)
        Die ("out of memory");
    for (i=0; i<t->nIntNodes; i++)                  /* the locked nodes and their root-ward paths */
        {
        p = t->intDownPass[i];
        if (p->isLocked == NO)
            continue;
        for (q=p; q->anc != NULL && want[q->index] == 0; q=q->anc)
            want[q->index] = 1;
        }
    want[node->index] = 1;
    FinalPass (t, division, chain, want);
    free (want);
    for (i=0; i<t->nIntNodes; i++)
        {
        p = t->intDownPass[i];
        if (p->isLocked == NO)
            continue;
        Download (m, m->condLikeScratchIndex[p->index], m->siteScalerIndex[chain],
                  Row (m->condLikes, m->condLikeScratchIndex[p->index], (size_t) m->condLikeLength),
                  Row (m->scalers, m->siteScalerIndex[chain], (size_t) m->numChars));
        }
    return (YES);
}
