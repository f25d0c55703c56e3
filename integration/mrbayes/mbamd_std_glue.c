/*
 * mbamd_std_glue.c -- standard (morphology) data on the engine: the MrBayes side (see mbamd_std_glue.h).
 * Compiled and linked with the reference's own sources; our code, no reference source in it.
 *
 * What the reference does per evaluation for such a division (LaunchLogLikeForDivision, src/likelihood.c:7851-7972: flip the
 * index tables, TiProbs_Std per touched branch, CondLikeDown_Std / CondLikeRoot_Std per touched node, CondLikeScaler_Std,
 * Likelihood_Std) is kept as a FLOW -- the same index tables (condLikeIndex, tiProbsIndex, nodeScalerIndex, siteScalerIndex)
 * name the engine's buffers, so accept / reject (ResetFlips, src/mcmc.c:15695) needs no knowledge of the binding -- while the
 * arithmetic on conditional likelihoods runs on the device, one engine instance per transition-matrix class of the division.
 */
#include "bayes.h"
#include "mcmc.h"
#include "model.h"
#include "utils.h"
#include "libhmsbeagle/beagle.h"
#include "libhmsbeagle/mbamd_reports.h"
#include "mbamd_std_glue.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if !defined (BEAGLE_ENABLED)
#error "the standard-data binding needs the BEAGLE build (the engine is linked as libhmsbeagle)"
#endif

/* index-table flips of the reference (src/likelihood.c:5625-5683; external linkage, prototypes only in its .c files) */
void FlipCondLikeSpace (ModelInfo *m, int chain, int nodeIndex);
void FlipNodeScalerSpace (ModelInfo *m, int chain, int nodeIndex);
void FlipSiteScalerSpace (ModelInfo *m, int chain);
void FlipTiProbsSpace (ModelInfo *m, int chain, int nodeIndex);

extern int  *chainId;               /* (src/mcmc.c; the reference's likelihood.c declares it the same way) */
#define LIKE_EPSILON 1.0e-300       /* src/likelihood.c:44 */
#define MBAMD_STD_MAXCLASSES (3 * MAX_STD_STATES)

typedef struct
    {
    int     instance;       /* engine instance of this class */
    int     nStates;        /* states of its characters */
    int     tiIndex;        /* offset of the class's matrices [category][from][to] inside a branch's tiProbs array */
    int     nChars;         /* compressed characters of the class ... */
    int     *chars;         /* ... and which (division-local indices, ascending) */
    } StdClass;

typedef struct
    {
    int         ready;      /* 0: not looked at, 1: served, -1: refused */
    int         nClasses;
    StdClass    cls[MBAMD_STD_MAXCLASSES];
    BeagleOperation *ops;
    int         *scaleIdx;
    double      *mat, *site, *lnSite;
    int         *off;       /* ancestral states: first state of character c inside one rate category of a host row ... */
    int         numReps;    /* ... and the states of all characters together */
    double      *part;      /* a node's conditional likelihoods of one class [category][character][state] */
    float       *partF, *lnScaleF;
    } StdDivision;

static StdDivision  *stdDiv = NULL;
static int          stdDivCount = 0;

static void Die (const char *what)
{
    fprintf (stderr, "mbamd standard data: %s\n", what);
    exit (1);
}

static int EnvOff (void)
{
    const char *s = getenv("MBAMD_DEVICE_STD");
    return (s != NULL && s[0] == '0') ? YES : NO;
}

void MbamdStdFinalize (void)
{
    int d, g;
    for (d=0; d<stdDivCount; d++)
        for (g=0; stdDiv != NULL && g<stdDiv[d].nClasses; g++)
            if (stdDiv[d].cls[g].instance >= 0)
                {
                beagleFinalizeInstance (stdDiv[d].cls[g].instance);
                stdDiv[d].cls[g].instance = -1;
                }
}

/* the classes of a division, their instances, tip data, frequencies and weights: once per division */
static int Setup (ModelInfo *m, int d)
{
    int             c, g, i, j, s, n, nSet, last, ambiguous, rc, *states, resource;
    double          *partials, *freqs, *w;
    BitsLong        *bits;
    StdDivision     *sd = &stdDiv[d];
    StdClass        *cl;
    BeagleInstanceDetails details;
    Tree            *t = GetTree (m->brlens, 0, 0);

    sd->nClasses = 0;
    for (c=0; c<m->numChars; c++)
        {
        for (g=0; g<sd->nClasses; g++)
            if (sd->cls[g].tiIndex == m->tiIndex[c])
                break;
        if (g == sd->nClasses)
            {
            if (g == MBAMD_STD_MAXCLASSES)
                return (ERROR);
            sd->cls[g].tiIndex = m->tiIndex[c];
            sd->cls[g].nStates = m->nStates[c];
            sd->cls[g].nChars = 0;
            sd->cls[g].chars = (int *) SafeCalloc (m->numChars, sizeof(int));
            sd->cls[g].instance = -1;
            sd->nClasses++;
            }
        if (sd->cls[g].nStates != m->nStates[c])
            return (ERROR);                                 /* (one matrix class, two state counts: not the layout this file assumes) */
        sd->cls[g].chars[sd->cls[g].nChars++] = c;
        }
    sd->ops = (BeagleOperation *) SafeCalloc (t->nIntNodes + 1, sizeof(BeagleOperation));
    sd->scaleIdx = (int *) SafeCalloc (t->nIntNodes + 1, sizeof(int));
    sd->mat = (double *) SafeCalloc ((size_t) m->numRateCats * MAX_STD_STATES * MAX_STD_STATES, sizeof(double));
    sd->site = (double *) SafeCalloc (m->numChars + 1, sizeof(double));
    sd->lnSite = (double *) SafeCalloc (m->numChars + 1, sizeof(double));
    states = (int *) SafeCalloc (m->numChars + 1, sizeof(int));
    partials = (double *) SafeCalloc ((size_t) (m->numChars + 1) * MAX_STD_STATES, sizeof(double));
    freqs = (double *) SafeCalloc (MAX_STD_STATES, sizeof(double));
    w = (double *) SafeCalloc (m->numRateCats + m->numChars + 1, sizeof(double));
    if (!sd->ops || !sd->scaleIdx || !sd->mat || !sd->site || !sd->lnSite || !states || !partials || !freqs || !w)
        Die ("out of memory");

    resource = (beagleResourceNumber >= 0 && beagleResourceNumber != 99) ? beagleResourceNumber : (beagleResourceCount > 0 ? beagleResource[0] : 0);
    for (g=0; g<sd->nClasses; g++)
        {
        cl = &sd->cls[g];
        n = cl->nStates;
        cl->instance = beagleCreateInstance (numLocalTaxa, m->numCondLikes, 0, n, cl->nChars, 1, m->numTiProbs, m->numRateCats,
                                             m->numScalers, &resource, 1, beagleFlags, 0L, &details);
        if (cl->instance < 0)
            return (ERROR);
        if (g == 0)
            MrBayesPrint ("%s   Division %d (standard data): %d transition-matrix classes on %s\n", spacer, d+1, sd->nClasses, details.implName);
        /* tips: the state sets of the compressed matrix (InitChainCondLikes fills the host arrays from the same bits, src/mcmc.c:6303-6330) */
        for (i=0; i<numLocalTaxa; i++)
            {
            ambiguous = NO;
            for (j=0; j<cl->nChars; j++)
                {
                bits = m->parsSets[i] + (size_t) cl->chars[j] * m->nParsIntsPerSite;
                for (s=nSet=0, last=-1; s<n; s++)
                    {
                    partials[(size_t) j * n + s] = IsBitSet (s, bits) ? 1.0 : 0.0;
                    if (IsBitSet (s, bits))
                        {
                        nSet++;
                        last = s;
                        }
                    }
                if (nSet == 1)
                    states[j] = last;
                else if (nSet == n)
                    states[j] = n;                          /* missing / inapplicable: every state compatible */
                else
                    ambiguous = YES;
                }
            rc = (ambiguous == YES) ? beagleSetTipPartials (cl->instance, i, partials) : beagleSetTipStates (cl->instance, i, states);
            if (rc != BEAGLE_SUCCESS)
                return (ERROR);
            }
        for (s=0; s<n; s++)
            freqs[s] = 1.0 / n;                             /* SYMPI_EQUAL: the only frequencies served */
        for (s=0; s<m->numRateCats; s++)
            w[s] = 1.0 / m->numRateCats;
        if (beagleSetStateFrequencies (cl->instance, 0, freqs) != BEAGLE_SUCCESS || beagleSetCategoryWeights (cl->instance, 0, w) != BEAGLE_SUCCESS)
            return (ERROR);
        for (j=0; j<cl->nChars; j++)
            w[j] = 1.0;                                     /* (the weighted sum is taken on the host: dummy characters, coding correction) */
        if (beagleSetPatternWeights (cl->instance, w) != BEAGLE_SUCCESS)
            return (ERROR);
        if (mbamdSetDeferredResult (cl->instance, 1) != BEAGLE_SUCCESS)
            return (ERROR);
        }
    free (states);
    free (partials);
    free (freqs);
    free (w);
    return (NO_ERROR);
}

int MbamdStdServes (ModelInfo *m)
{
    int         d = (int) (m - modelSettings);
    static int  registered = NO;

    if (m->dataType != STANDARD || m->parsModelId == YES || EnvOff () == YES)
        return (NO);
    if (stdDiv == NULL)
        {
        stdDivCount = numCurrentDivisions;
        stdDiv = (StdDivision *) SafeCalloc (stdDivCount, sizeof(StdDivision));
        if (!stdDiv)
            Die ("out of memory");
        }
    if (d < 0 || d >= stdDivCount)
        return (NO);
    if (stdDiv[d].ready != 0)
        return (stdDiv[d].ready > 0 ? YES : NO);
    stdDiv[d].ready = -1;
    if (m->stateFreq->paramId != SYMPI_EQUAL || m->numBetaCats != 1)
        {
        MrBayesPrint ("%s   Division %d (standard data) stays on the host kernels: unequal state frequencies are not served by the engine\n", spacer, d+1);
        return (NO);
        }
    if (m->printSiteRates == YES)
        {
        /* (ancestral states are served: MbamdStdMaterialise below; the site-rate read-out is not) */
        MrBayesPrint ("%s   Division %d (standard data) stays on the host kernels: site rates are read from host arrays\n", spacer, d+1);
        return (NO);
        }
    if ((beagleFlags & BEAGLE_FLAG_PRECISION_DOUBLE) != 0 || tryToUseBEAGLE == NO)
        return (NO);
    if (Setup (m, d) == ERROR)
        {
        MrBayesPrint ("%s   Division %d (standard data) stays on the host kernels: engine set-up failed (%s)\n", spacer, d+1, mbamdGetLastError());
        return (NO);
        }
    if (registered == NO)
        {
        atexit (MbamdStdFinalize);
        registered = YES;
        }
    stdDiv[d].ready = 1;
    return (YES);
}

/* the matrices TiProbs_Std wrote for branch `p` -> every class's instance */
static void SendMatrices (ModelInfo *m, StdDivision *sd, int chain, TreeNode *p)
{
    int         g, i, n, len, idx = m->tiProbsIndex[chain][p->index];
    CLFlt       *tiP = m->tiProbs[idx];

    for (g=0; g<sd->nClasses; g++)
        {
        n = sd->cls[g].nStates;
        len = m->numRateCats * n * n;
        for (i=0; i<len; i++)
            sd->mat[i] = tiP[sd->cls[g].tiIndex + i];
        if (beagleSetTransitionMatrix (sd->cls[g].instance, idx, sd->mat, 0.0) != BEAGLE_SUCCESS)
            Die (mbamdGetLastError());
        }
}

void MbamdStdLogLike (int chain, int d, MrBFlt *lnL)
{
    int             i, j, g, nOps = 0, nScale = 0, rooted, cum, parent, child, prob, zero = 0, whichSitePats;
    double          sum, pUnobserved = 0.0, pObserved, v;
    ModelInfo       *m = &modelSettings[d];
    StdDivision     *sd = &stdDiv[d];
    Tree            *tree = GetTree (m->brlens, chain, state[chain]);
    TreeNode        *p, *top = tree->root->left;
    CLFlt           *nSitesOfPat;
    BeagleOperation *op;

    rooted = tree->isRooted;
    /* the reference's pass over the tree (src/likelihood.c:7882-7965), with its flips and its own TiProbs_Std; the conditional
       likelihoods of a touched node become one operation of the list */
    FlipSiteScalerSpace (m, chain);
    for (i=0; i<tree->nIntNodes; i++)
        {
        p = tree->intDownPass[i];
        if (p->left->upDateTi == YES)
            {
            FlipTiProbsSpace (m, chain, p->left->index);
            m->TiProbs (p->left, d, chain);
            SendMatrices (m, sd, chain, p->left);
            }
        if (p->right->upDateTi == YES)
            {
            FlipTiProbsSpace (m, chain, p->right->index);
            m->TiProbs (p->right, d, chain);
            SendMatrices (m, sd, chain, p->right);
            }
        if (rooted == NO && p->anc->anc == NULL)
            {
            FlipTiProbsSpace (m, chain, p->index);
            m->TiProbs (p, d, chain);
            SendMatrices (m, sd, chain, p);
            }
        if (p->upDateCl == YES)
            {
            FlipCondLikeSpace (m, chain, p->index);
            FlipNodeScalerSpace (m, chain, p->index);
            op = &sd->ops[nOps++];
            op->destinationPartials = m->condLikeIndex[chain][p->index];
            op->destinationScaleWrite = m->nodeScalerIndex[chain][p->index];
            op->destinationScaleRead = BEAGLE_OP_NONE;
            op->child1Partials = m->condLikeIndex[chain][p->left->index];
            op->child1TransitionMatrix = m->tiProbsIndex[chain][p->left->index];
            op->child2Partials = m->condLikeIndex[chain][p->right->index];
            op->child2TransitionMatrix = m->tiProbsIndex[chain][p->right->index];
            }
        sd->scaleIdx[nScale++] = m->nodeScalerIndex[chain][p->index];
        }
    cum = m->siteScalerIndex[chain];
    whichSitePats = chainId[chain] % chainParams.numChains;
    nSitesOfPat = numSitesOfPat + (whichSitePats*numCompressedChars) + m->compCharStart;

    /* every class: partials, the exponents of ALL interior nodes into the cumulative buffer, integration at the top of the tree --
       over the edge to the root tip on an unrooted tree (CondLikeRoot_Std folds that tip in, src/likelihood.c:4496) */
    parent = m->condLikeIndex[chain][top->index];
    child = m->condLikeIndex[chain][tree->root->index];
    prob = m->tiProbsIndex[chain][top->index];
    /* (the instances return from the integration call without waiting -- mbamdSetDeferredResult in Setup --: every class's work is
       on its stream before the first result is waited for) */
    for (g=0; g<sd->nClasses; g++)
        {
        const int inst = sd->cls[g].instance;
        if (nOps > 0 && beagleUpdatePartials (inst, sd->ops, nOps, BEAGLE_OP_NONE) != BEAGLE_SUCCESS)
            Die (mbamdGetLastError());
        if (beagleResetScaleFactors (inst, cum) != BEAGLE_SUCCESS || beagleAccumulateScaleFactors (inst, sd->scaleIdx, nScale, cum) != BEAGLE_SUCCESS)
            Die (mbamdGetLastError());
        if (rooted == NO)
            j = beagleCalculateEdgeLogLikelihoods (inst, &parent, &child, &prob, NULL, NULL, &zero, &zero, &cum, 1, &sum, NULL, NULL);
        else
            j = beagleCalculateRootLogLikelihoods (inst, &parent, &zero, &zero, &cum, 1, &sum);
        if (j != BEAGLE_SUCCESS && j != BEAGLE_ERROR_FLOATING_POINT)
            Die (mbamdGetLastError());
        }
    for (g=0; g<sd->nClasses; g++)
        {
        const int inst = sd->cls[g].instance;
        j = mbamdFetchLogLikelihood (inst, &sum);
        if (j != BEAGLE_SUCCESS && j != BEAGLE_ERROR_FLOATING_POINT)
            Die (mbamdGetLastError());
        if (beagleGetSiteLogLikelihoods (inst, sd->site) != BEAGLE_SUCCESS)
            Die (mbamdGetLastError());
        for (j=0; j<sd->cls[g].nChars; j++)
            sd->lnSite[sd->cls[g].chars[j]] = sd->site[j];
        }
    /* Likelihood_Std (src/likelihood.c:7359-7440) over ln values, characters in the reference's order: the dummy characters'
       probabilities add up to the probability of an unobservable pattern, the others carry the division's likelihood */
    (*lnL) = 0.0;
    for (j=0; j<m->numDummyChars; j++)
        pUnobserved += exp (sd->lnSite[j]);
    for (j=m->numDummyChars; j<m->numChars; j++)
        {
        v = sd->lnSite[j];
        if (!(v > -1.0e300))                                /* (like < LIKE_EPSILON, or not a number) */
            {
            (*lnL) = MRBFLT_NEG_MAX;
            abortMove = YES;
            return;
            }
        (*lnL) += v * nSitesOfPat[j];
        }
    pObserved = 1.0 - pUnobserved;
    if (pObserved < LIKE_EPSILON)
        pObserved = LIKE_EPSILON;
    (*lnL) -= log (pObserved) * (m->numUncompressedChars);
}

/*
 * Ancestral states (report ancstates=yes) for a division on the engine.  The reference's final pass and read-out -- CondLikeUp_Std
 * (src/likelihood.c:4824-4930) and PrintAncStates_Std (src/mcmc.c:11074) -- run UNCHANGED on the host arrays of the division
 * (a standard division is not a BEAGLE division to the reference, so InitChainCondLikes allocated them); this function fills
 * them, once per printed sample, with the down-pass conditional likelihoods the device holds: called in front of the final-pass
 * loop of PrintStates (src/mcmc.c:13138).  CondLikeUp_Std is homogeneous of degree 0 in a node's own down-pass values, per
 * category, so every node but the top one may come with whatever power-of-two scale the device stored; the top node's values
 * start the recursion and are read with all categories of a character at ONE scale (mbamdGetScaledPartials) -- the reference's
 * own arrays have that property because CondLikeScaler_Std scales a character's categories together.  On an unrooted tree the
 * top node of the reference carries the root tip as a third child (CondLikeRoot_Std, src/likelihood.c:4496): that factor is
 * multiplied in here (the device folds it into the edge integration instead).
 * Returns YES when the division is served (the host arrays are now filled), NO otherwise (nothing done).
 */
int MbamdStdMaterialise (int chain, int d)
{
    int             a, c, g, i, j, k, s, n, nK, rc, buf;
    size_t          len;
    double          sum;
    ModelInfo       *m = &modelSettings[d];
    StdDivision     *sd;
    Tree            *tree;
    TreeNode        *p, *top;
    CLFlt           *host, *tiP;
    BitsLong        *bits;

    if (stdDiv == NULL || d < 0 || d >= stdDivCount || stdDiv[d].ready != 1)
        return (NO);
    sd = &stdDiv[d];
    nK = m->numRateCats;
    if (sd->off == NULL)
        {
        sd->off = (int *) SafeCalloc (m->numChars + 1, sizeof(int));
        len = (size_t) nK * (m->numChars + 1) * MAX_STD_STATES;
        sd->part = (double *) SafeCalloc (len, sizeof(double));
        sd->partF = (float *) SafeCalloc (len, sizeof(float));
        sd->lnScaleF = (float *) SafeCalloc (m->numChars + 1, sizeof(float));
        if (!sd->off || !sd->part || !sd->partF || !sd->lnScaleF)
            Die ("out of memory");
        for (c=0, sd->numReps=0; c<m->numChars; c++)
            {
            sd->off[c] = sd->numReps;
            sd->numReps += m->nStates[c];
            }
        }
    tree = GetTree (m->brlens, chain, state[chain]);
    top = tree->root->left;
    for (i=0; i<tree->nIntNodes; i++)
        {
        p = tree->intDownPass[i];
        buf = m->condLikeIndex[chain][p->index];
        host = m->condLikes[buf];
        for (g=0; g<sd->nClasses; g++)
            {
            n = sd->cls[g].nStates;
            if (p == top)
                rc = mbamdGetScaledPartials (sd->cls[g].instance, buf, m->siteScalerIndex[chain], sd->partF, sd->lnScaleF);
            else
                rc = beagleGetPartials (sd->cls[g].instance, buf, BEAGLE_OP_NONE, sd->part);
            if (rc != BEAGLE_SUCCESS)
                Die (mbamdGetLastError());
            for (k=0; k<nK; k++)
                for (j=0; j<sd->cls[g].nChars; j++)
                    for (s=0; s<n; s++)
                        {
                        len = ((size_t) k * sd->cls[g].nChars + j) * n + s;
                        host[(size_t) k * sd->numReps + sd->off[sd->cls[g].chars[j]] + s] = (p == top) ? (CLFlt) sd->partF[len] : (CLFlt) sd->part[len];
                        }
            }
        }
    if (tree->isRooted == NO)
        {
        host = m->condLikes[m->condLikeIndex[chain][top->index]];
        for (k=0; k<nK; k++)
            for (c=0; c<m->numChars; c++)
                {
                n = m->nStates[c];
                tiP = m->tiProbs[m->tiProbsIndex[chain][top->index]] + m->tiIndex[c] + k * n * n;
                bits = m->parsSets[tree->root->index] + (size_t) c * m->nParsIntsPerSite;
                for (a=0; a<n; a++)
                    {
                    for (j=0, sum=0.0; j<n; j++)
                        if (IsBitSet (j, bits))
                            sum += tiP[a*n + j];
                    host[(size_t) k * sd->numReps + sd->off[c] + a] *= (CLFlt) sum;
                    }
                }
        }
    return (YES);
}
