/*
 * mbamd_std_glue.c -- standard (morphology) data on the engine: the MrBayes side (see mbamd_std_glue.h).
 * Compiled and linked with the reference's own sources; our code, no reference source in it.
 *
 * What the reference does per evaluation for such a division (LaunchLogLikeForDivision, src/likelihood.c:7851-7972: flip the
 * index tables, TiProbs_Std per touched branch, CondLikeDown_Std / CondLikeRoot_Std per touched node, CondLikeScaler_Std,
 * Likelihood_Std) is kept as a FLOW -- the same index tables (condLikeIndex, tiProbsIndex, nodeScalerIndex, siteScalerIndex)
 * name the engine's buffers, so accept / reject (ResetFlips, src/mcmc.c:15695) needs no knowledge of the binding -- while the
 * arithmetic runs on the device, one engine instance per transition-matrix class of the division: conditional likelihoods,
 * rescaling, integration AND the transition matrices.  TiProbs_Std (src/likelihood.c:10066-10462) has a closed form per class; each
 * is the matrix exponential of a rate matrix whose eigen-system is known in closed form too (ClassEigen below), so a class hands the
 * engine that eigen-system once and from then on only the touched branches' lengths (beagleUpdateTransitionMatrices: one call per
 * class and evaluation -- no host matrices, no upload per branch).  Binary characters under a symmetric-beta prior on their state
 * frequencies (numBetaCats > 1, src/likelihood.c:1942-1955, 7401-7470) are a mixture over beta categories with their own
 * frequencies: the class's instance holds one set of buffers per beta category ("parts", as the reference's BEAGLE path does for the
 * omega classes of a codon model, src/mbbeagle.c:1040-1104) and the integration mixes them.
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

#define MBAMD_STD_MAXBETA 8     /* = the engine's MBAMD_MAX_SUBSETS (csrc/mbamd_kernels.h): one integration mixes at most eight buffer sets */

typedef struct
    {
    int     instance;       /* engine instance of this class */
    int     nStates;        /* states of its characters */
    int     cType;          /* UNORD or ORD */
    int     nParts;         /* beta categories of a binary class under unequal frequencies (else 1): buffer sets of the instance */
    int     nChars;         /* compressed characters of the class ... */
    int     *chars;         /* ... and which (division-local indices, ascending) */
    double  rates[64];      /* category rates / beta-category frequencies the instance holds (sent again only when they change) */
    double  freqs[2 * MBAMD_STD_MAXBETA];
    int     haveRates;
    } StdClass;

typedef struct
    {
    int         ready;      /* 0: not looked at, 1: served, -1: refused */
    int         nClasses;
    StdClass    cls[MBAMD_STD_MAXCLASSES];
    BeagleOperation *ops, *opsPart;
    int         *scaleIdx, *scalePart, *matIdx, *matPart;
    double      *matLen;
    double      *site, *lnSite;
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

/* everything the binding holds, gone: the instances, the class tables, the scratch arrays.  Called when the reference frees its own
   chain memory (FreeChainMemory, src/mcmc.c:4403 -- a later analysis of the session may have other chains, taxa, characters,
   categories or priors: MbamdStdServes looks at the model again) and at exit. */
void MbamdStdFinalize (void)
{
    int d, g;
    for (d=0; d<stdDivCount && stdDiv != NULL; d++)
        {
        StdDivision *sd = &stdDiv[d];
        for (g=0; g<sd->nClasses; g++)
            {
            if (sd->cls[g].instance >= 0)
                beagleFinalizeInstance (sd->cls[g].instance);
            sd->cls[g].instance = -1;
            free (sd->cls[g].chars);
            sd->cls[g].chars = NULL;
            }
        free (sd->ops); free (sd->opsPart); free (sd->scaleIdx); free (sd->scalePart); free (sd->matIdx); free (sd->matPart); free (sd->matLen);
        free (sd->site); free (sd->lnSite); free (sd->off); free (sd->part); free (sd->partF); free (sd->lnScaleF);
        }
    free (stdDiv);
    stdDiv = NULL;
    stdDivCount = 0;
}

/* The eigen-system of a class's rate matrix under equal state frequencies (row-major U, its inverse, the eigenvalues):
 *   unordered, n states: every change at the same rate, one expected change per unit time: Q = (J - n I) / (n - 1), eigenvalues 0 and
 *     -n / (n - 1) (n - 1 times), diagonalised by ANY orthonormal basis that contains the constant vector;
 *   ordered, n states: changes between neighbours only: Q = mu (A - D) on the path 0 - 1 - ... - (n - 1), mu = n / (2 (n - 1)) for one
 *     expected change per unit time; the path Laplacian's eigenvectors are the cosines u_s(i) = cos ((2 i + 1) s pi / (2 n)), its
 *     eigenvalues -2 mu (1 - cos (s pi / n)).
 * The cosine basis serves both.  exp (Q v) is what TiProbs_Std spells out term by term for 2 ... 10 unordered and 3 ... 6 ordered states
 * (src/likelihood.c:10141-10400; e.g. four ordered states: exponents -4/3, 2 (sqrt 2 - 2) / 3, -2 (sqrt 2 + 2) / 3 = the three values here). */
static void ClassEigen (int n, int ordered, double *U, double *Uinv, double *lam)
{
    int     i, s;
    double  pi = 3.14159265358979323846, mu = n / (2.0 * (n - 1.0)), norm;

    for (s=0; s<n; s++)
        {
        norm = (s == 0) ? sqrt (1.0 / n) : sqrt (2.0 / n);
        for (i=0; i<n; i++)
            {
            U[i*n + s] = norm * cos ((2.0 * i + 1.0) * s * pi / (2.0 * n));
            Uinv[s*n + i] = U[i*n + s];
            }
        if (s == 0)
            lam[s] = 0.0;
        else
            lam[s] = (ordered == YES) ? -2.0 * mu * (1.0 - cos (s * pi / n)) : -n / (n - 1.0);
        }
}

/* binary character with stationary frequencies (pi0, pi1), one expected change per unit time (src/likelihood.c:10409-10424) */
static void BinaryEigen (double pi0, double pi1, double *U, double *Uinv, double *lam)
{
    U[0] = 1.0;  U[1] = pi1;
    U[2] = 1.0;  U[3] = -pi0;
    Uinv[0] = pi0;  Uinv[1] = pi1;
    Uinv[2] = 1.0;  Uinv[3] = -1.0;
    lam[0] = 0.0;
    lam[1] = -1.0 / (2.0 * pi0 * pi1);
}

/* buffer of part b that belongs to the reference's index i: tips are shared by the parts, everything else comes in sets */
static int PartBuffer (int i, int b, int nParts)
{
    return (i < numLocalTaxa) ? i : numLocalTaxa + (i - numLocalTaxa) * nParts + b;
}

/* the classes of a division, their instances, tip data, eigen-systems, frequencies and weights: once per analysis */
static int Setup (ModelInfo *m, int d, int unequal)
{
    int             b, c, g, i, j, s, n, nSet, last, ambiguous, rc, *states, resource, nB;
    double          *partials, *freqs, *w, U[MAX_STD_STATES*MAX_STD_STATES], Uinv[MAX_STD_STATES*MAX_STD_STATES], lam[MAX_STD_STATES];
    BitsLong        *bits;
    StdDivision     *sd = &stdDiv[d];
    StdClass        *cl;
    BeagleInstanceDetails details;
    Tree            *t = GetTree (m->brlens, 0, 0);

    sd->nClasses = 0;
    for (c=0; c<m->numChars; c++)
        {
        if (m->cType[c] != UNORD && m->cType[c] != ORD)
            return (ERROR);                                 /* (irreversible characters: TiProbs_Std has no matrices for them either) */
        for (g=0; g<sd->nClasses; g++)
            if (sd->cls[g].nStates == m->nStates[c] && (sd->cls[g].cType == m->cType[c] || m->nStates[c] == 2))
                break;
        if (g == sd->nClasses)
            {
            if (g == MBAMD_STD_MAXCLASSES)
                return (ERROR);
            sd->cls[g].nStates = m->nStates[c];
            sd->cls[g].cType = (m->nStates[c] == 2) ? UNORD : m->cType[c];
            sd->cls[g].nParts = (unequal == YES && m->nStates[c] == 2) ? m->numBetaCats : 1;
            sd->cls[g].nChars = 0;
            sd->cls[g].haveRates = NO;
            sd->cls[g].chars = (int *) SafeCalloc (m->numChars, sizeof(int));
            sd->cls[g].instance = -1;
            sd->nClasses++;
            }
        sd->cls[g].chars[sd->cls[g].nChars++] = c;
        }
    nB = (unequal == YES) ? m->numBetaCats : 1;
    if (nB > MBAMD_STD_MAXBETA || m->numRateCats > 64)
        return (ERROR);
    n = (t->nIntNodes + 1) * nB;
    sd->ops = (BeagleOperation *) SafeCalloc (t->nIntNodes + 1, sizeof(BeagleOperation));
    sd->opsPart = (BeagleOperation *) SafeCalloc (n, sizeof(BeagleOperation));
    sd->scaleIdx = (int *) SafeCalloc (t->nIntNodes + 1, sizeof(int));
    sd->scalePart = (int *) SafeCalloc (n, sizeof(int));
    sd->matIdx = (int *) SafeCalloc (2 * t->nNodes + 2, sizeof(int));
    sd->matPart = (int *) SafeCalloc (2 * t->nNodes + 2, sizeof(int));
    sd->matLen = (double *) SafeCalloc (2 * t->nNodes + 2, sizeof(double));
    sd->site = (double *) SafeCalloc (m->numChars + 1, sizeof(double));
    sd->lnSite = (double *) SafeCalloc (m->numChars + 1, sizeof(double));
    states = (int *) SafeCalloc (m->numChars + 1, sizeof(int));
    partials = (double *) SafeCalloc ((size_t) (m->numChars + 1) * MAX_STD_STATES, sizeof(double));
    freqs = (double *) SafeCalloc (MAX_STD_STATES, sizeof(double));
    w = (double *) SafeCalloc (m->numRateCats + m->numChars + 1, sizeof(double));
    if (!sd->ops || !sd->opsPart || !sd->scaleIdx || !sd->scalePart || !sd->matIdx || !sd->matPart || !sd->matLen || !sd->site || !sd->lnSite ||
        !states || !partials || !freqs || !w)
        Die ("out of memory");

    resource = (beagleResourceNumber >= 0 && beagleResourceNumber != 99) ? beagleResourceNumber : (beagleResourceCount > 0 ? beagleResource[0] : 0);
    rc = NO_ERROR;
    for (g=0; g<sd->nClasses && rc == NO_ERROR; g++)
        {
        cl = &sd->cls[g];
        n = cl->nStates;
        nB = cl->nParts;
        cl->instance = beagleCreateInstance (numLocalTaxa, numLocalTaxa + (m->numCondLikes - numLocalTaxa) * nB, 0, n, cl->nChars, nB, m->numTiProbs * nB,
                                             m->numRateCats, m->numScalers * nB, &resource, 1, beagleFlags, 0L, &details);
        if (cl->instance < 0)
            {
            rc = ERROR;
            break;
            }
        if (g == 0)
            MrBayesPrint ("%s   Division %d (standard data): %d transition-matrix classes on %s\n", spacer, d+1, sd->nClasses, details.implName);
        /* tips: the state sets of the compressed matrix (InitChainCondLikes fills the host arrays from the same bits, src/mcmc.c:6303-6330) */
        for (i=0; i<numLocalTaxa && rc == NO_ERROR; i++)
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
            if (((ambiguous == YES) ? beagleSetTipPartials (cl->instance, i, partials) : beagleSetTipStates (cl->instance, i, states)) != BEAGLE_SUCCESS)
                rc = ERROR;
            }
        for (b=0; b<nB && rc == NO_ERROR; b++)
            {
            /* equal frequencies: the class's eigen-system, for good; a binary class under unequal frequencies: per beta category, with
               whatever the frequencies are at every evaluation (MbamdStdLogLike) */
            for (s=0; s<n; s++)
                freqs[s] = 1.0 / n;
            for (s=0; s<m->numRateCats; s++)
                w[s] = 1.0 / (m->numRateCats * nB);         /* (Likelihood_Std: catFreq = rateFreq / nBetaCats, src/likelihood.c:7459-7462) */
            if (nB == 1)
                {
                ClassEigen (n, cl->cType == ORD ? YES : NO, U, Uinv, lam);
                if (beagleSetEigenDecomposition (cl->instance, 0, U, Uinv, lam) != BEAGLE_SUCCESS || beagleSetStateFrequencies (cl->instance, 0, freqs) != BEAGLE_SUCCESS)
                    rc = ERROR;
                }
            else
                cl->freqs[2*b] = cl->freqs[2*b+1] = -1.0;
            if (rc == NO_ERROR && beagleSetCategoryWeights (cl->instance, b, w) != BEAGLE_SUCCESS)
                rc = ERROR;
            }
        for (j=0; j<cl->nChars; j++)
            w[j] = 1.0;                                     /* (the weighted sum is taken on the host: dummy characters, coding correction) */
        if (rc == NO_ERROR && (beagleSetPatternWeights (cl->instance, w) != BEAGLE_SUCCESS || mbamdSetDeferredResult (cl->instance, 1) != BEAGLE_SUCCESS))
            rc = ERROR;
        }
    free (states);
    free (partials);
    free (freqs);
    free (w);
    return (rc);
}

int MbamdStdServes (ModelInfo *m)
{
    int         c, d = (int) (m - modelSettings), unequal;
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
    unequal = (m->stateFreq->paramId != SYMPI_EQUAL) ? YES : NO;
    if (unequal == YES)
        {
        /* unequal state frequencies: binary characters are a mixture over the beta categories (served); a character with more
           states has an eigen-system of its own (src/likelihood.c:10426-10455) -- a class per character: not served */
        for (c=0; c<m->numChars; c++)
            if (m->nStates[c] != 2)
                {
                MrBayesPrint ("%s   Division %d (standard data) stays on the host kernels: unequal state frequencies of characters with more than two states are not served by the engine\n", spacer, d+1);
                return (NO);
                }
        if (m->printAncStates == YES)
            {
            MrBayesPrint ("%s   Division %d (standard data) stays on the host kernels: ancestral states under unequal state frequencies are read from host arrays\n", spacer, d+1);
            return (NO);
            }
        if (m->numBetaCats > MBAMD_STD_MAXBETA)
            {
            /* (`lset nbetacat` goes up to MAX_RATE_CATS - 1; the engine's integration mixes at most eight buffer sets: saying YES here
               would end in Die() at the first evaluation instead of on the host kernels) */
            MrBayesPrint ("%s   Division %d (standard data) stays on the host kernels: %d beta categories, the engine mixes at most %d\n", spacer, d+1, m->numBetaCats, MBAMD_STD_MAXBETA);
            return (NO);
            }
        }
    else if (m->numBetaCats != 1)
        return (NO);
    if (m->printSiteRates == YES)
        {
        /* (ancestral states are served: MbamdStdMaterialise below; the site-rate read-out is not) */
        MrBayesPrint ("%s   Division %d (standard data) stays on the host kernels: site rates are read from host arrays\n", spacer, d+1);
        return (NO);
        }
    if ((beagleFlags & BEAGLE_FLAG_PRECISION_DOUBLE) != 0 || tryToUseBEAGLE == NO)
        return (NO);
    if (Setup (m, d, unequal) == ERROR)
        {
        int g;
        MrBayesPrint ("%s   Division %d (standard data) stays on the host kernels: engine set-up failed (%s)\n", spacer, d+1, mbamdGetLastError());
        for (g=0; g<stdDiv[d].nClasses; g++)             /* (the instances created before the failure) */
            if (stdDiv[d].cls[g].instance >= 0)
                {
                beagleFinalizeInstance (stdDiv[d].cls[g].instance);
                stdDiv[d].cls[g].instance = -1;
                }
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

/* the length TiProbs_Std uses for the branch above `p` (src/likelihood.c:10095-10133): a relaxed-clock model's effective length, clamped */
static double BranchLength (ModelInfo *m, TreeNode *p, int chain)
{
    double length;

    if (m->cppEvents != NULL)
        length = GetParamSubVals (m->cppEvents, chain, state[chain])[p->index];
    else if (m->tk02BranchRates != NULL)
        length = GetParamSubVals (m->tk02BranchRates, chain, state[chain])[p->index];
    else if (m->wnBranchRates != NULL)
        length = GetParamSubVals (m->wnBranchRates, chain, state[chain])[p->index];
    else if (m->ilnBranchRates != NULL)
        length = GetParamSubVals (m->ilnBranchRates, chain, state[chain])[p->index];
    else if (m->igrBranchRates != NULL)
        length = GetParamSubVals (m->igrBranchRates, chain, state[chain])[p->index];
    else if (m->mixedBrchRates != NULL)
        length = GetParamSubVals (m->mixedBrchRates, chain, state[chain])[p->index];
    else
        length = p->length;
    if (length > BRLENS_MAX)
        length = BRLENS_MAX;
    else if (length < BRLENS_MIN)
        length = BRLENS_MIN;
    return (length);
}

void MbamdStdLogLike (int chain, int d, MrBFlt *lnL)
{
    int             i, j, k, g, b, nB, nOps = 0, nScale = 0, nMat = 0, rooted, whichSitePats, changed;
    int             parent[MBAMD_STD_MAXBETA], child[MBAMD_STD_MAXBETA], prob[MBAMD_STD_MAXBETA], part[MBAMD_STD_MAXBETA], cum[MBAMD_STD_MAXBETA];
    double          sum, pUnobserved = 0.0, pObserved, v, baseRate, theRate = 1.0, *catRate, rates[64], *bs, U[4], Uinv[4], lam[2];
    ModelInfo       *m = &modelSettings[d];
    StdDivision     *sd = &stdDiv[d];
    Tree            *tree = GetTree (m->brlens, chain, state[chain]);
    TreeNode        *p, *top = tree->root->left;
    CLFlt           *nSitesOfPat;
    BeagleOperation *op;
    StdClass        *cl;

    rooted = tree->isRooted;
    /* the reference's pass over the tree (src/likelihood.c:7882-7965) with its flips; a touched branch becomes an entry of the matrix
       list (index, length), the conditional likelihoods of a touched node one operation of the list */
    FlipSiteScalerSpace (m, chain);
    for (i=0; i<tree->nIntNodes; i++)
        {
        p = tree->intDownPass[i];
        if (p->left->upDateTi == YES)
            {
            FlipTiProbsSpace (m, chain, p->left->index);
            sd->matIdx[nMat] = m->tiProbsIndex[chain][p->left->index];
            sd->matLen[nMat++] = BranchLength (m, p->left, chain);
            }
        if (p->right->upDateTi == YES)
            {
            FlipTiProbsSpace (m, chain, p->right->index);
            sd->matIdx[nMat] = m->tiProbsIndex[chain][p->right->index];
            sd->matLen[nMat++] = BranchLength (m, p->right, chain);
            }
        if (rooted == NO && p->anc->anc == NULL)
            {
            FlipTiProbsSpace (m, chain, p->index);
            sd->matIdx[nMat] = m->tiProbsIndex[chain][p->index];
            sd->matLen[nMat++] = BranchLength (m, p, chain);
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
    whichSitePats = chainId[chain] % chainParams.numChains;
    nSitesOfPat = numSitesOfPat + (whichSitePats*numCompressedChars) + m->compCharStart;

    /* the rates TiProbs_Std multiplies a branch length with (src/likelihood.c:10083-10093): base rate x category rate */
    baseRate = GetRate (d, chain);
    if (m->shape != NULL)
        catRate = GetParamSubVals (m->shape, chain, state[chain]);
    else if (m->mixtureRates != NULL)
        catRate = GetParamSubVals (m->mixtureRates, chain, state[chain]);
    else
        catRate = &theRate;
    for (k=0; k<m->numRateCats; k++)
        rates[k] = baseRate * catRate[k];
    bs = GetParamStdStateFreqs (m->stateFreq, chain, state[chain]);

    /* every class: rates and (beta categories) eigen-systems if they changed, the touched branches' matrices, partials, the exponents
       of ALL interior nodes into the cumulative buffer, integration at the top of the tree -- over the edge to the root tip on an
       unrooted tree (CondLikeRoot_Std folds that tip in, src/likelihood.c:4496).  The instances return from the integration call
       without waiting (mbamdSetDeferredResult in Setup): every class's work is on its stream before the first result is waited for. */
    for (g=0; g<sd->nClasses; g++)
        {
        cl = &sd->cls[g];
        nB = cl->nParts;
        changed = (cl->haveRates == NO) ? YES : NO;
        for (k=0; k<m->numRateCats && changed == NO; k++)
            if (cl->rates[k] != rates[k])
                changed = YES;
        if (changed == YES)
            {
            if (beagleSetCategoryRates (cl->instance, rates) != BEAGLE_SUCCESS)
                Die (mbamdGetLastError());
            for (k=0; k<m->numRateCats; k++)
                cl->rates[k] = rates[k];
            cl->haveRates = YES;
            }
        for (b=0; b<nB; b++)
            {
            if (nB > 1 && (cl->freqs[2*b] != bs[2*b] || cl->freqs[2*b+1] != bs[2*b+1]))
                {
                /* (the frequencies of beta category b moved -- or moved back after a rejected proposal: the matrices of the touched
                    branches are made below from this eigen-system, the others sit in the buffers the index flips restored) */
                BinaryEigen (bs[2*b], bs[2*b+1], U, Uinv, lam);
                if (beagleSetEigenDecomposition (cl->instance, b, U, Uinv, lam) != BEAGLE_SUCCESS || beagleSetStateFrequencies (cl->instance, b, bs + 2*b) != BEAGLE_SUCCESS)
                    Die (mbamdGetLastError());
                cl->freqs[2*b] = bs[2*b];
                cl->freqs[2*b+1] = bs[2*b+1];
                }
            if (nMat > 0)
                {
                for (i=0; i<nMat; i++)
                    sd->matPart[i] = sd->matIdx[i] * nB + b;
                if (beagleUpdateTransitionMatrices (cl->instance, b, sd->matPart, NULL, NULL, sd->matLen, nMat) != BEAGLE_SUCCESS)
                    Die (mbamdGetLastError());
                }
            }
        for (b=0; b<nB; b++)
            {
            for (i=0; i<nOps; i++)
                {
                op = &sd->opsPart[i];
                op->destinationPartials = PartBuffer (sd->ops[i].destinationPartials, b, nB);
                op->destinationScaleWrite = sd->ops[i].destinationScaleWrite * nB + b;
                op->destinationScaleRead = BEAGLE_OP_NONE;
                op->child1Partials = PartBuffer (sd->ops[i].child1Partials, b, nB);
                op->child1TransitionMatrix = sd->ops[i].child1TransitionMatrix * nB + b;
                op->child2Partials = PartBuffer (sd->ops[i].child2Partials, b, nB);
                op->child2TransitionMatrix = sd->ops[i].child2TransitionMatrix * nB + b;
                }
            for (i=0; i<nScale; i++)
                sd->scalePart[i] = sd->scaleIdx[i] * nB + b;
            cum[b] = m->siteScalerIndex[chain] * nB + b;
            parent[b] = PartBuffer (m->condLikeIndex[chain][top->index], b, nB);
            child[b] = PartBuffer (m->condLikeIndex[chain][tree->root->index], b, nB);
            prob[b] = m->tiProbsIndex[chain][top->index] * nB + b;
            part[b] = b;
            if (nOps > 0 && beagleUpdatePartials (cl->instance, sd->opsPart, nOps, BEAGLE_OP_NONE) != BEAGLE_SUCCESS)
                Die (mbamdGetLastError());
            if (beagleResetScaleFactors (cl->instance, cum[b]) != BEAGLE_SUCCESS || beagleAccumulateScaleFactors (cl->instance, sd->scalePart, nScale, cum[b]) != BEAGLE_SUCCESS)
                Die (mbamdGetLastError());
            }
        if (rooted == NO)
            j = beagleCalculateEdgeLogLikelihoods (cl->instance, parent, child, prob, NULL, NULL, part, part, cum, nB, &sum, NULL, NULL);
        else
            j = beagleCalculateRootLogLikelihoods (cl->instance, parent, part, part, cum, nB, &sum);
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
    /* the matrices CondLikeUp_Std multiplies with live on the device since round 5: the reference's own TiProbs_Std fills the host
       arrays of every branch here, once per printed sample (a few dozen numbers per branch) */
    for (i=0; i<tree->nNodes; i++)
        {
        p = tree->allDownPass[i];
        if (p->anc != NULL && (p->anc->anc != NULL || tree->isRooted == NO))
            m->TiProbs (p, d, chain);
        }
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
