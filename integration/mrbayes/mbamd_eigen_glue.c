/*
 * mbamd_eigen_glue.c -- the MrBayes side of the device eigen-solver (see mbamd_eigen_glue.h).
 * Compiled and linked with the reference's own sources; our code, no reference source in it.
 */
#include "bayes.h"
#include "mcmc.h"
#include "model.h"
#include "utils.h"
#include "libhmsbeagle/beagle.h"
#include "mbamd_eigen_glue.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if !defined (BEAGLE_ENABLED)
#error "the eigen binding needs the BEAGLE build (m->useBeagle, m->beagleInstance)"
#endif

extern int      numLocalChains;         /* (src/mcmc.c; declared like this in src/likelihood.c:53) */

static double   *flatQ = NULL;
static size_t   flatCap = 0;
static long     nDevice = 0;

static void Report (void)
{
    if (getenv("MBAMD_STATS") != NULL)
        fprintf (stderr, "mbamd eigen: %ld rate-matrix sets decomposed on the device\n", nDevice);
}

static int Active (ModelInfo *m, int n)
{
    const char *s = getenv("MBAMD_DEVICE_EIGEN");
    if (s != NULL && s[0] == '0')
        return (NO);
    if (m->useBeagle == NO || m->dataType == STANDARD || n > 64 || n < 2)
        return (NO);
    if ((beagleFlags & BEAGLE_FLAG_PRECISION_DOUBLE) != 0)
        return (NO);                /* (the double-precision engine takes finished eigen-systems only) */
    return (YES);
}

int MbamdGetEigens (ModelInfo *m, int chain, MrBFlt ***allQ, int part, int n, MrBFlt **q, MrBFlt *eigenValues, MrBFlt *eigvalsImag,
                    MrBFlt **eigvecs, MrBFlt **inverseEigvecs, MrBComplex **Ceigvecs, MrBComplex **CinverseEigvecs)
{
    int         i, j, k, parts, divisionOffset, rc;
    size_t      need;
    MrBFlt      *bs, covBF[64], *swr, probOn;
    double      pi[64];

    if (Active (m, n) == NO)
        return GetEigens (n, q, eigenValues, eigvalsImag, eigvecs, inverseEigvecs, Ceigvecs, CinverseEigvecs);

    /* what UpDateCijk copies out of these and sends is ignored by the engine (shield): well-defined values all the same */
    for (i=0; i<n; i++)
        {
        eigenValues[i] = 0.0;
        eigvalsImag[i] = 0.0;
        for (j=0; j<n; j++)
            eigvecs[i][j] = inverseEigvecs[i][j] = (i == j) ? 1.0 : 0.0;
        }
    if (part > 0)
        return (NO);                /* sent with part 0 */

    parts = (m->nCijkParts > 1) ? m->nCijkParts : 1;
    need = (size_t) parts * n * n;
    if (need > flatCap)
        {
        flatCap = need;
        flatQ = (double *) realloc (flatQ, flatCap * sizeof(double));
        if (!flatQ)
            {
            fprintf (stderr, "mbamd eigen: out of memory\n");
            exit (1);
            }
        }
    for (k=0; k<parts; k++)
        for (i=0; i<n; i++)
            for (j=0; j<n; j++)
                flatQ[((size_t) k * n + i) * n + j] = allQ[k][i][j];

    /* the stationary distribution the rate matrices are reversible under (covarion: on / off copies, as src/mbbeagle.c:1154-1172) */
    bs = GetParamSubVals (m->stateFreq, chain, state[chain]);
    if (m->switchRates != NULL)
        {
        swr = GetParamVals (m->switchRates, chain, state[chain]);
        probOn = swr[0] / (swr[0] + swr[1]);
        for (i=0; i<n/2; i++)
            {
            covBF[i] = bs[i] * probOn;
            covBF[i+n/2] = bs[i] * (1.0 - probOn);
            }
        bs = covBF;
        }
    for (i=0; i<n; i++)
        pi[i] = bs[i];

    divisionOffset = 0;
    if (m->useBeagleMultiPartitions == YES)
        divisionOffset = (numLocalChains + 1) * m->nCijkParts * m->divisionIndex;
    /* FlipCijkSpace has run (src/likelihood.c:10497): cijkIndex[chain] is the buffer being written, cijkScratchIndex the
       chain's current state -- the warm start */
    rc = mbamdSetRateMatricesFrom (m->beagleInstance, m->cijkIndex[chain] + divisionOffset, parts, flatQ, pi, 2,
                                   m->cijkScratchIndex + divisionOffset);
    if (rc != BEAGLE_SUCCESS)
        {
        fprintf (stderr, "mbamd eigen: mbamdSetRateMatricesFrom failed (%s)\n", mbamdGetLastError());
        exit (1);
        }
    if (nDevice++ == 0)
        atexit (Report);
    return (NO);
}
