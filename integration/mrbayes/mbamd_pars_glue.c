/*
 * mbamd_pars_glue.c -- the MrBayes side of the device parsimony scorer (see mbamd_pars_glue.h).
 * Compiled and linked with the reference's own sources; our code, no reference source in it.
 *
 * The device keeps a mirror of m->parsSets for every division (created on first use, all sets uploaded once: tips from
 * InitParsSets, interior sets as the start-tree builder left them).  From then on the patched moves update the mirror
 * only; set MBAMD_PARS_CHECK=1 to run the reference's host functions next to every device call and compare the state
 * sets word for word and the candidate lengths exactly (the parity pin of the scorer against the reference itself).
 */
#include "bayes.h"
#include "mcmc.h"
#include "utils.h"
#include "libhmsbeagle/mbamd_parsimony.h"
#include "mbamd_pars_glue.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if !defined (BEAGLE_ENABLED)
#error "the device parsimony binding needs the BEAGLE build (m->useBeagle, m->beagleInstance)"
#endif

extern int *chainId;

static int      *parsHandle = NULL;        /* per division: device instance or -1 */
/* per division: what the mirror was filled from.  The reference frees and re-allocates m->parsSets between analyses
   (FreeChainMemory / InitParsSets, src/mcmc.c:4437, 6870+) and malloc readily hands out the same address again, so the
   pointer alone does not identify the sets: the shape, the likelihood instance of the run (every analysis creates a new
   one) and a checksum of the tip rows -- which no move ever writes -- are kept next to it. */
typedef struct
    {
    BitsLong    **sets, *row0;
    int         numParsSets, numChars, nInts, beagleInstance, compCharStart, numTips;
    BitsLong    tipSum;
    }
ParsOwner;
static ParsOwner *parsOwner = NULL;
static int      nHandles = 0;
static int      *opBuf = NULL;
static int      opCap = 0;
static int      envRead = NO, envOff = NO, envCheck = NO;
static MrBFlt   *keptLengths = NULL;       /* check mode: the device's candidate lengths */
static int      keptN = 0;
static long     nCompared = 0;

static void ReadEnv (void)
{
    const char *s;
    if (envRead == YES)
        return;
    envRead = YES;
    s = getenv("MBAMD_DEVICE_PARSIMONY");
    if (s != NULL && s[0] == '0')
        envOff = YES;
    s = getenv("MBAMD_PARS_CHECK");
    if (s != NULL && s[0] != '0')
        envCheck = YES;
}

static void Die (const char *what)
{
    fprintf (stderr, "mbamd parsimony: %s (%s)\n", what, mbamdGetLastError());
    exit (1);
}

static void ReportCheck (void)
{
    if (envCheck == YES)
        fprintf (stderr, "mbamd parsimony check: %ld comparisons against the host functions, all equal\n", nCompared);
}

static int *Ops (int n)
{
    if (4 * n > opCap)
        {
        opCap = 8 * n + 64;
        opBuf = (int *) realloc (opBuf, (size_t) opCap * sizeof(int));
        if (!opBuf)
            Die ("out of memory");
        }
    return opBuf;
}

/* a cheap checksum over the tip rows of m->parsSets (rows 0 .. numLocalTaxa-1: InitParsSets fills them, no move writes them):
   about 4096 words spread over all tips, so that the test costs a few microseconds per call whatever the matrix size
   (Handle() runs several times per proposal) */
static BitsLong TipSum (ModelInfo *m, int numTips)
{
    int         i;
    size_t      c, n = (size_t) m->numChars * (size_t) m->nParsIntsPerSite, step;
    BitsLong    h = 1469598103934665603UL;

    step = (n * (size_t) (numTips > 0 ? numTips : 1)) / 4096 + 1;
    if (step % 2 == 0)
        step++;
    for (i=0; i<numTips; i++)
        for (c=(size_t) i % step; c<n; c+=step)
            h = (h ^ m->parsSets[i][c]) * 1099511628211UL;
    return h;
}

static void FillOwner (ParsOwner *o, ModelInfo *m)
{
    o->sets = m->parsSets;
    o->row0 = m->parsSets[0];
    o->numParsSets = m->numParsSets;
    o->numChars = m->numChars;
    o->nInts = m->nParsIntsPerSite;
    o->beagleInstance = m->beagleInstance;
    o->compCharStart = m->compCharStart;
    o->numTips = numLocalTaxa < m->numParsSets ? numLocalTaxa : m->numParsSets;
    o->tipSum = TipSum (m, o->numTips);
}

static int SameOwner (ParsOwner *o, ModelInfo *m)
{
    if (o->sets != m->parsSets || o->row0 != m->parsSets[0] || o->numParsSets != m->numParsSets || o->numChars != m->numChars ||
        o->nInts != m->nParsIntsPerSite || o->beagleInstance != m->beagleInstance || o->compCharStart != m->compCharStart ||
        o->numTips != (numLocalTaxa < m->numParsSets ? numLocalTaxa : m->numParsSets))
        return (NO);
    return (o->tipSum == TipSum (m, o->numTips) ? YES : NO);
}

/* the device instance of a division, created (and filled with the host's sets) on first use */
static int Handle (int division)
{
    int         i, c, bits, id;
    BitsLong    any;
    ModelInfo   *m = &modelSettings[division];

    if (division >= nHandles)
        {
        parsHandle = (int *) realloc (parsHandle, (size_t) (division + 1) * sizeof(int));
        parsOwner = (ParsOwner *) realloc (parsOwner, (size_t) (division + 1) * sizeof(ParsOwner));
        if (!parsHandle || !parsOwner)
            Die ("out of memory");
        for (i=nHandles; i<=division; i++)
            {
            parsHandle[i] = -1;
            memset (&parsOwner[i], 0, sizeof(ParsOwner));
            }
        if (nHandles == 0)
            atexit (ReportCheck);
        nHandles = division + 1;
        }
    if (parsHandle[division] >= 0 && SameOwner (&parsOwner[division], m) == YES)
        return parsHandle[division];
    if (parsHandle[division] >= 0)
        {
        /* the host's sets were re-allocated (another analysis in the same session): the mirror is of no use any more */
        mbamdParsFinalizeInstance (parsHandle[division]);
        parsHandle[division] = -1;
        }

    bits = 64 * m->nParsIntsPerSite;
    if (m->nParsIntsPerSite == 1)
        {
        any = 0;
        for (i=0; i<m->numParsSets; i++)
            for (c=0; c<m->numChars; c++)
                any |= m->parsSets[i][c];
        for (bits=1; bits<64 && (any >> bits) != 0; bits++)
            ;
        }
    id = mbamdParsCreateInstance (m->numParsSets, m->numChars, m->nParsIntsPerSite, bits, m->beagleInstance);
    if (id < 0)
        Die ("mbamdParsCreateInstance failed");
    for (i=0; i<m->numParsSets; i++)
        if (mbamdParsSetSets (id, i, (const unsigned long long *) m->parsSets[i]) != BEAGLE_SUCCESS)
            Die ("mbamdParsSetSets failed");
    parsHandle[division] = id;
    FillOwner (&parsOwner[division], m);
    return id;
}

int MbamdParsActive (Tree *t)
{
    int         n;
    ModelInfo   *m;

    ReadEnv ();
    if (envOff == YES)
        return (NO);
    for (n=0; n<t->nRelParts; n++)
        {
        m = &modelSettings[t->relParts[n]];
        if (m->useBeagle == NO || m->parsSets == NULL || m->nParsIntsPerSite < 1 || m->nParsIntsPerSite > 2)
            return (NO);
        }
    return (YES);
}

static int CountInterior (TreeNode *p)
{
    if (p->left == NULL)
        return 0;
    return 1 + CountInterior (p->left) + CountInterior (p->right);
}

static int *DownOps (TreeNode *p, int *o)      /* GetParsDP's order: left subtree, right subtree, the node */
{
    if (p->left == NULL)
        return o;
    o = DownOps (p->left, o);
    o = DownOps (p->right, o);
    o[0] = p->index; o[1] = p->left->index; o[2] = p->right->index; o[3] = -1;
    return o + 4;
}

static int *FinalOps (TreeNode *p, int *o)     /* GetParsFP's order: the node, left subtree, right subtree */
{
    if (p->left == NULL)
        return o;
    o[0] = p->index; o[1] = p->left->index; o[2] = p->right->index; o[3] = p->anc->index;
    o = FinalOps (p->left, o + 4);
    return FinalOps (p->right, o);
}

/* check mode: the device's sets of the nodes an operation list wrote == the host's */
static void CompareSets (Tree *t, const int *ops, int n, const char *what)
{
    int                 i, d, c, words;
    unsigned long long  *buf;
    ModelInfo           *m;

    for (d=0; d<t->nRelParts; d++)
        {
        m = &modelSettings[t->relParts[d]];
        words = m->numChars * m->nParsIntsPerSite;
        buf = (unsigned long long *) malloc ((size_t) words * sizeof(unsigned long long));
        if (!buf)
            Die ("out of memory");
        for (i=0; i<n; i++)
            {
            if (mbamdParsGetSets (Handle(t->relParts[d]), ops[4*i], buf) != BEAGLE_SUCCESS)
                Die ("mbamdParsGetSets failed");
            for (c=0; c<words; c++)
                if (buf[c] != (unsigned long long) m->parsSets[ops[4*i]][c])
                    {
                    fprintf (stderr, "mbamd parsimony check: %s, division %d, node %d, word %d: device %llx host %llx\n",
                             what, t->relParts[d], ops[4*i], c, buf[c], (unsigned long long) m->parsSets[ops[4*i]][c]);
                    exit (1);
                    }
            nCompared++;
            }
        free (buf);
        }
}

MrBFlt MbamdGetParsDP (Tree *t, TreeNode *p, int chain)
{
    int         d, n, *ops;
    double      length, total;
    MrBFlt      host;
    ModelInfo   *m;

    if (MbamdParsActive (t) == NO)
        return GetParsDP (t, p, chain);
    n = CountInterior (p);
    ops = Ops (n);
    DownOps (p, ops);
    total = 0.0;
    for (d=0; d<t->nRelParts; d++)
        {
        m = &modelSettings[t->relParts[d]];
        if (envCheck == YES)
            {
            /* the weights GetFitchPartials uses for its return value (src/mcmc.c:4812) */
            if (mbamdParsSetPatternWeights (Handle(t->relParts[d]), numSitesOfPat + ((1 % chainParams.numChains) * numCompressedChars) + m->compCharStart) != BEAGLE_SUCCESS)
                Die ("mbamdParsSetPatternWeights failed");
            }
        if (mbamdParsDownPass (Handle(t->relParts[d]), ops, n, envCheck == YES ? &length : NULL) != BEAGLE_SUCCESS)
            Die ("mbamdParsDownPass failed");
        if (envCheck == YES)
            total += length;
        }
    if (envCheck == YES)
        {
        host = GetParsDP (t, p, chain);
        if (host != total)
            {
            fprintf (stderr, "mbamd parsimony check: down-pass length: device %.17g host %.17g\n", total, host);
            exit (1);
            }
        CompareSets (t, ops, n, "down-pass");
        return host;
        }
    return 0.0;
}

void MbamdGetParsFP (Tree *t, TreeNode *p, int chain)
{
    int     d, n, *ops;

    if (MbamdParsActive (t) == NO)
        {
        GetParsFP (t, p, chain);
        return;
        }
    n = CountInterior (p);
    ops = Ops (n);
    FinalOps (p, ops);
    for (d=0; d<t->nRelParts; d++)
        if (mbamdParsFinalPass (Handle(t->relParts[d]), ops, n) != BEAGLE_SUCCESS)
            Die ("mbamdParsFinalPass failed");
    if (envCheck == YES)
        {
        GetParsFP (t, p, chain);
        CompareSets (t, ops, n, "final pass");
        }
}

int MbamdParsLengths (Tree *t, int chain, int kind, TreeNode **pRoot, int nRoot, TreeNode **pCrown, int nCrown,
                      TreeNode *a, TreeNode *b, TreeNode *u, TreeNode *v, CLFlt *nSitesOfPat, MrBFlt warpFactor,
                      MrBFlt *parLength)
{
    int         i, j, d, n, *q;
    double      *len;
    ModelInfo   *m;

    (void) chain;
    n = nRoot * nCrown;
    if (n <= 0)                 /* no candidate: nothing to score (and malloc(0) may legally return NULL) */
        return (NO_ERROR);
    q = Ops (n);
    for (j=0; j<nCrown; j++)
        for (i=0; i<nRoot; i++, q+=4)
            {
            if (kind == 0)
                { q[0] = pRoot[i]->index; q[1] = pRoot[i]->anc->index; q[2] = v->index; q[3] = -1; }
            else if (kind == 1)
                { q[0] = u->index; q[1] = -1; q[2] = pCrown[j]->index; q[3] = pCrown[j]->anc->index; }
            else if (kind == 2)
                { q[0] = a->index; q[1] = b->index; q[2] = pCrown[j]->index; q[3] = pCrown[j]->anc->index; }
            else
                { q[0] = pRoot[i]->index; q[1] = pRoot[i]->anc->index; q[2] = pCrown[j]->index; q[3] = pCrown[j]->anc->index; }
            }
    len = (double *) malloc ((size_t) n * sizeof(double));
    if (!len)
        return (ERROR);
    for (i=0; i<n; i++)
        parLength[i] = 0.0;
    for (d=0; d<t->nRelParts; d++)
        {
        m = &modelSettings[t->relParts[d]];
        if (mbamdParsSetPatternWeights (Handle(t->relParts[d]), nSitesOfPat + m->compCharStart) != BEAGLE_SUCCESS ||
            mbamdParsScore (Handle(t->relParts[d]), opBuf, n, len) != BEAGLE_SUCCESS)
            {
            free (len);
            Die ("mbamdParsScore failed");
            }
        for (i=0; i<n; i++)
            parLength[i] += warpFactor * len[i];
        }
    free (len);
    return (NO_ERROR);
}

static double   *markedLen = NULL;         /* [relPart][node index] lengths of the marked nodes, MbamdParsMarkedLengths */
static int      markedParts = 0, markedNodes = 0;
static MrBFlt   markedKept = 0.0;

void MbamdParsMarkedLengths (Tree *t, int chain, TreeNode *v, CLFlt *nSitesOfPat)
{
    int         i, d, n, maxIndex, *q, *who;
    double      *len;
    TreeNode    *p;
    ModelInfo   *m;

    (void) chain;
    if (MbamdParsActive (t) == NO)
        return;
    n = 0;
    maxIndex = 0;
    for (i=0; i<t->nNodes; i++)
        {
        p = t->allDownPass[i];
        if (p->index > maxIndex)
            maxIndex = p->index;
        if (p->marked == YES)
            n++;
        }
    if (t->nRelParts * (maxIndex + 1) > markedParts * markedNodes)
        {
        markedLen = (double *) realloc (markedLen, (size_t) t->nRelParts * (size_t) (maxIndex + 1) * sizeof(double));
        if (!markedLen)
            Die ("out of memory");
        }
    markedParts = t->nRelParts;
    markedNodes = maxIndex + 1;
    if (n == 0)
        return;
    q = Ops (n);
    who = (int *) malloc ((size_t) n * sizeof(int));
    len = (double *) malloc ((size_t) n * sizeof(double));
    if (!who || !len)
        Die ("out of memory");
    n = 0;
    for (i=0; i<t->nNodes; i++)
        {
        p = t->allDownPass[i];
        if (p->marked == NO)
            continue;
        q[4*n] = p->index; q[4*n+1] = p->anc->index; q[4*n+2] = v->index; q[4*n+3] = -1;
        who[n++] = p->index;
        }
    for (d=0; d<t->nRelParts; d++)
        {
        m = &modelSettings[t->relParts[d]];
        if (mbamdParsSetPatternWeights (Handle(t->relParts[d]), nSitesOfPat + m->compCharStart) != BEAGLE_SUCCESS ||
            mbamdParsScore (Handle(t->relParts[d]), opBuf, n, len) != BEAGLE_SUCCESS)
            Die ("mbamdParsScore failed");
        for (i=0; i<n; i++)
            markedLen[(size_t) d * markedNodes + who[i]] = len[i];
        }
    free (who);
    free (len);
}

int MbamdParsMarkedLength (Tree *t, int n, TreeNode *p, MrBFlt *length)
{
    if (MbamdParsActive (t) == NO)
        return (NO);
    if (n < 0 || n >= markedParts || p->index < 0 || p->index >= markedNodes)
        Die ("MbamdParsMarkedLength: lengths were not prepared for this node");
    if (envCheck == YES)
        {
        markedKept = markedLen[(size_t) n * markedNodes + p->index];
        return (NO);
        }
    *length = markedLen[(size_t) n * markedNodes + p->index];
    return (YES);
}

void MbamdParsMarkedCheck (Tree *t, int n, TreeNode *p, MrBFlt length)
{
    if (envCheck == NO || MbamdParsActive (t) == NO)
        return;
    if (length != markedKept)
        {
        fprintf (stderr, "mbamd parsimony check: marked node %d, part %d: device %.17g host %.17g\n", p->index, n, markedKept, length);
        exit (1);
        }
    nCompared++;
}

int MbamdParsHostToo (Tree *t, MrBFlt *parLength, int n)
{
    if (MbamdParsActive (t) == NO)
        return (YES);
    if (envCheck == NO)
        return (NO);
    keptLengths = (MrBFlt *) realloc (keptLengths, (size_t) n * sizeof(MrBFlt));
    if (!keptLengths)
        Die ("out of memory");
    memcpy (keptLengths, parLength, (size_t) n * sizeof(MrBFlt));
    keptN = n;
    return (YES);
}

void MbamdParsCompare (MrBFlt *parLength, int n)
{
    int     i;

    if (envCheck == NO || keptN != n)
        {
        keptN = 0;
        return;
        }
    for (i=0; i<n; i++)
        if (parLength[i] != keptLengths[i])
            {
            fprintf (stderr, "mbamd parsimony check: candidate %d of %d: device %.17g host %.17g\n", i, n, keptLengths[i], parLength[i]);
            exit (1);
            }
    nCompared += n;
    keptN = 0;
}
