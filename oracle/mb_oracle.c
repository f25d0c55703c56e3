/*
 * mb_oracle.c -- TEST INFRASTRUCTURE ONLY (see mb_oracle.h).
 *
 * Scalar CPU restatement of the MrBayes likelihood path, written from the algorithm in the
 * reference's src/likelihood.c (function/line citations at each routine).  Never linked into
 * the product library.
 */
#include "mb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/utils.c:9734 */
void mbo_calc_cijk(int n, const double *u, const double *uinv, double *cijk)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            for (int s = 0; s < n; s++)
                *cijk++ = u[i * n + s] * uinv[s * n + j];
}

/* one category of TiProbs_Gen / TiProbs_GenCov: src/likelihood.c:9498-9545 */
static void tiprobs_one(int n, const double *eigvals, const double *cijk, double t,
                        const double *bs, float *out)
{
    if (t < MBO_TIME_MIN) {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                *out++ = (i == j) ? 1.0f : 0.0f;
    } else if (t > MBO_TIME_MAX) {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                *out++ = (float) bs[j];
    } else {
        double ev[128];
        for (int s = 0; s < n; s++)
            ev[s] = exp(eigvals[s] * t);
        const double *c = cijk;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                double sum = 0.0;
                for (int s = 0; s < n; s++)
                    sum += (*c++) * ev[s];
                *out++ = (float) ((sum < 0.0) ? 0.0 : sum);
            }
    }
}

void mbo_tiprobs_gen(int n, int K, const double *eigvals, const double *cijk,
                     double length, const double *rate, const double *bs, float *out)
{
    for (int k = 0; k < K; k++)
        tiprobs_one(n, eigvals, cijk, length * rate[k], bs, out + (size_t) k * n * n);
}

/* Closed-form 4x4 transition probabilities of the nst=2 (HKY85 / F84-style kappa) and nst=1 (Jukes-Cantor) models:
 * TiProbs_Hky (src/likelihood.c:9709-9838) and TiProbs_JukesCantor (src/likelihood.c:9846-9960), t = length*rate[k]
 * (rate[k] = baseRate*catRate[k]), identity below TIME_MIN, stationary above TIME_MAX (src/bayes.h:321-322).
 * The BEAGLE seam never calls them (MrBayes sends an eigen-system for every nst), so these are the reference for
 * "the engine's eigen path reproduces the closed forms". */
#define MBO_TIME_MIN 1.0E-11
#define MBO_TIME_MAX 100.0
void mbo_tiprobs_hky(int K, double kap, const double *pis, double length, const double *rate, float *out)
{
    double beta = 0.5 / ((pis[0] + pis[2]) * (pis[1] + pis[3]) + kap * ((pis[0] * pis[2]) + (pis[1] * pis[3])));
    double bigPi_j[4] = {pis[0] + pis[2], pis[1] + pis[3], pis[0] + pis[2], pis[1] + pis[3]};
    int index = 0;
    for (int k = 0; k < K; k++) {
        double t = length * rate[k];
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                if (t < MBO_TIME_MIN) { out[index++] = (i == j) ? 1.0f : 0.0f; continue; }
                if (t > MBO_TIME_MAX) { out[index++] = (float) pis[j]; continue; }
                double bigPij = bigPi_j[j], pij = pis[j];
                double u = 1.0 / bigPij - 1.0;
                double w = -beta * (1.0 + bigPij * (kap - 1.0));
                double x = exp(-beta * t);
                double y = exp(w * t);
                double z = (bigPij - pij) / bigPij;
                if (i == j)
                    out[index++] = (float) (pij + pij * u * x + z * y);
                else if ((i == 0 && j == 2) || (i == 2 && j == 0) || (i == 1 && j == 3) || (i == 3 && j == 1))
                    out[index++] = (float) (pij + pij * u * x - (pij / bigPij) * y);
                else
                    out[index++] = (float) (pij * (1.0 - x));
            }
    }
}

void mbo_tiprobs_jc(int K, double length, const double *rate, float *out)
{
    int index = 0;
    for (int k = 0; k < K; k++) {
        double t = length * rate[k];
        float pChange = (float) (0.25 - 0.25 * exp(-(4.0 / 3.0) * t));
        float pNoChange = (float) (0.25 + 0.75 * exp(-(4.0 / 3.0) * t));
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                if (t < MBO_TIME_MIN) out[index++] = (i == j) ? 1.0f : 0.0f;
                else if (t > MBO_TIME_MAX) out[index++] = 0.25f;
                else out[index++] = (i == j) ? pNoChange : pChange;
            }
    }
}

void mbo_tiprobs_gencov(int n, int K, const double *eigvals, const double *cijk,
                        double t, const double *bs, float *out)
{
    for (int k = 0; k < K; k++)
        tiprobs_one(n, eigvals + (size_t) k * n, cijk + (size_t) k * n * n * n, t, bs,
                    out + (size_t) k * n * n);
}

/* Pre-gathered transition columns for a compact tip: pre[k][state][i] = ti[k][i][state],
 * plus an all-ones row for gap/missing.  src/likelihood.c:236-259 (numStates == numModelStates). */
static float *gather_columns(int n, int K, const float *ti)
{
    float *pre = (float *) malloc(sizeof(float) * (size_t) K * (n + 1) * n);
    float *p = pre;
    for (int k = 0; k < K; k++) {
        const float *t = ti + (size_t) k * n * n;
        for (int s = 0; s < n; s++)
            for (int i = 0; i < n; i++)
                *p++ = t[i * n + s];
        for (int i = 0; i < n; i++)
            *p++ = 1.0f;
    }
    return pre;
}

/* factor contributed by one child at (k, c): f[i] = sum_j ti[k][i][j] * cl[k][c][j]
 * (dense, src/likelihood.c:296-306) or the gathered column (src/likelihood.c:318-328). */
static inline void child_factor(int n, int k, int c, int P, const float *cl, const int *st,
                                const float *ti, const float *pre, float *f)
{
    if (st) {
        const float *col = pre + ((size_t) k * (n + 1) + st[c]) * n;
        for (int i = 0; i < n; i++)
            f[i] = col[i];
    } else {
        const float *t = ti + (size_t) k * n * n;
        const float *v = cl + ((size_t) k * P + c) * n;
        for (int i = 0; i < n; i++) {
            float acc = 0.0f;
            for (int j = 0; j < n; j++)
                acc += t[i * n + j] * v[j];
            f[i] = acc;
        }
    }
}

void mbo_condlike_down(int n, int K, int P,
                       const float *clL, const int *stL, const float *tiL,
                       const float *clR, const int *stR, const float *tiR,
                       float *clP)
{
    float *preL = stL ? gather_columns(n, K, tiL) : NULL;
    float *preR = stR ? gather_columns(n, K, tiR) : NULL;
    float fl[128], fr[128];
    for (int k = 0; k < K; k++)
        for (int c = 0; c < P; c++) {
            child_factor(n, k, c, P, clL, stL, tiL, preL, fl);
            child_factor(n, k, c, P, clR, stR, tiR, preR, fr);
            float *o = clP + ((size_t) k * P + c) * n;
            for (int i = 0; i < n; i++)
                o[i] = fl[i] * fr[i];
        }
    free(preL);
    free(preR);
}

void mbo_condlike_root(int n, int K, int P,
                       const float *clL, const int *stL, const float *tiL,
                       const float *clR, const int *stR, const float *tiR,
                       const float *clA, const int *stA, const float *tiA,
                       float *clP)
{
    float *preL = stL ? gather_columns(n, K, tiL) : NULL;
    float *preR = stR ? gather_columns(n, K, tiR) : NULL;
    float *preA = stA ? gather_columns(n, K, tiA) : NULL;
    float fl[128], fr[128], fa[128];
    for (int k = 0; k < K; k++)
        for (int c = 0; c < P; c++) {
            child_factor(n, k, c, P, clL, stL, tiL, preL, fl);
            child_factor(n, k, c, P, clR, stR, tiR, preR, fr);
            child_factor(n, k, c, P, clA, stA, tiA, preA, fa);
            float *o = clP + ((size_t) k * P + c) * n;
            for (int i = 0; i < n; i++)
                o[i] = fl[i] * fr[i] * fa[i];     /* src/likelihood.c:2268-2285 */
        }
    free(preL);
    free(preR);
    free(preA);
}

/* src/likelihood.c:4939-4988 */
void mbo_condlike_scaler(int n, int K, int P, float *clP, float *scP, float *lnScaler)
{
    for (int c = 0; c < P; c++) {
        float scaler = 0.0f;
        for (int k = 0; k < K; k++) {
            const float *v = clP + ((size_t) k * P + c) * n;
            for (int i = 0; i < n; i++)
                if (v[i] > scaler)
                    scaler = v[i];
        }
        for (int k = 0; k < K; k++) {
            float *v = clP + ((size_t) k * P + c) * n;
            for (int i = 0; i < n; i++)
                v[i] /= scaler;
        }
        scP[c] = (float) log(scaler);
        lnScaler[c] += scP[c];
    }
}

/* src/likelihood.c:5764-5917 (Gen) and 6975-7040 (NY98: same with per-class weights) */
int mbo_likelihood(int n, int K, int P, const float *clP, const double *bs, const double *catw,
                   const float *lnScaler, const float *nSitesOfPat,
                   double pInvar, const float *clInvar, double *lnL, double *siteLnL)
{
    double total = 0.0;
    for (int c = 0; c < P; c++) {
        double like = 0.0;
        for (int k = 0; k < K; k++) {
            const float *v = clP + ((size_t) k * P + c) * n;
            double catLike = 0.0;
            for (int j = 0; j < n; j++)
                catLike += v[j] * bs[j];
            like += catLike * catw[k];
        }
        double lnLike;
        if (clInvar != NULL && pInvar > 0.0) {
            double likeI = 0.0;
            for (int j = 0; j < n; j++)
                likeI += clInvar[(size_t) c * n + j] * bs[j] * pInvar;
            if (lnScaler[c] < -200.0) {
                if (likeI > 1E-70)
                    lnLike = log(likeI);
                else
                    lnLike = log(like) + lnScaler[c];
            } else
                lnLike = log(like + (likeI / exp(lnScaler[c]))) + lnScaler[c];
        } else
            lnLike = lnScaler[c] + log(like);
        if (like < MBO_LIKE_EPSILON) {
            *lnL = -1.7976931348623157e308;
            return 1;
        }
        if (siteLnL)
            siteLnL[c] = lnLike;
        total += lnLike * nSitesOfPat[c];
    }
    *lnL = total;
    return 0;
}

/* src/likelihood.c:7851-7972 (native back-end), everything dirty */
int mbo_tree_loglike(int n, int K, int P, int N,
                     const int *left, const int *right, const double *length,
                     const int *intDownPass, int rootTip, int rootLeft,
                     const int *tipStates, const int *tipIsPartial, const float *tipPartials,
                     int nEigen, const double *eigvals, const double *cijk, const double *rate,
                     const double *bs, const double *catw,
                     double pInvar, const float *clInvar, const float *nSitesOfPat,
                     int useShortcuts, double *lnL, double *siteLnL)
{
    const int nNodes = 2 * N - 2;
    const size_t clLen = (size_t) K * P * n;
    const size_t tiLen = (size_t) K * n * n;
    float **cl = (float **) calloc(nNodes, sizeof(float *));
    float *ti = (float *) malloc(sizeof(float) * tiLen * nNodes);
    float *lnScaler = (float *) calloc(P, sizeof(float));
    float *scP = (float *) malloc(sizeof(float) * P);
    const int **st = (const int **) calloc(nNodes, sizeof(int *));

    /* tip conditional likelihoods: 1.0 for every compatible state, same for all categories
       (InitChainCondLikes, src/mcmc.c:6302-6420) */
    for (int t = 0; t < N; t++) {
        int dense = (tipIsPartial && tipIsPartial[t]) || !useShortcuts;
        if (!dense) {
            st[t] = tipStates + (size_t) t * P;
            continue;
        }
        cl[t] = (float *) malloc(sizeof(float) * clLen);
        for (int k = 0; k < K; k++)
            for (int c = 0; c < P; c++) {
                float *o = cl[t] + ((size_t) k * P + c) * n;
                if (tipIsPartial && tipIsPartial[t]) {
                    memcpy(o, tipPartials + ((size_t) t * P + c) * n, sizeof(float) * n);
                } else {
                    int s = tipStates[(size_t) t * P + c];
                    for (int i = 0; i < n; i++)
                        o[i] = (s == n || s == i) ? 1.0f : 0.0f;
                }
            }
    }

    /* transition probabilities for every branch (m->TiProbs) */
    for (int v = 0; v < nNodes; v++) {
        if (v == rootTip)
            continue;
        if (nEigen == 1)
            mbo_tiprobs_gen(n, K, eigvals, cijk, length[v], rate, bs, ti + tiLen * v);
        else
            mbo_tiprobs_gencov(n, K, eigvals, cijk, length[v] * rate[0], bs, ti + tiLen * v);
    }

    for (int a = 0; a < N - 2; a++) {
        int p = intDownPass[a];
        int l = left[p], r = right[p];
        cl[p] = (float *) malloc(sizeof(float) * clLen);
        if (p == rootLeft) {
            mbo_condlike_root(n, K, P, cl[l], st[l], ti + tiLen * l, cl[r], st[r], ti + tiLen * r,
                              cl[rootTip], st[rootTip], ti + tiLen * p, cl[p]);
        } else {
            mbo_condlike_down(n, K, P, cl[l], st[l], ti + tiLen * l, cl[r], st[r], ti + tiLen * r, cl[p]);
            mbo_condlike_scaler(n, K, P, cl[p], scP, lnScaler);   /* rescaleFreq = 1, src/mcmc.c:6156-6163 */
        }
    }

    int rc = mbo_likelihood(n, K, P, cl[rootLeft], bs, catw, lnScaler, nSitesOfPat,
                            pInvar, clInvar, lnL, siteLnL);

    for (int v = 0; v < nNodes; v++)
        free(cl[v]);
    free(cl);
    free(ti);
    free(lnScaler);
    free(scP);
    free((void *) st);
    return rc;
}
