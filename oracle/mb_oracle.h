/*
 * mb_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C (CPU, scalar) restatement of the reference likelihood path of NBISweden/MrBayes
 * (reference src/likelihood.c).  It exists so that tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg can check the HIP engine; nothing in the product
 * (mrbayes_amd/, include/) may include, link, import or call it.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks mbo_tree_loglike against
 * log-likelihoods produced by the real reference binaries (oracle/_ref/mb*, built from
 * /root/reference/src by oracle/Makefile) on the fixtures under tests/golden/.
 *
 * Storage follows the reference: conditional likelihoods / transition probabilities are
 * `float` (CLFlt, src/bayes.h:110), parameters and the final sum are `double` (MrBFlt).
 * Layouts are the scalar reference layouts: cl[k][c][i], ti[k][i][j] (row = from-state).
 */
#ifndef MB_ORACLE_H_
#define MB_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

#define MBO_TIME_MIN 1.0E-11   /* src/bayes.h:321 */
#define MBO_TIME_MAX 100.0     /* src/bayes.h:322 */
#define MBO_LIKE_EPSILON 1.0e-300 /* src/likelihood.c:44 */

/* CalcCijk, src/utils.c:9734: c[i][j][s] = U[i][s] * Uinv[s][j] */
void mbo_calc_cijk(int n, const double *u, const double *uinv, double *cijk);

/* TiProbs_Gen inner body, src/likelihood.c:9498-9545, for one branch and K categories.
 * t_k = length * rate[k]  (rate[k] already holds baseRate*catRate[k]*correctionFactor).
 * bs = stationary frequencies (only used past TIME_MAX).  out: float [K][n][n]. */
void mbo_tiprobs_gen(int n, int K, const double *eigvals, const double *cijk,
                     double length, const double *rate, const double *bs, float *out);

/* TiProbs_GenCov, src/likelihood.c:9568-9700: one eigen-system per category, no category rates.
 * eigvals: [K][n], cijk: [K][n^3]; t = length*baseRate*correctionFactor. */
void mbo_tiprobs_gencov(int n, int K, const double *eigvals, const double *cijk,
                        double t, const double *bs, float *out);

/* CondLikeDown_Gen, src/likelihood.c:204-375.  A child is either dense (cl != NULL,
 * float [K][P][n]) or a compact tip (states != NULL, int [P], value n = gap/missing) which takes
 * the reference's pre-gathered-column short-cut (src/likelihood.c:236-285). */
void mbo_condlike_down(int n, int K, int P,
                       const float *clL, const int *stL, const float *tiL,
                       const float *clR, const int *stR, const float *tiR,
                       float *clP);

/* CondLikeRoot_Gen, src/likelihood.c:2152-2395: three-way product (left, right, ancestor tip). */
void mbo_condlike_root(int n, int K, int P,
                       const float *clL, const int *stL, const float *tiL,
                       const float *clR, const int *stR, const float *tiR,
                       const float *clA, const int *stA, const float *tiA,
                       float *clP);

/* CondLikeScaler_Gen, src/likelihood.c:4939-4988: per-pattern max over (k,i), divide,
 * node scaler = (float)log(max), site scaler += node scaler. */
void mbo_condlike_scaler(int n, int K, int P, float *clP, float *scP, float *lnScaler);

/* Likelihood_Gen / Likelihood_NY98, src/likelihood.c:5764-5917, 6975-7040.
 * catw[k]: category weight ((1-pInvar)/K for rate categories, omegaCatFreq[k] for NY98).
 * pInvar>0 with clInvar != NULL adds the invariable-sites term exactly like the scalar reference
 * (src/likelihood.c:5866-5899).  Returns 0, or 1 when a site likelihood drops below
 * LIKE_EPSILON (the reference sets abortMove, src/likelihood.c:5852-5860).
 * siteLnL (optional, [P]): lnScaler[c] + log(like_c). */
int mbo_likelihood(int n, int K, int P, const float *clP, const double *bs, const double *catw,
                   const float *lnScaler, const float *nSitesOfPat,
                   double pInvar, const float *clInvar, double *lnL, double *siteLnL);

/* Closed-form 4x4 transition probabilities: TiProbs_Hky (src/likelihood.c:9709) and TiProbs_JukesCantor (:9846);
 * out[k][i][j], rate[k] = baseRate*catRate[k]. */
void mbo_tiprobs_hky(int K, double kap, const double *pis, double length, const double *rate, float *out);
void mbo_tiprobs_jc(int K, double length, const double *rate, float *out);

/* Whole-tree evaluation in the order of LaunchLogLikeForDivision (native back-end),
 * src/likelihood.c:7851-7972, all nodes dirty, rescale at every interior non-root node.
 *   left/right/length: [2N-2]; intDownPass: [N-2] post-order; rootTip: tip used as calculation
 *   root; rootLeft = top interior node (its `length` is the branch to rootTip).
 *   tipStates: [N][P] int (value n = missing) or, when tipIsPartial[t], tipPartials + t*P*n
 *   holds float [P][n] 0/1 (replicated over categories here, as InitChainCondLikes does).
 *   nEigen = 1 -> TiProbs_Gen with rate[K]; nEigen = K -> TiProbs_GenCov (rate[0] = scalar).
 *   useShortcuts = 0 expands compact tips to dense 0/1 partials first (what the SIMD kernels do).
 */
int mbo_tree_loglike(int n, int K, int P, int N,
                     const int *left, const int *right, const double *length,
                     const int *intDownPass, int rootTip, int rootLeft,
                     const int *tipStates, const int *tipIsPartial, const float *tipPartials,
                     int nEigen, const double *eigvals, const double *cijk, const double *rate,
                     const double *bs, const double *catw,
                     double pInvar, const float *clInvar, const float *nSitesOfPat,
                     int useShortcuts, double *lnL, double *siteLnL);

#ifdef __cplusplus
}
#endif
#endif
