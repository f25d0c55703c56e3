/*
 * pars_oracle.c -- TEST INFRASTRUCTURE ONLY (see mb_oracle.h): plain-C restatement of the reference's Fitch-parsimony
 * routines on `BitsLong` state sets, the checker of the device scorer (include/libhmsbeagle/mbamd_parsimony.h).
 *
 * Parity status: PINNED -- (0) golden vectors from the reference's own parsimony-model likelihood (Likelihood_Pars,
 * src/likelihood.c:7593-7700; tests/golden/parsmodel.json written by tools/gen_golden_pars.py from oracle/_ref/mb with
 * `lset parsmodel=yes`): tests/test_oracle_golden.py; (1) tests/test_mrbayes_dropin.py runs the reference binary with the device binding
 * (oracle/_ref/mb_emu_pars / mb_amd_pars, integration/mrbayes/mbamd_pars_glue.c) under MBAMD_PARS_CHECK=1, where every
 * GetParsDP / GetParsFP / candidate loop is ALSO executed by the reference's own host functions and the sets and lengths
 * are compared word for word inside the process; (2) the same binary must reproduce the unpatched binary's MCMC
 * trajectory; (3) this restatement is compared with the device on random forests (tests/engine_checks.py).
 *
 * sets: [setCount][P * words] 64-bit words (m->parsSets, reference src/mcmc.c:6887-6895); words = nParsIntsPerSite.
 */
#include <stdint.h>
#include <stddef.h>

typedef uint64_t BitsLong;

/* GetFitchPartials (src/mcmc.c:4794-4846) applied to the operations of a down-pass in GetParsDP's order (:4849-4876).
 * ops: n x {destination, source1, source2, unused}; nodeLen (may be NULL): the length each operation added. */
double mbo_pars_down(BitsLong *sets, int P, int words, const int *ops, int n, const float *w, double *nodeLen)
{
    double total = 0.0;
    for (int i = 0; i < n; ++i) {
        BitsLong *pD = sets + (size_t) ops[4 * i] * P * words;
        const BitsLong *pS1 = sets + (size_t) ops[4 * i + 1] * P * words, *pS2 = sets + (size_t) ops[4 * i + 2] * P * words;
        double length = 0.0;
        if (words == 1) {
            for (int c = 0; c < P; ++c) {
                BitsLong x = pS1[c] & pS2[c];
                if (x == 0) {
                    length += w[c];
                    x = pS1[c] | pS2[c];
                }
                pD[c] = x;
            }
        } else {
            for (int c = 0, j = 0; c < P; ++c, j += 2) {
                BitsLong x0 = pS1[j] & pS2[j], x1 = pS1[j + 1] & pS2[j + 1];
                if ((x0 | x1) == 0) {
                    length += w[c];
                    x0 = pS1[j] | pS2[j];
                    x1 = pS1[j + 1] | pS2[j + 1];
                }
                pD[j] = x0;
                pD[j + 1] = x1;
            }
        }
        if (nodeLen) nodeLen[i] = length;
        total += length;
    }
    return total;
}

/* GetParsFP (src/mcmc.c:4881-4954) over a pre-order node list: n x {node, left, right, ancestor}, in place. */
void mbo_pars_final(BitsLong *sets, int P, int words, const int *ops, int n)
{
    for (int i = 0; i < n; ++i) {
        BitsLong *pP = sets + (size_t) ops[4 * i] * P * words;
        const BitsLong *pL = sets + (size_t) ops[4 * i + 1] * P * words, *pR = sets + (size_t) ops[4 * i + 2] * P * words;
        const BitsLong *pA = sets + (size_t) ops[4 * i + 3] * P * words;
        if (words == 1) {
            for (int c = 0; c < P; ++c) {
                BitsLong x = pP[c] & pA[c];
                if (x != pA[c]) {
                    if ((pL[c] & pR[c]) != 0)
                        x = ((pL[c] | pR[c]) & pA[c]) | pP[c];
                    else
                        x = pP[c] | pA[c];
                }
                pP[c] = x;
            }
        } else {
            for (int c = 0, j = 0; c < P; ++c, j += 2) {
                BitsLong x0 = pP[j] & pA[j], x1 = pP[j + 1] & pA[j + 1];
                if (x0 != pA[j] || x1 != pA[j + 1]) {
                    x0 = pL[j] & pR[j];
                    x1 = pL[j + 1] & pR[j + 1];
                    if ((x0 | x1) != 0) {
                        x0 = ((pL[j] | pR[j]) & pA[j]) | pP[j];
                        x1 = ((pL[j + 1] | pR[j + 1]) & pA[j + 1]) | pP[j + 1];
                    } else {
                        x0 = pP[j] | pA[j];
                        x1 = pP[j + 1] | pA[j + 1];
                    }
                }
                pP[j] = x0;
                pP[j + 1] = x1;
            }
        }
    }
}

/* Candidate lengths of the parsimony-biased moves (src/proposal.c:10783-10876, 13430-13472): for every tuple
 * {a, b, c, d} (-1 = no set) the summed weight of the patterns with ((A | B) & (C | D)) == 0. */
void mbo_pars_score(const BitsLong *sets, int P, int words, const int *tuples, int n, const float *w, double *out)
{
    for (int i = 0; i < n; ++i) {
        const BitsLong *q[4];
        for (int k = 0; k < 4; ++k) q[k] = tuples[4 * i + k] >= 0 ? sets + (size_t) tuples[4 * i + k] * P * words : NULL;
        double length = 0.0;
        for (int c = 0; c < P; ++c) {
            BitsLong any = 0;
            for (int v = 0; v < words; ++v) {
                const size_t j = (size_t) c * words + v;
                const BitsLong x = (q[0] ? q[0][j] : 0) | (q[1] ? q[1][j] : 0), y = (q[2] ? q[2][j] : 0) | (q[3] ? q[3][j] : 0);
                any |= x & y;
            }
            if (any == 0) length += w[c];
        }
        out[i] = length;
    }
}
