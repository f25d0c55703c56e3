/*
 * mbamd_eigen_glue.h -- what a MrBayes maintainer adds so that the eigen-systems of a division on the engine are computed on
 * the device: in src/likelihood.c, UpDateCijk (:10476-10804), the three calls
 *       isComplex = GetEigens (n, q[..], eigenValues, ...)
 * become
 *       isComplex = MbamdGetEigens (m, whichChain, q, <part>, n, q[..], eigenValues, ...)
 * (integration/mrbayes/patches/patch_eigen.py does exactly this to a temporary copy; oracle/Makefile: ref-amd-full).  For a division that does not
 * run on the engine -- or MBAMD_DEVICE_EIGEN=0 -- the call IS GetEigens.  Otherwise the first part's call sends ALL rate
 * matrices of the division (the omega classes of a codon model, the rate categories of a covarion model: the reference builds
 * them all before it decomposes the first, src/likelihood.c:10688-10716) to mbamdSetRateMatricesFrom in one asynchronous
 * launch, warm-started from the eigenvectors of the chain's current state, and shields the eigen buffers against the host
 * result UpDateCijk sends afterwards (beagleSetEigenDecomposition, :10652 / :10752): the host never decomposes anything.
 * Our code against the reference's public types; no reference source in it.
 */
#ifndef MBAMD_EIGEN_GLUE_H_
#define MBAMD_EIGEN_GLUE_H_

int     MbamdGetEigens (ModelInfo *m, int chain, MrBFlt ***allQ, int part, int n, MrBFlt **q, MrBFlt *eigenValues, MrBFlt *eigvalsImag,
                        MrBFlt **eigvecs, MrBFlt **inverseEigvecs, MrBComplex **Ceigvecs, MrBComplex **CinverseEigvecs);

#endif
