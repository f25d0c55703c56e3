/*
 * mbamd_compress_glue.h -- site-pattern compression in O(columns x taxa) instead of O(columns^2 x taxa) (SURVEY 8(f) row 4,
 * second half): what a MrBayes maintainer adds to CompressData (src/model.c:2466-2720).  Our code, written against the
 * reference's public types; it contains no reference source.  See INTEGRATION.md, "Pattern compression".
 *
 * The reference decides "is this column a site pattern we already have?" by comparing it with every pattern kept so far
 * (src/model.c:2624-2645): 76 % of the start-up time at 500 x 20 000, about a minute at 1000 x 50 000.  The binding
 * replaces that search by a hash table over the columns kept so far (exact comparison on a hash hit, so the answer -- the
 * FIRST identical kept column, or none -- is the reference's):
 *
 *     // is it unique?
 *     isSame = NO;
 *     if (mp->dataType != CONTINUOUS && MbamdCompressActive () == YES)
 *         isSame = MbamdFindSamePattern (tempMatrix, numLocalChar, numLocalTaxa, m->nCharsPerSite, m->compMatrixStart, newColumn, &i);
 *     else if (mp->dataType != CONTINUOUS)
 *         { ...the reference's search, unchanged... }
 *
 * MBAMD_HASH_COMPRESS=0 switches it off; MBAMD_COMPRESS_CHECK=1 runs the reference's search as well and compares.
 * integration/mrbayes/patches/patch_pars.py applies the edit to a temporary copy of model.c for the _ref/mb_*_pars binaries.
 */
#ifndef MBAMD_COMPRESS_GLUE_H_
#define MBAMD_COMPRESS_GLUE_H_

int     MbamdCompressActive (void);
/* matrix: row-major [taxon][rowLength] BitsLong; the candidate is the `width` columns at `newColumn`; kept patterns are
 * the column groups first, first+width, ... below newColumn.  Returns YES and the matching column in *where, or NO -- and
 * then remembers the candidate as kept (the caller appends it at newColumn, src/model.c:2657-2670). */
int     MbamdFindSamePattern (BitsLong *matrix, int rowLength, int nRows, int width, int first, int newColumn, int *where);
/* MBAMD_COMPRESS_CHECK=1: the reference's search ran too; compare its answer */
int     MbamdCompressCheckWanted (void);
void    MbamdCompressCompare (int isSameHost, int whereHost, int isSame, int where);

#endif
