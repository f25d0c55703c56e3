/*
 * mbamd_compress_glue.c -- hash-table lookup of identical data columns for CompressData (see mbamd_compress_glue.h).
 * Compiled and linked with the reference's own sources; our code, no reference source in it.
 */
#include "bayes.h"
#include "mbamd_compress_glue.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    unsigned long long  hash;
    int                 column;        /* first column of a kept pattern, or -1: empty */
} Slot;

static Slot     *table = NULL;
static size_t   tableSize = 0, tableUsed = 0;
static int      tableFirst = -1;       /* the division (its first compressed column) the table belongs to */
static int      envRead = NO, envOff = NO, envCheck = NO;
static long     nCompared = 0;

static void ReadEnv (void)
{
    const char *s;
    if (envRead == YES)
        return;
    envRead = YES;
    s = getenv("MBAMD_HASH_COMPRESS");
    if (s != NULL && s[0] == '0')
        envOff = YES;
    s = getenv("MBAMD_COMPRESS_CHECK");
    if (s != NULL && s[0] != '0')
        envCheck = YES;
}

static void Report (void)
{
    if (envCheck == YES)
        fprintf (stderr, "mbamd compression check: %ld columns, the reference's search agreed on all of them\n", nCompared);
}

int MbamdCompressActive (void)
{
    ReadEnv ();
    return envOff == YES ? NO : YES;
}

int MbamdCompressCheckWanted (void)
{
    ReadEnv ();
    return envCheck;
}

void MbamdCompressCompare (int isSameHost, int whereHost, int isSame, int where)
{
    static int registered = NO;
    if (registered == NO)
        {
        registered = YES;
        atexit (Report);
        }
    if (isSameHost != isSame || (isSame == YES && whereHost != where))
        {
        fprintf (stderr, "mbamd compression check: reference search says (%d, column %d), hash table says (%d, column %d)\n",
                 isSameHost, whereHost, isSame, where);
        exit (1);
        }
    nCompared++;
}

static unsigned long long HashColumns (const BitsLong *matrix, int rowLength, int nRows, int width, int column)
{
    unsigned long long  h = 0x9E3779B97F4A7C15ull;
    int                 j, k;
    for (j=0; j<nRows; j++)
        for (k=0; k<width; k++)
            {
            h ^= (unsigned long long) matrix[(size_t) j * rowLength + column + k];
            h *= 0xFF51AFD7ED558CCDull;
            h ^= h >> 29;
            }
    return h;
}

static int SameColumns (const BitsLong *matrix, int rowLength, int nRows, int width, int a, int b)
{
    int j, k;
    for (j=0; j<nRows; j++)
        for (k=0; k<width; k++)
            if (matrix[(size_t) j * rowLength + a + k] != matrix[(size_t) j * rowLength + b + k])
                return (NO);
    return (YES);
}

static void Insert (Slot *t, size_t size, unsigned long long hash, int column)
{
    size_t at = (size_t) (hash & (size - 1));
    while (t[at].column >= 0)
        at = (at + 1) & (size - 1);
    t[at].hash = hash;
    t[at].column = column;
}

int MbamdFindSamePattern (BitsLong *matrix, int rowLength, int nRows, int width, int first, int newColumn, int *where)
{
    unsigned long long  hash;
    size_t              at, i;

    if (tableFirst != first || newColumn == first || table == NULL)
        {
        /* a new division (or a new call of CompressData): start an empty table */
        if (table == NULL)
            {
            tableSize = 1024;
            table = (Slot *) malloc (tableSize * sizeof(Slot));
            if (!table)
                { fprintf (stderr, "mbamd compression: out of memory\n"); exit (1); }
            }
        for (i=0; i<tableSize; i++)
            table[i].column = -1;
        tableUsed = 0;
        tableFirst = first;
        }
    hash = HashColumns (matrix, rowLength, nRows, width, newColumn);
    at = (size_t) (hash & (tableSize - 1));
    while (table[at].column >= 0)
        {
        if (table[at].hash == hash && SameColumns (matrix, rowLength, nRows, width, table[at].column, newColumn) == YES)
            {
            *where = table[at].column;
            return (YES);
            }
        at = (at + 1) & (tableSize - 1);
        }
    /* not there: the caller keeps it at newColumn */
    if (2 * (tableUsed + 1) > tableSize)
        {
        Slot    *bigger = (Slot *) malloc (2 * tableSize * sizeof(Slot));
        if (!bigger)
            { fprintf (stderr, "mbamd compression: out of memory\n"); exit (1); }
        for (i=0; i<2*tableSize; i++)
            bigger[i].column = -1;
        for (i=0; i<tableSize; i++)
            if (table[i].column >= 0)
                Insert (bigger, 2 * tableSize, table[i].hash, table[i].column);
        free (table);
        table = bigger;
        tableSize *= 2;
        }
    Insert (table, tableSize, hash, newColumn);
    tableUsed++;
    *where = newColumn;
    return (NO);
}
