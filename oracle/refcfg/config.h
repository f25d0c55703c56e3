/* Hand-written stand-in for the config.h that the reference's ./configure would generate
 * (reference src/config.h.in); used only by oracle/Makefile's `ref-amd*` / `ref-mpi` targets, which compile the
 * UNMODIFIED reference sources with BEAGLE support against include/libhmsbeagle/beagle.h (and / or its MPI code path).
 * Not a copy of any reference file: just the handful of macros src/bayes.h:4-19 expects. */
#ifndef MBAMD_REF_CONFIG_H_
#define MBAMD_REF_CONFIG_H_
#define PACKAGE_NAME "MrBayes"
#define PACKAGE_VERSION "3.2.8"
#define HOST_CPU "x86_64"
#define HOST_TYPE "x86_64-pc-linux-gnu"
#define COMPILER_VENDOR "gnu"
#define COMPILER_VERSION "oracle-recipe"
#define UNIX_VERSION 1
#define HAVE_UNISTD_H 1
#define HAVE_SSE 1
#define HAVE_AVX 1
#define HAVE_FMA3 1
#ifndef MBAMD_REFCFG_NO_BEAGLE
#define BEAGLE_ENABLED 1
#endif
#ifdef MBAMD_REFCFG_MPI          /* the reference's MPI build against integration/mpi_shim/mpi.h (oracle/Makefile: ref-mpi) */
#define MPI_ENABLED 1
#endif
/* BEAGLE_V3_ENABLED deliberately undefined: non-v3 surface (reference configure.ac:220-223) */
#endif
