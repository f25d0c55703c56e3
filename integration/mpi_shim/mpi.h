/*
 * mpi.h -- a single-node message-passing shim with exactly the MPI surface the reference uses (13 functions:
 * src/bayes.c:177-217, src/mcmc.c chain exchange / print gathers), so that the reference's own MPI build
 * (-DMPI_ENABLED: one rank per group of chains, GPU picked by rank, src/mbbeagle.c:201-207) runs N ranks x N GPUs of one
 * node without an MPI installation (SURVEY 8(e) option (i); this image has none).  Ranks are processes started by
 * mbamd_mpirun; they talk over a full mesh of UNIX socket pairs created before the fork.  Not a general MPI: one
 * communicator, blocking collectives built on point-to-point, no wildcards, no derived datatypes.
 * Chain states never touch the GPUs' memory here -- the reference exchanges a few doubles per swap attempt and strings
 * at print time (src/mcmc.c:603-700, 12011-12080, 13194-13510) -- so there is nothing for RCCL to carry.
 */
#ifndef MBAMD_MPI_SHIM_H_
#define MBAMD_MPI_SHIM_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
typedef struct MbamdMpiRequest *MPI_Request;
typedef struct {
    int MPI_SOURCE, MPI_TAG, MPI_ERROR;
    int bytes;
} MPI_Status;

#define MPI_COMM_WORLD 0
#define MPI_SUCCESS 0
#define MPI_ERR_OTHER 15

/* datatypes: the value is the element size in the high byte and a kind in the low byte */
#define MPI_CHAR          ((1 << 8) | 1)
#define MPI_INT           ((4 << 8) | 2)
#define MPI_LONG          ((8 << 8) | 3)
#define MPI_LONG_LONG     ((8 << 8) | 4)
#define MPI_UNSIGNED_LONG ((8 << 8) | 5)
#define MPI_FLOAT         ((4 << 8) | 6)
#define MPI_DOUBLE        ((8 << 8) | 7)

#define MPI_SUM 1
#define MPI_MAX 2
#define MPI_MIN 3

int MPI_Init(int *argc, char ***argv);
int MPI_Finalize(void);
int MPI_Comm_size(MPI_Comm comm, int *size);
int MPI_Comm_rank(MPI_Comm comm, int *rank);
int MPI_Barrier(MPI_Comm comm);
int MPI_Bcast(void *buf, int count, MPI_Datatype type, int root, MPI_Comm comm);
int MPI_Send(const void *buf, int count, MPI_Datatype type, int dest, int tag, MPI_Comm comm);
int MPI_Recv(void *buf, int count, MPI_Datatype type, int source, int tag, MPI_Comm comm, MPI_Status *status);
int MPI_Isend(const void *buf, int count, MPI_Datatype type, int dest, int tag, MPI_Comm comm, MPI_Request *request);
int MPI_Irecv(void *buf, int count, MPI_Datatype type, int source, int tag, MPI_Comm comm, MPI_Request *request);
int MPI_Waitall(int count, MPI_Request *requests, MPI_Status *statuses);
int MPI_Reduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype type, MPI_Op op, int root, MPI_Comm comm);
int MPI_Allreduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype type, MPI_Op op, MPI_Comm comm);

#ifdef __cplusplus
}
#endif
#endif
