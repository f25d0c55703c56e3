/* Self-test of integration/mpi_shim (run under mbamd_mpirun -n N): every function of the shim, large messages in both
 * directions at once (no deadlock on full socket buffers), ordering between a pair, reductions in rank order. */
#include "mpi.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "rank %d: check failed: %s (line %d)\n", rank, #c, __LINE__); exit(1); } } while (0)

int main(int argc, char **argv)
{
    int rank, size, i, j;
    MPI_Init(&argc, &argv);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);

    /* broadcast, reduce, allreduce */
    long seed = rank == 0 ? 12345 : -1;
    MPI_Bcast(&seed, 1, MPI_LONG, 0, MPI_COMM_WORLD);
    CHECK(seed == 12345);
    double x = rank + 0.5, sum = -1.0;
    MPI_Reduce(&x, &sum, 1, MPI_DOUBLE, MPI_SUM, 0, MPI_COMM_WORLD);
    if (rank == 0) CHECK(sum == size * (size - 1) / 2.0 + 0.5 * size);
    int v[3] = {rank, -rank, 1}, w[3];
    MPI_Allreduce(v, w, 3, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
    CHECK(w[0] == size * (size - 1) / 2 && w[1] == -w[0] && w[2] == size);
    MPI_Allreduce(v, w, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
    CHECK(w[0] == size - 1);
    MPI_Allreduce(v, w, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
    CHECK(w[0] == 0);

    if (size > 1) {
        /* every pair exchanges 2 MiB in both directions at the same time: Isend / Irecv / Waitall */
        const int n = 1 << 18;                     /* doubles */
        double *out = malloc(sizeof(double) * n), *in = malloc(sizeof(double) * n * (size_t) size);
        MPI_Request *req = malloc(sizeof(MPI_Request) * 2 * (size_t) size);
        MPI_Status *st = malloc(sizeof(MPI_Status) * 2 * (size_t) size);
        int nreq = 0;
        for (i = 0; i < n; i++) out[i] = rank * 1000003.0 + i;
        for (j = 0; j < size; j++) {
            if (j == rank) continue;
            MPI_Irecv(in + (size_t) j * n, n, MPI_DOUBLE, j, 7, MPI_COMM_WORLD, &req[nreq++]);
            MPI_Isend(out, n, MPI_DOUBLE, j, 7, MPI_COMM_WORLD, &req[nreq++]);
        }
        MPI_Waitall(nreq, req, st);
        for (j = 0; j < size; j++)
            if (j != rank)
                for (i = 0; i < n; i += 4099) CHECK(in[(size_t) j * n + i] == j * 1000003.0 + i);
        /* messages between a pair arrive in order per tag, and tags are matched out of arrival order */
        if (rank == 0) {
            int a = 1, b = 2, c = 3;
            MPI_Send(&a, 1, MPI_INT, 1, 100, MPI_COMM_WORLD);
            MPI_Send(&b, 1, MPI_INT, 1, 200, MPI_COMM_WORLD);
            MPI_Send(&c, 1, MPI_INT, 1, 100, MPI_COMM_WORLD);
        } else if (rank == 1) {
            int a = 0, b = 0, c = 0;
            MPI_Status s;
            MPI_Recv(&b, 1, MPI_INT, 0, 200, MPI_COMM_WORLD, &s);
            MPI_Recv(&a, 1, MPI_INT, 0, 100, MPI_COMM_WORLD, &s);
            MPI_Recv(&c, 1, MPI_INT, 0, 100, MPI_COMM_WORLD, &s);
            CHECK(a == 1 && b == 2 && c == 3 && s.MPI_SOURCE == 0 && s.MPI_TAG == 100);
        }
        /* strings to rank 0, the way the reference gathers print lines (src/mcmc.c:12050-12080) */
        char line[64];
        if (rank == 0) {
            for (j = 1; j < size; j++) {
                MPI_Status s;
                MPI_Recv(line, 64, MPI_CHAR, j, 9, MPI_COMM_WORLD, &s);
                char want[64];
                snprintf(want, sizeof want, "hello from %d", j);
                CHECK(strcmp(line, want) == 0);
            }
        } else {
            snprintf(line, sizeof line, "hello from %d", rank);
            MPI_Send(line, (int) strlen(line) + 1, MPI_CHAR, 0, 9, MPI_COMM_WORLD);
        }
        free(out); free(in); free(req); free(st);
    }
    MPI_Barrier(MPI_COMM_WORLD);
    if (rank == 0) printf("mpi shim ok: %d ranks\n", size);
    MPI_Finalize();
    return 0;
}
