/*
 * mbamd_mpirun -- start N ranks of a program built against the shim (mpi.h):   mbamd_mpirun -n N program [args...]
 * Creates the full mesh of socket pairs, forks, sets MBAMD_MPI_RANK / _SIZE / _FDS in each child and execs the program;
 * waits for all ranks and returns the first non-zero exit status (killing the others if a rank dies).
 */
#define _GNU_SOURCE
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

int main(int argc, char **argv)
{
    int n = 0, i, j, status = 0, left;
    int (*fd)[256];
    pid_t pid[256];
    if (argc < 4 || strcmp(argv[1], "-n") != 0 || (n = atoi(argv[2])) < 1 || n > 256) {
        fprintf(stderr, "usage: %s -n <ranks 1..256> program [args...]\n", argv[0]);
        return 2;
    }
    fd = calloc((size_t) n, sizeof *fd);
    for (i = 0; i < n; i++) {
        fd[i][i] = -1;
        for (j = i + 1; j < n; j++) {
            int sv[2];
            if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) { perror("socketpair"); return 2; }
            fd[i][j] = sv[0];
            fd[j][i] = sv[1];
        }
    }
    for (i = 0; i < n; i++) {
        pid[i] = fork();
        if (pid[i] < 0) {                               /* stop the ranks already started: they would wait for this one for ever */
            perror("fork");
            for (j = 0; j < i; j++) kill(pid[j], SIGTERM);
            for (j = 0; j < i; j++) (void) waitpid(pid[j], NULL, 0);
            return 2;
        }
        if (pid[i] == 0) {
            char buf[32], *list = malloc((size_t) n * 12 + 1), *p = list;
            int a, b;
            for (a = 0; a < n; a++)                     /* keep only this rank's ends */
                for (b = 0; b < n; b++)
                    if (a != i && a != b) close(fd[a][b]);
            for (b = 0; b < n; b++) p += sprintf(p, "%s%d", b ? "," : "", fd[i][b]);
            snprintf(buf, sizeof buf, "%d", i);
            setenv("MBAMD_MPI_RANK", buf, 1);
            snprintf(buf, sizeof buf, "%d", n);
            setenv("MBAMD_MPI_SIZE", buf, 1);
            setenv("MBAMD_MPI_FDS", list, 1);
            execvp(argv[3], argv + 3);
            perror(argv[3]);
            _exit(127);
        }
    }
    for (i = 0; i < n; i++)
        for (j = 0; j < n; j++)
            if (i != j) close(fd[i][j]);
    for (left = n; left > 0; left--) {
        int st = 0;
        pid_t who = wait(&st);
        int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if (who < 0) break;
        for (i = 0; i < n; i++)
            if (pid[i] == who) pid[i] = -1;             /* reaped: the number may belong to somebody else from now on */
        if (code != 0 && status == 0) {
            status = code;
            for (i = 0; i < n; i++)
                if (pid[i] > 0) kill(pid[i], SIGTERM);  /* exact PIDs of our own children that are still running */
        }
    }
    return status;
}
