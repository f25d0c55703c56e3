/*
 * mbamd_mpi.c -- the shim behind mpi.h (see there).  Every rank holds one non-blocking stream socket per peer (file
 * descriptors inherited from mbamd_mpirun: MBAMD_MPI_RANK, MBAMD_MPI_SIZE, MBAMD_MPI_FDS="fd,fd,...", -1 for itself).
 * A message is a 12-byte header {tag, bytes, magic} and its payload.  All progress happens inside calls: pending sends
 * are flushed and whatever has arrived is read into per-peer queues, so two ranks that send to each other before either
 * receives cannot deadlock on full socket buffers.  Messages between a pair stay in order (MPI's non-overtaking rule).
 */
#define _GNU_SOURCE
#include "mpi.h"

#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define MAGIC 0x4d42414d
#define TAG_BARRIER (-101)
#define TAG_BCAST (-102)
#define TAG_REDUCE (-103)

typedef struct Msg {
    int tag, bytes;
    char *data;
    struct Msg *next;
} Msg;

typedef struct Out {
    char *data;                 /* header + payload */
    size_t len, done;
    struct Out *next;
} Out;

typedef struct {
    int fd;
    Msg *head, *tail;           /* arrived, not yet matched */
    Out *ohead, *otail;         /* to be written */
    /* receive state machine */
    int hdr[3];
    size_t hdrDone;
    Msg *cur;
    size_t curDone;
    int closed;                 /* read() returned 0: the peer has exited or closed its end; nothing more will come */
} Peer;

struct MbamdMpiRequest {
    int isRecv, peer, tag, bytes, done;
    void *buf;
    Out *out;                   /* send: complete when this buffer has been written */
    int got;                    /* receive: bytes delivered */
};

static int g_rank = 0, g_size = 1, g_init = 0;
static long g_moved = 0;     /* bytes read or written so far: did a progress call change anything? */
static Peer *g_peer = NULL;

static void die(const char *what)
{
    fprintf(stderr, "mbamd_mpi[rank %d]: %s (%s)\n", g_rank, what, strerror(errno));
    _exit(70);
}

static void die_peer(int peer)
{
    fprintf(stderr, "mbamd_mpi[rank %d]: rank %d exited (or closed its socket) while a message from it was awaited\n", g_rank, peer);
    _exit(71);
}

static int elem(MPI_Datatype t) { return t >> 8; }

/* write as much of the pending output of peer p as the socket takes */
static void flush_peer(Peer *p)
{
    while (p->ohead) {
        Out *o = p->ohead;
        ssize_t n = write(p->fd, o->data + o->done, o->len - o->done);
        if (n < 0) {
            if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) return;
            die("write to a peer failed (did a rank exit?)");
        }
        o->done += (size_t) n;
        g_moved += n;
        if (o->done < o->len) return;
        p->ohead = o->next;
        if (!p->ohead) p->otail = NULL;
        free(o->data);
        o->data = NULL;         /* the owner (send_blocking / MPI_Waitall) sees data == NULL and frees the node */
    }
}

/* read whatever peer p has sent into its queue */
static void drain_peer(Peer *p)
{
    for (;;) {
        if (!p->cur) {
            ssize_t n = read(p->fd, (char *) p->hdr + p->hdrDone, sizeof p->hdr - p->hdrDone);
            if (n < 0) {
                if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) return;
                die("read from a peer failed");
            }
            if (n == 0) { p->closed = 1; return; }      /* peer closed: nothing more will come (the waiters check) */
            p->hdrDone += (size_t) n;
            g_moved += n;
            if (p->hdrDone < sizeof p->hdr) return;
            if (p->hdr[2] != MAGIC) die("corrupt message header");
            p->cur = (Msg *) calloc(1, sizeof(Msg));
            p->cur->tag = p->hdr[0];
            p->cur->bytes = p->hdr[1];
            p->cur->data = (char *) malloc(p->cur->bytes > 0 ? (size_t) p->cur->bytes : 1);
            if (!p->cur->data) die("out of memory");
            p->curDone = 0;
            p->hdrDone = 0;
        }
        while (p->curDone < (size_t) p->cur->bytes) {
            ssize_t n = read(p->fd, p->cur->data + p->curDone, (size_t) p->cur->bytes - p->curDone);
            if (n < 0) {
                if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) return;
                die("read from a peer failed");
            }
            if (n == 0) die("a peer closed its socket in the middle of a message");
            p->curDone += (size_t) n;
            g_moved += n;
        }
        if (p->tail) p->tail->next = p->cur; else p->head = p->cur;
        p->tail = p->cur;
        p->cur = NULL;
    }
}

/* move what can be moved; if nothing could (and the caller has nothing better to do) sleep until a socket is ready */
static void progress(int block)
{
    struct pollfd fds[256];
    int i, n = 0;
    const long before = g_moved;
    for (i = 0; i < g_size; i++) {
        if (i == g_rank) continue;
        flush_peer(&g_peer[i]);
        drain_peer(&g_peer[i]);
    }
    if (!block || g_moved != before) return;          /* something arrived or left: let the caller look again first */
    {   /* ranks meet every generation (swap attempts): spin a little before paying for a sleep and a wake-up */
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (;;) {
            for (i = 0; i < g_size; i++) {
                if (i == g_rank) continue;
                flush_peer(&g_peer[i]);
                drain_peer(&g_peer[i]);
            }
            if (g_moved != before) return;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 200000L) break;
        }
    }
    for (i = 0; i < g_size && n < 256; i++) {
        if (i == g_rank || g_peer[i].closed) continue;  /* (a closed socket is always "ready": polling it would spin) */
        fds[n].fd = g_peer[i].fd;
        fds[n].events = (short) (POLLIN | (g_peer[i].ohead ? POLLOUT : 0));
        fds[n].revents = 0;
        n++;
    }
    (void) poll(fds, (nfds_t) n, 100);
}

static Out *post_send(int dest, int tag, const void *buf, int bytes)
{
    Peer *p = &g_peer[dest];
    Out *o = (Out *) calloc(1, sizeof(Out));
    int hdr[3] = {tag, bytes, MAGIC};
    o->len = sizeof hdr + (size_t) bytes;
    o->data = (char *) malloc(o->len);
    if (!o->data) die("out of memory");
    memcpy(o->data, hdr, sizeof hdr);
    if (bytes > 0) memcpy(o->data + sizeof hdr, buf, (size_t) bytes);
    if (p->otail) p->otail->next = o; else p->ohead = o;
    p->otail = o;
    flush_peer(p);
    return o;
}

/* take the first queued message from `src` with `tag`; returns 0 if none has arrived yet */
static int match(int src, int tag, void *buf, int maxBytes, int *got)
{
    Peer *p = &g_peer[src];
    Msg *m, *prev = NULL;
    for (m = p->head; m; prev = m, m = m->next) {
        if (m->tag != tag) continue;
        if (m->bytes > maxBytes) die("message longer than the receive buffer");
        if (m->bytes > 0) memcpy(buf, m->data, (size_t) m->bytes);
        *got = m->bytes;
        if (prev) prev->next = m->next; else p->head = m->next;
        if (p->tail == m) p->tail = prev;
        free(m->data);
        free(m);
        return 1;
    }
    return 0;
}

static void send_blocking(int dest, int tag, const void *buf, int bytes)
{
    Out *o;
    if (dest == g_rank) die("send to self is not supported");
    o = post_send(dest, tag, buf, bytes);
    while (o->data != NULL) progress(1);               /* flush_peer clears data once everything is written */
    free(o);
}

static int recv_blocking(int src, int tag, void *buf, int maxBytes)
{
    int got = 0;
    if (src == g_rank) die("receive from self is not supported");
    for (;;) {
        if (match(src, tag, buf, maxBytes, &got)) return got;
        /* everything the peer ever sent is in its queue once its end is closed: no match now = no match ever */
        if (g_peer[src].closed) die_peer(src);
        progress(1);
    }
}

int MPI_Init(int *argc, char ***argv)
{
    const char *r = getenv("MBAMD_MPI_RANK"), *s = getenv("MBAMD_MPI_SIZE"), *f = getenv("MBAMD_MPI_FDS");
    int i;
    (void) argc; (void) argv;
    if (g_init) return MPI_SUCCESS;
    g_init = 1;
    if (!r || !s || !f) {                               /* started without the launcher: a world of one */
        g_rank = 0;
        g_size = 1;
        g_peer = (Peer *) calloc(1, sizeof(Peer));
        return MPI_SUCCESS;
    }
    g_rank = atoi(r);
    g_size = atoi(s);
    if (g_size < 1 || g_size > 256 || g_rank < 0 || g_rank >= g_size) die("bad MBAMD_MPI_RANK / MBAMD_MPI_SIZE");
    g_peer = (Peer *) calloc((size_t) g_size, sizeof(Peer));
    for (i = 0; i < g_size; i++) {
        char *end;
        long fd = strtol(f, &end, 10);
        g_peer[i].fd = (int) fd;
        f = (*end == ',') ? end + 1 : end;
        if (i != g_rank) {
            int fl = fcntl(g_peer[i].fd, F_GETFL, 0);
            if (fl < 0 || fcntl(g_peer[i].fd, F_SETFL, fl | O_NONBLOCK) < 0) die("bad socket in MBAMD_MPI_FDS");
        }
    }
    return MPI_SUCCESS;
}

int MPI_Finalize(void)
{
    int i;
    if (!g_init) return MPI_SUCCESS;
    if (g_size > 1) MPI_Barrier(MPI_COMM_WORLD);        /* nobody closes a socket a peer still reads */
    for (i = 0; i < g_size; i++)
        if (i != g_rank && g_size > 1) close(g_peer[i].fd);
    g_init = 0;
    return MPI_SUCCESS;
}

int MPI_Comm_size(MPI_Comm comm, int *size) { (void) comm; *size = g_size; return MPI_SUCCESS; }
int MPI_Comm_rank(MPI_Comm comm, int *rank) { (void) comm; *rank = g_rank; return MPI_SUCCESS; }

int MPI_Send(const void *buf, int count, MPI_Datatype type, int dest, int tag, MPI_Comm comm)
{
    (void) comm;
    send_blocking(dest, tag, buf, count * elem(type));
    return MPI_SUCCESS;
}

int MPI_Recv(void *buf, int count, MPI_Datatype type, int source, int tag, MPI_Comm comm, MPI_Status *status)
{
    int got;
    (void) comm;
    got = recv_blocking(source, tag, buf, count * elem(type));
    if (status) {
        status->MPI_SOURCE = source;
        status->MPI_TAG = tag;
        status->MPI_ERROR = MPI_SUCCESS;
        status->bytes = got;
    }
    return MPI_SUCCESS;
}

int MPI_Isend(const void *buf, int count, MPI_Datatype type, int dest, int tag, MPI_Comm comm, MPI_Request *request)
{
    struct MbamdMpiRequest *q = (struct MbamdMpiRequest *) calloc(1, sizeof *q);
    (void) comm;
    if (dest == g_rank) die("send to self is not supported");
    q->isRecv = 0;
    q->peer = dest;
    q->tag = tag;
    q->out = post_send(dest, tag, buf, count * elem(type));      /* the payload is copied: the caller's buffer is free at once */
    *request = q;
    return MPI_SUCCESS;
}

int MPI_Irecv(void *buf, int count, MPI_Datatype type, int source, int tag, MPI_Comm comm, MPI_Request *request)
{
    struct MbamdMpiRequest *q = (struct MbamdMpiRequest *) calloc(1, sizeof *q);
    (void) comm;
    if (source == g_rank) die("receive from self is not supported");
    q->isRecv = 1;
    q->peer = source;
    q->tag = tag;
    q->buf = buf;
    q->bytes = count * elem(type);
    *request = q;
    return MPI_SUCCESS;
}

int MPI_Waitall(int count, MPI_Request *requests, MPI_Status *statuses)
{
    int i, left = count;
    for (i = 0; i < count; i++)
        if (requests[i] == NULL) left--;
    while (left > 0) {
        for (i = 0; i < count; i++) {
            struct MbamdMpiRequest *q = requests[i];
            if (!q || q->done) continue;
            if (q->isRecv) {
                if (match(q->peer, q->tag, q->buf, q->bytes, &q->got)) q->done = 1;
                else if (g_peer[q->peer].closed) die_peer(q->peer);
            } else if (q->out->data == NULL) {
                free(q->out);
                q->out = NULL;
                q->done = 1;
            }
            if (q->done) left--;
        }
        if (left > 0) progress(1);
    }
    for (i = 0; i < count; i++) {
        struct MbamdMpiRequest *q = requests[i];
        if (!q) continue;
        if (statuses) {
            statuses[i].MPI_SOURCE = q->peer;
            statuses[i].MPI_TAG = q->tag;
            statuses[i].MPI_ERROR = MPI_SUCCESS;
            statuses[i].bytes = q->got;
        }
        free(q);
        requests[i] = NULL;
    }
    return MPI_SUCCESS;
}

int MPI_Barrier(MPI_Comm comm)
{
    int i, token = 0;
    (void) comm;
    if (g_size == 1) return MPI_SUCCESS;
    if (g_rank == 0) {
        for (i = 1; i < g_size; i++) (void) recv_blocking(i, TAG_BARRIER, &token, sizeof token);
        for (i = 1; i < g_size; i++) send_blocking(i, TAG_BARRIER, &token, sizeof token);
    } else {
        send_blocking(0, TAG_BARRIER, &token, sizeof token);
        (void) recv_blocking(0, TAG_BARRIER, &token, sizeof token);
    }
    return MPI_SUCCESS;
}

int MPI_Bcast(void *buf, int count, MPI_Datatype type, int root, MPI_Comm comm)
{
    int i, bytes = count * elem(type);
    (void) comm;
    if (g_size == 1) return MPI_SUCCESS;
    if (g_rank == root) {
        for (i = 0; i < g_size; i++)
            if (i != root) send_blocking(i, TAG_BCAST, buf, bytes);
    } else {
        (void) recv_blocking(root, TAG_BCAST, buf, bytes);
    }
    return MPI_SUCCESS;
}

static void combine(void *acc, const void *in, int count, MPI_Datatype type, MPI_Op op)
{
    int i;
#define COMBINE(T)                                                              \
    for (i = 0; i < count; i++) {                                               \
        T a = ((T *) acc)[i], b = ((const T *) in)[i];                          \
        ((T *) acc)[i] = op == MPI_SUM ? (T) (a + b) : op == MPI_MAX ? (a > b ? a : b) : (a < b ? a : b); \
    }
    switch (type) {
        case MPI_CHAR: COMBINE(char) break;
        case MPI_INT: COMBINE(int) break;
        case MPI_LONG: COMBINE(long) break;
        case MPI_LONG_LONG: COMBINE(long long) break;
        case MPI_UNSIGNED_LONG: COMBINE(unsigned long) break;
        case MPI_FLOAT: COMBINE(float) break;
        case MPI_DOUBLE: COMBINE(double) break;
        default: die("reduction on an unknown datatype");
    }
#undef COMBINE
}

/* the root adds the contributions in rank order: every run of the same job gives the same bits */
int MPI_Reduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype type, MPI_Op op, int root, MPI_Comm comm)
{
    int i, bytes = count * elem(type);
    (void) comm;
    if (g_rank != root) {
        send_blocking(root, TAG_REDUCE, sendbuf, bytes);
        return MPI_SUCCESS;
    }
    {
        char *tmp = (char *) malloc(bytes > 0 ? (size_t) bytes : 1), *acc = (char *) malloc(bytes > 0 ? (size_t) bytes : 1);
        int first = 1;
        if (!tmp || !acc) die("out of memory");
        for (i = 0; i < g_size; i++) {
            const void *src = sendbuf;
            if (i != root) {
                (void) recv_blocking(i, TAG_REDUCE, tmp, bytes);
                src = tmp;
            }
            if (first) { memcpy(acc, src, (size_t) bytes); first = 0; }
            else combine(acc, src, count, type, op);
        }
        memcpy(recvbuf, acc, (size_t) bytes);
        free(tmp);
        free(acc);
    }
    return MPI_SUCCESS;
}

int MPI_Allreduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype type, MPI_Op op, MPI_Comm comm)
{
    MPI_Reduce(sendbuf, recvbuf, count, type, op, 0, comm);
    return MPI_Bcast(recvbuf, count, type, 0, comm);
}
