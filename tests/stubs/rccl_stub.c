/*
 * TEST INFRASTRUCTURE: a stand-in for librccl.so that moves HOST memory between the processes of one machine, so that the
 * multi-rank glue of the C-ABI (gpd_comm_unique_id / gpd_comm_init / gpd_comm_count / gpd_allgather_obs / gpd_comm_destroy,
 * include/gpd.h) can run with world_size > 1 where there is neither a GPU nor RCCL (libgpd.so resolves RCCL with dlopen;
 * GPD_RCCL_LIB points it here).  Implements exactly the entry points libgpd.so binds (the collective ones and the grouped
 * point-to-point ones), with RCCL's signatures; the
 * communicator is a POSIX shared-memory segment named by the unique id, holding a process-shared barrier and one slot per
 * rank.  Nothing here is part of the product.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
enum { ncclSuccess = 0, ncclInvalidArgument = 4, ncclSystemError = 2 };

#define SLOT_BYTES (4u << 20)
typedef struct {
    pthread_barrier_t barrier;
    int nranks;
    int ready;
    char data[];
} Segment;
typedef struct { int rank, nranks; Segment* seg; size_t bytes; char name[64]; } Comm;
typedef Comm* ncclComm_t;

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof(id->internal), "/gpd_rccl_stub_%d_%ld", (int)getpid(), (long)ts.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm* c = calloc(1, sizeof(Comm));
    c->rank = rank; c->nranks = nranks;
    c->bytes = sizeof(Segment) + (size_t)nranks * SLOT_BYTES;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return ncclSystemError;
    } else {
        for (int tries = 0; tries < 20000 && fd < 0; ++tries) { fd = shm_open(c->name, O_RDWR, 0600); if (fd < 0) usleep(500); }
        if (fd < 0) return ncclSystemError;
    }
    c->seg = mmap(NULL, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->seg == MAP_FAILED) return ncclSystemError;
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->seg->barrier, &a, (unsigned)nranks);
        c->seg->nranks = nranks;
        __sync_synchronize();
        c->seg->ready = 1;
    } else {
        for (int tries = 0; tries < 20000 && !((volatile Segment*)c->seg)->ready; ++tries) usleep(500);
        if (!c->seg->ready || c->seg->nranks != nranks) return ncclInvalidArgument;
    }
    pthread_barrier_wait(&c->seg->barrier);              /* (collective, like the real one) */
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { if (!comm || !count) return ncclInvalidArgument; *count = comm->nranks; return ncclSuccess; }

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, void* stream) {
    (void)stream;
    if (!comm || !sendbuff || !recvbuff || datatype != 7 /* ncclFloat32 */) return ncclInvalidArgument;
    const size_t bytes = sendcount * 4;
    if (bytes > SLOT_BYTES) return ncclInvalidArgument;
    memcpy(comm->seg->data + (size_t)comm->rank * SLOT_BYTES, sendbuff, bytes);      /* (before the barrier: in-place calls alias) */
    pthread_barrier_wait(&comm->seg->barrier);
    for (int r = 0; r < comm->nranks; ++r) memcpy((char*)recvbuff + (size_t)r * bytes, comm->seg->data + (size_t)r * SLOT_BYTES, bytes);
    pthread_barrier_wait(&comm->seg->barrier);
    return ncclSuccess;
}

/* ---- grouped point-to-point: operations are queued between ncclGroupStart and ncclGroupEnd; the group end writes this rank's
 * sends into its slot as [peer, count, payload] records, meets the others at the barrier, and picks the k-th record addressed to
 * it out of every peer's slot for its k-th receive from that peer (RCCL's matching rule: order per pair). ---- */
typedef struct { int send, peer; void* ptr; size_t count; Comm* comm; } Op;
static __thread Op g_ops[256];
static __thread int g_nops = 0, g_depth = 0;

ncclResult_t ncclGroupStart(void) { if (g_depth++ == 0) g_nops = 0; return ncclSuccess; }

static ncclResult_t queue_op(int send, void* ptr, size_t count, ncclDataType_t dt, int peer, Comm* comm) {
    if (!comm || !ptr || dt != 7 || peer < 0 || peer >= comm->nranks || g_depth == 0 || g_nops >= 256) return ncclInvalidArgument;
    g_ops[g_nops++] = (Op){send, peer, ptr, count, comm};
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, void* stream) {
    (void)stream; return queue_op(1, (void*)buf, count, dt, peer, comm);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, void* stream) {
    (void)stream; return queue_op(0, buf, count, dt, peer, comm);
}

ncclResult_t ncclGroupEnd(void) {
    if (g_depth == 0) return ncclInvalidArgument;
    if (--g_depth > 0 || g_nops == 0) return ncclSuccess;
    Comm* c = g_ops[0].comm;
    char* mine = c->seg->data + (size_t)c->rank * SLOT_BYTES;
    size_t off = 0;
    for (int i = 0; i < g_nops; ++i) {
        if (!g_ops[i].send) continue;
        const size_t bytes = g_ops[i].count * 4;
        if (off + 16 + bytes + 16 > SLOT_BYTES) return ncclInvalidArgument;
        int64_t hdr[2] = {g_ops[i].peer, (int64_t)g_ops[i].count};
        memcpy(mine + off, hdr, 16); memcpy(mine + off + 16, g_ops[i].ptr, bytes);
        off += 16 + bytes;
    }
    int64_t end[2] = {-1, 0};
    memcpy(mine + off, end, 16);
    pthread_barrier_wait(&c->seg->barrier);
    ncclResult_t rc = ncclSuccess;
    for (int i = 0; i < g_nops; ++i) {
        if (g_ops[i].send) continue;
        int kth = 0;                                         /* this is my kth receive from that peer */
        for (int j = 0; j < i; ++j) if (!g_ops[j].send && g_ops[j].peer == g_ops[i].peer) ++kth;
        const char* p = c->seg->data + (size_t)g_ops[i].peer * SLOT_BYTES;
        int found = 0;
        for (size_t o = 0;;) {
            int64_t hdr[2];
            memcpy(hdr, p + o, 16);
            if (hdr[0] < 0) break;
            if (hdr[0] == c->rank && kth-- == 0) {
                if ((size_t)hdr[1] != g_ops[i].count) rc = ncclInvalidArgument;      /* (sizes of a matched pair must agree) */
                else memcpy(g_ops[i].ptr, p + o + 16, g_ops[i].count * 4);
                found = 1;
                break;
            }
            o += 16 + (size_t)hdr[1] * 4;
        }
        if (!found) rc = ncclInvalidArgument;
    }
    pthread_barrier_wait(&c->seg->barrier);
    g_nops = 0;
    return rc;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    pthread_barrier_wait(&comm->seg->barrier);
    munmap(comm->seg, comm->bytes);
    if (comm->rank == 0) shm_unlink(comm->name);
    free(comm);
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "invalid argument (stub)" : "system error (stub)"; }
