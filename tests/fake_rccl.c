/*
 * fake_rccl.c -- TEST-ONLY stand-in for librccl.so, selected with LT_RCCL_LIB (lt_gather.cpp::load_rccl).
 *
 * Purpose: the GPU box of the test tier has ONE GPU, and the real RCCL refuses two ranks on one device.  This
 * library implements the six nccl entry points lt_gather.cpp binds, for ranks that are processes of one node, over
 * a POSIX shared-memory segment: ncclAllGather = hipMemcpy D2H of the rank's block into its slot, a process-shared
 * barrier, hipMemcpy H2D of every slot into the receive buffer.  With it two rank PROCESSES on GPU 0 drive the real
 * N > 1 code of lt_gather.cpp (id-file publish / wait loop, CommInitRank with world > 1, rank-major receive layout,
 * count agreement check, lt_gather_host, lt_gather_barrier).  It says nothing about xGMI performance and is never
 * used by the product or by bench.py's reported numbers.
 *
 * Build (tests/fake_rccl.py does it): gcc -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
 *                                      fake_rccl.c -L/opt/rocm/lib -lamdhip64 -lrt -o libfake_rccl.so
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define FAKE_MAX_RANKS 16
#define FAKE_SLOT_BYTES ((size_t)8 << 20) /* per rank and exchange round; larger calls go in rounds */
#define FAKE_MAGIC 0x6c74666bu            /* "ltfk" */
#define FAKE_TIMEOUT_S 120.0

typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 };

typedef struct {
    _Atomic uint32_t magic;
    _Atomic int arrived, generation, failed;
    int nranks;
} seg_header;

struct ncclComm {
    int rank, nranks;
    seg_header* hdr;
    unsigned char* slots; /* nranks * FAKE_SLOT_BYTES */
    size_t map_bytes;
    unsigned long gathers; /* for the test that wants to know this library really carried the traffic */
};
typedef struct ncclComm* ncclComm_t;

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void nap(void) {
    struct timespec t = {0, 50000};
    nanosleep(&t, NULL);
}

/* process-shared generation barrier; fails (instead of hanging the GPU box) when a peer died or never arrives */
static ncclResult_t seg_barrier(struct ncclComm* c) {
    seg_header* h = c->hdr;
    const int gen = atomic_load(&h->generation);
    if (atomic_fetch_add(&h->arrived, 1) + 1 == c->nranks) {
        atomic_store(&h->arrived, 0);
        atomic_fetch_add(&h->generation, 1);
        return ncclSuccess;
    }
    const double t0 = now_s();
    int spins = 0;
    while (atomic_load(&h->generation) == gen) {
        if (atomic_load(&h->failed)) return ncclSystemError;
        if (++spins > 2000) nap();
        if (now_s() - t0 > FAKE_TIMEOUT_S) {
            atomic_store(&h->failed, 1);
            return ncclSystemError;
        }
    }
    return ncclSuccess;
}

static size_t seg_bytes(int nranks) { return 4096 + (size_t)nranks * FAKE_SLOT_BYTES; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "/lt_fake_rccl_%ld_%lld%09ld", (long)getpid(), (long long)t.tv_sec, t.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > FAKE_MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    if (id.internal[0] != '/') return ncclInvalidArgument;
    const size_t bytes = seg_bytes(nranks);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(id.internal, O_RDWR | O_CREAT | O_EXCL, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
            if (fd >= 0) close(fd);
            return ncclSystemError;
        }
    } else {
        const double t0 = now_s();
        struct stat st;
        for (;;) { /* rank 0 creates and sizes the segment; wait for both */
            fd = shm_open(id.internal, O_RDWR, 0600);
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == bytes) break;
            if (fd >= 0) close(fd);
            if (now_s() - t0 > FAKE_TIMEOUT_S) return ncclSystemError;
            nap();
        }
    }
    void* p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    struct ncclComm* c = (struct ncclComm*)calloc(1, sizeof *c);
    c->rank = rank;
    c->nranks = nranks;
    c->hdr = (seg_header*)p;
    c->slots = (unsigned char*)p + 4096;
    c->map_bytes = bytes;
    if (rank == 0) {
        c->hdr->nranks = nranks;
        atomic_store(&c->hdr->magic, FAKE_MAGIC); /* a fresh shm segment is zero-filled: counters start at 0 */
    } else {
        const double t0 = now_s();
        while (atomic_load(&c->hdr->magic) != FAKE_MAGIC) {
            if (now_s() - t0 > FAKE_TIMEOUT_S) { munmap(p, bytes); free(c); return ncclSystemError; }
            nap();
        }
        if (c->hdr->nranks != nranks) { munmap(p, bytes); free(c); return ncclInvalidArgument; }
    }
    ncclResult_t r = seg_barrier(c);     /* everybody has mapped the segment ... */
    if (rank == 0) shm_unlink(id.internal); /* ... so the name can go: nothing is left behind if a rank dies later */
    if (r != ncclSuccess) { munmap(p, bytes); free(c); return r; }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->nranks;
    return ncclSuccess;
}

static size_t dtype_bytes(int dt) {
    switch (dt) {
        case 0: case 1: return 1;           /* ncclInt8 / ncclUint8 */
        case 2: case 3: case 7: return 4;   /* ncclInt32 / ncclUint32 / ncclFloat32 */
        case 4: case 5: case 8: return 8;   /* ncclInt64 / ncclUint64 / ncclFloat64 */
        case 6: case 9: return 2;           /* ncclFloat16 / ncclBfloat16 */
        default: return 0;
    }
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, ncclComm_t comm,
                           hipStream_t stream) {
    if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
    const size_t unit = dtype_bytes(datatype);
    if (!unit) return ncclInvalidArgument;
    const size_t bytes = sendcount * unit;
    /* stream semantics: everything enqueued before the collective is done before it reads, and the result is
     * there before anything enqueued after it runs -- trivially true for a call that synchronises */
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    for (size_t off = 0; off < bytes || off == 0; off += FAKE_SLOT_BYTES) {
        const size_t n = bytes - off < FAKE_SLOT_BYTES ? bytes - off : FAKE_SLOT_BYTES;
        if (n && hipMemcpy(comm->slots + (size_t)comm->rank * FAKE_SLOT_BYTES, (const char*)sendbuff + off, n,
                           hipMemcpyDeviceToHost) != hipSuccess)
            return ncclUnhandledCudaError;
        ncclResult_t r = seg_barrier(comm); /* every slot is written */
        if (r != ncclSuccess) return r;
        for (int p = 0; p < comm->nranks && n; ++p)
            if (hipMemcpy((char*)recvbuff + (size_t)p * bytes + off, comm->slots + (size_t)p * FAKE_SLOT_BYTES, n,
                          hipMemcpyHostToDevice) != hipSuccess)
                return ncclUnhandledCudaError;
        r = seg_barrier(comm);              /* every slot is read: the next round may overwrite */
        if (r != ncclSuccess) return r;
        if (bytes == 0) break;
    }
    comm->gathers++;
    if (getenv("LT_FAKE_RCCL_TRACE"))
        fprintf(stderr, "[fake_rccl] rank %d/%d all-gather #%lu of %zu bytes\n", comm->rank, comm->nranks, comm->gathers, bytes);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    munmap((void*)comm->hdr, comm->map_bytes);
    free(comm);
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (fake_rccl)";
        case ncclUnhandledCudaError: return "HIP call failed (fake_rccl)";
        case ncclSystemError: return "shared-memory segment / peer timeout (fake_rccl)";
        case ncclInvalidArgument: return "invalid argument (fake_rccl)";
        default: return "internal error (fake_rccl)";
    }
}
