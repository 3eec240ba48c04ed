// lt_gather_*: the one exchange step of the multi-GPU path -- an RCCL all-gather of the fixed 64-byte lane records
// (BASELINE north_star: "RCCL over xGMI only for the final gather of fitted coefficients").
// The reference has no counterpart (it is single-process Python).
//
// One communicator per rank (= per process = per GPU).  librccl.so is opened with dlopen at lt_gather_init, so a
// single-GPU user of the library never loads RCCL.  The ncclUniqueId travels through a file: rank 0 creates it and
// publishes it under a temporary name + rename (atomic on one node); the other ranks wait for the file.
//
// Data path: lt_gather_stage() copies records of context slots into the send buffer, stream-ordered behind the slots'
// searches and without a host wait; lt_gather_records() lets the gather stream wait for the context's streams through
// events, runs ONE ncclAllGather (records of all staged steps) and copies the result into page-locked host memory.
// 4096 frames on 8 GPUs are 32 KiB per rank: latency-bound, so xGMI ring/link bandwidth does not matter here.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "lt_internal.h"

using namespace lt;

namespace {

// The six RCCL entry points this file uses, declared here (their ABI is NCCL's and stable) so that building the library
// needs no RCCL headers; the library itself is found at run time.
struct ncclUniqueId { char internal[128]; };
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;                 // ncclSuccess == 0
constexpr ncclResult_t ncclSuccess = 0;
constexpr int ncclUint8 = 1;              // ncclDataType_t
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.handle) return LT_OK;
    std::vector<std::string> names;
    if (const char* e = std::getenv("LT_RCCL_LIB")) names.push_back(e);
    names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"});
    std::string tried;
    for (const auto& n : names) {
        void* h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { tried += n + ": " + (dlerror() ? dlerror() : "?") + "; "; continue; }
        Rccl r;
        r.handle = h;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));
        if (r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy && r.GetErrorString && r.CommCount) {
            g_rccl = r;
            return LT_OK;
        }
        dlclose(h);
        tried += n + ": missing nccl symbols; ";
    }
    return set_error(LT_ERR_STATE, "librccl.so could not be loaded (%s)", tried.c_str());
}

#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) return set_error(LT_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)
#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return set_error(LT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

struct lt_gather {
    lt_ctx* ctx = nullptr;
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> events;     // one per context stream: the gather waits for the staged copies
    int cap = 0;                        // records per rank the buffers hold
    int agreed_records = -1;            // the n_records every rank was last seen to pass to lt_gather_records
    lt_lane_record *d_send = nullptr, *d_recv = nullptr, *h_recv = nullptr;
    unsigned char *d_blob = nullptr, *h_blob = nullptr;   // small host-to-host all-gathers (barrier, timing)
    size_t blob_cap = 0;
};

namespace {

int ensure_blob(lt_gather* g, size_t bytes) {
    if (bytes <= g->blob_cap) return LT_OK;
    if (g->d_blob) (void)hipFree(g->d_blob);
    if (g->h_blob) (void)hipHostFree(g->h_blob);
    g->d_blob = g->h_blob = nullptr;
    g->blob_cap = 0;
    const size_t cap = std::max<size_t>(bytes, 256);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->d_blob), cap * ((size_t)g->world + 1)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&g->h_blob), cap * ((size_t)g->world + 1), hipHostMallocDefault));
    g->blob_cap = cap;
    return LT_OK;
}

}  // namespace

namespace {
// librccl prints a version banner on stdout when its first communicator comes up; a caller that writes its result to
// stdout (bench.py prints ONE JSON line) must not find it there.  While this object lives, fd 1 points at stderr.
struct StdoutToStderr {
    int saved = -1;
    StdoutToStderr() {
        std::fflush(stdout);
        saved = dup(1);
        if (saved >= 0) dup2(2, 1);
    }
    ~StdoutToStderr() {
        std::fflush(stdout);
        if (saved >= 0) {
            dup2(saved, 1);
            close(saved);
        }
    }
};
}  // namespace

extern "C" {

int lt_gather_init(lt_ctx* ctx, int rank, int world, const char* id_path, int timeout_s, lt_gather** out) {
    if (!ctx || !out || !id_path) return set_error(LT_ERR_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return set_error(LT_ERR_INVALID, "rank %d outside world %d", rank, world);
    *out = nullptr;
    StdoutToStderr quiet;
    int rc = load_rccl();
    if (rc) return rc;
    const int device = ctx_device(ctx);
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    std::memset(&id, 0, sizeof id);
    if (rank == 0) {
        NCCL_TRY(g_rccl.GetUniqueId(&id));
        // a fresh file of our own (never through a symlink or onto somebody else's file), then an atomic rename
        const std::string tmp = std::string(id_path) + ".tmp." + std::to_string((long)getpid());
        (void)unlink(tmp.c_str());
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
        if (fd < 0) return set_error(LT_ERR_STATE, "cannot create %s", tmp.c_str());
        const ssize_t w = write(fd, &id, sizeof id);
        close(fd);
        if (w != (ssize_t)sizeof id || std::rename(tmp.c_str(), id_path) != 0) {
            std::remove(tmp.c_str());
            return set_error(LT_ERR_STATE, "cannot publish the RCCL id at %s", id_path);
        }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            struct stat st;
            if (stat(id_path, &st) == 0 && (size_t)st.st_size == sizeof id) {
                const int fd = open(id_path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
                if (fd >= 0) {
                    const ssize_t r = read(fd, &id, sizeof id);
                    close(fd);
                    if (r == (ssize_t)sizeof id) break;
                }
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > (timeout_s > 0 ? timeout_s : 120))
                return set_error(LT_ERR_STATE, "rank %d: no RCCL id at %s after %d s (is rank 0 running?)", rank, id_path,
                                 timeout_s > 0 ? timeout_s : 120);
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    lt_gather* g = new lt_gather;
    g->ctx = ctx;
    g->rank = rank;
    g->world = world;
    g->device = device;
    ncclResult_t nr = g_rccl.CommInitRank(&g->comm, world, id, rank);
    if (nr != ncclSuccess) {
        delete g;
        return set_error(LT_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(nr));
    }
    int count = 0;
    if (g_rccl.CommCount(g->comm, &count) != ncclSuccess || count != world) {
        lt_gather_destroy(g);
        return set_error(LT_ERR_STATE, "RCCL communicator has %d ranks, expected %d", count, world);
    }
    if (stream_get(&g->stream, SK_PLAIN, 0) != hipSuccess) {
        lt_gather_destroy(g);
        return set_error(LT_ERR_HIP, "hipStreamCreate failed");
    }
    *out = g;
    return LT_OK;
}

int lt_gather_world(lt_gather* g, int* rank, int* world) {
    if (!g) return set_error(LT_ERR_INVALID, "null gather");
    if (rank) *rank = g->rank;
    if (world) *world = g->world;
    return LT_OK;
}

int lt_gather_reserve(lt_gather* g, int records_per_rank) {
    if (!g || records_per_rank < 0) return set_error(LT_ERR_INVALID, "bad argument");
    if (records_per_rank <= g->cap) return LT_OK;
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(hipStreamSynchronize(g->stream));
    int rc = ctx_sync(g->ctx);
    if (rc) return rc;
    if (g->d_send) (void)hipFree(g->d_send);
    if (g->d_recv) (void)hipFree(g->d_recv);
    if (g->h_recv) (void)hipHostFree(g->h_recv);
    g->d_send = g->d_recv = g->h_recv = nullptr;
    g->cap = 0;
    const size_t n = (size_t)records_per_rank;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->d_send), n * sizeof(lt_lane_record)));
    HIP_TRY(hipMemset(g->d_send, 0, n * sizeof(lt_lane_record)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g->d_recv), n * g->world * sizeof(lt_lane_record)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&g->h_recv), n * g->world * sizeof(lt_lane_record), hipHostMallocDefault));
    g->cap = records_per_rank;
    g->agreed_records = -1;
    return LT_OK;
}

int lt_gather_stage(lt_gather* g, int first_slot, int n, int at) {
    if (!g) return set_error(LT_ERR_INVALID, "null gather");
    if (n < 0 || at < 0 || at + n > g->cap)
        return set_error(LT_ERR_CAPACITY, "records [%d, %d) outside the gather's capacity %d (lt_gather_reserve)", at, at + n, g->cap);
    return ctx_enqueue_records(g->ctx, first_slot, n, g->d_send + at);
}

int lt_gather_records(lt_gather* g, int n_records, lt_lane_record* out_host) {
    if (!g || !out_host) return set_error(LT_ERR_INVALID, "null argument");
    if (n_records < 0 || n_records > g->cap) return set_error(LT_ERR_CAPACITY, "%d records exceed the gather's capacity %d", n_records, g->cap);
    // Every rank must pass the same count and hold the same capacity: a mismatch would hang or misplace records inside
    // the collective, so it is checked with one small all-gather on EVERY call (also for 0, which is a collective no-op
    // only if it is 0 everywhere).  Unconditional: a check that ran only when this rank's count changed would itself be a
    // mismatched collective when one rank's count changes and another's does not.  8 bytes per rank, latency only.
    {
        const int32_t mine[2] = {n_records, g->cap};
        std::vector<int32_t> all(2 * (size_t)g->world);
        int rc = lt_gather_host(g, mine, sizeof mine, all.data());
        if (rc) return rc;
        for (int r = 0; r < g->world; ++r)
            if (all[2 * (size_t)r] != n_records || all[2 * (size_t)r + 1] != g->cap) {
                g->agreed_records = -1;
                return set_error(LT_ERR_STATE, "lt_gather_records: rank %d passes %d records (capacity %d), rank %d passes %d (capacity %d)",
                                 g->rank, n_records, g->cap, r, all[2 * (size_t)r], all[2 * (size_t)r + 1]);
            }
        g->agreed_records = n_records;
    }
    if (n_records == 0) return LT_OK;
    HIP_TRY(hipSetDevice(g->device));
    // the gather stream waits for whatever the context's streams have enqueued so far (the staged copies)
    hipStream_t streams[16];
    const int ns = ctx_streams(g->ctx, streams, 16);
    while ((int)g->events.size() < ns) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        g->events.push_back(e);
    }
    for (int i = 0; i < ns; ++i) {
        HIP_TRY(hipEventRecord(g->events[(size_t)i], streams[i]));
        HIP_TRY(hipStreamWaitEvent(g->stream, g->events[(size_t)i], 0));
    }
    const size_t bytes = (size_t)n_records * sizeof(lt_lane_record);
    NCCL_TRY(g_rccl.AllGather(g->d_send, g->d_recv, bytes, ncclUint8, g->comm, g->stream));
    HIP_TRY(hipMemcpyAsync(g->h_recv, g->d_recv, bytes * g->world, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    std::memcpy(out_host, g->h_recv, bytes * g->world);
    return LT_OK;
}

int lt_gather_host(lt_gather* g, const void* in, size_t bytes, void* out) {
    if (!g || (bytes && (!in || !out))) return set_error(LT_ERR_INVALID, "null argument");
    if (bytes == 0) return LT_OK;
    HIP_TRY(hipSetDevice(g->device));
    int rc = ensure_blob(g, bytes);
    if (rc) return rc;
    unsigned char* d_in = g->d_blob + g->blob_cap * (size_t)g->world;
    unsigned char* h_in = g->h_blob + g->blob_cap * (size_t)g->world;
    std::memcpy(h_in, in, bytes);
    HIP_TRY(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, g->stream));
    NCCL_TRY(g_rccl.AllGather(d_in, g->d_blob, bytes, ncclUint8, g->comm, g->stream));
    HIP_TRY(hipMemcpyAsync(g->h_blob, g->d_blob, bytes * g->world, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    std::memcpy(out, g->h_blob, bytes * g->world);
    return LT_OK;
}

int lt_gather_barrier(lt_gather* g) {
    if (!g) return set_error(LT_ERR_INVALID, "null gather");
    unsigned char in[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    std::vector<unsigned char> out(8 * (size_t)g->world);
    int rc = ctx_sync(g->ctx);
    if (rc) return rc;
    return lt_gather_host(g, in, sizeof in, out.data());
}

void lt_gather_destroy(lt_gather* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g->comm);
    for (auto e : g->events) (void)hipEventDestroy(e);
    stream_put(g->stream);       // (no stream of the library is ever destroyed: lt_api.cpp, StreamPool)
    if (g->d_send) (void)hipFree(g->d_send);
    if (g->d_recv) (void)hipFree(g->d_recv);
    if (g->h_recv) (void)hipHostFree(g->h_recv);
    if (g->d_blob) (void)hipFree(g->d_blob);
    if (g->h_blob) (void)hipHostFree(g->h_blob);
    delete g;
}

}  // extern "C"
