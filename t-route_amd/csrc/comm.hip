// comm.hip -- the multi-GPU plumbing of the engine behind the C ABI (include/trmc.h, "communicator" section):
//   * a communicator over the ranks of ONE node, one process (or thread) per rank, with two transports:
//       RCCL  (librccl.so, loaded with dlopen when the first communicator is made -- the single-GPU path never needs
//              it): ncclAllGather over xGMI, asynchronous on the caller's HIP stream;
//       SHM   a POSIX shared-memory segment: every rank stages its block through host memory.  For ranks that share a
//              device (a rehearsal of an N-rank job on a one-GPU box -- RCCL refuses two ranks on one device) and for
//              hosts without a GPU (host-pointer collectives only); synchronous;
//   * the few HIP runtime objects the host side needs to order a hand-off against a plan's stream without any other
//     GPU library in the process: device buffers, streams, events, an indexed row gather.
// What crosses ranks on the routing path is what the reference hands from one sub-network order to the next as
// flowveldepth_interorder (compute.py:882-897, consumed mc_reach.pyx:458-469) -- hydrographs of cut rows -- and at the
// end the outlet hydrographs; there is no other exchange.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

// RCCL: types and prototypes only -- the library itself is dlopen'ed, and a host without its headers still builds the
// single-GPU library: the handful of declarations the transport needs are then restated here (rccl.h, NCCL's public API)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm *ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream);
const char *ncclGetErrorString(ncclResult_t result);
}
#endif

#include "../../include/trmc.h"
#include "internal.hpp"

namespace {

using trmc::fail_with;

#define COMM_HIP_TRY(expr)                                                                                  \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail_with(e_ == hipErrorOutOfMemory ? TRMC_ENOMEM : TRMC_EHIP,                           \
                             std::string(#expr) + ": " + hipGetErrorString(e_));                            \
    } while (0)

// ---- RCCL, resolved at run time ------------------------------------------------------------------------------------
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.error = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}
int rccl_fail(const char *what, ncclResult_t rc)
{
    Rccl &r = rccl();
    return fail_with(TRMC_EHIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}

// ---- the shared-memory transport -----------------------------------------------------------------------------------
// A segment is only ever used by the launch that CREATED it: rank 0 removes whatever an earlier job left under the name
// (a crashed or killed job leaves its segment behind, with `attached` >= world and barrier counters mid-count -- trusting
// those let ranks skip the attach wait and barriers release early: silently wrong data) and creates the name exclusively;
// every other rank proves to itself that the segment it has mapped has a LIVE rank 0 of this launch behind it -- it
// writes a fresh random token into its `hello` slot and proceeds only once rank 0 has echoed that token into its `ack`
// slot -- and while it waits it keeps checking that the name still leads to the inode it mapped (if rank 0 has replaced
// the segment since, it starts over on the new one).  A dead segment never answers.
constexpr int kShmMaxWorld = 128;
constexpr uint32_t kShmMagic = 0x74726d63u; // "trmc"
struct ShmHeader {
    std::atomic<uint32_t> arrived;    // ranks inside the current barrier
    std::atomic<uint32_t> generation; // barriers completed
    std::atomic<uint32_t> attached;   // ranks that have completed the handshake and not yet left
    uint32_t world;
    std::atomic<uint32_t> magic;      // set by rank 0 when the header is ready
    uint32_t pad_[3];
    std::atomic<uint64_t> hello[kShmMaxWorld]; // rank r's token of this launch
    std::atomic<uint64_t> ack[kShmMaxWorld];   // rank 0's echo of it
};
constexpr size_t kShmHeader = 4096;
static_assert(sizeof(ShmHeader) <= kShmHeader, "the header must fit its page");

} // namespace

struct trmc_comm {
    int rank = 0, world = 1, device = -1;
    bool use_rccl = false;
    ncclComm_t nccl = nullptr;
    // shm
    std::string shm_name;
    int fd = -1;
    uint8_t *base = nullptr;
    size_t bytes = 0, capacity = 0; // mapped size; data area
    double timeout_s = 120.0;
    // staging for host-pointer collectives over RCCL
    void *stage = nullptr;
    size_t stage_bytes = 0;
    hipStream_t stage_stream = nullptr;
};

namespace {

ShmHeader *header(trmc_comm *c) { return reinterpret_cast<ShmHeader *>(c->base); }
uint8_t *area(trmc_comm *c) { return c->base + kShmHeader; }

int shm_barrier(trmc_comm *c)
{
    ShmHeader *h = header(c);
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.fetch_add(1, std::memory_order_acq_rel);
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while (h->generation.load(std::memory_order_acquire) == gen) {
        if (++spins < 2000)
            continue;
        sched_yield();
        if ((spins & 1023u) == 0
            && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s)
            return fail_with(TRMC_ESTATE, "communicator: a rank did not reach the barrier within "
                                              + std::to_string((int)c->timeout_s) + " s (rank " + std::to_string(c->rank) + " of "
                                              + std::to_string(c->world) + " waiting)");
    }
    return 0;
}

int stage_ensure(trmc_comm *c, size_t need)
{
    if (need <= c->stage_bytes) return 0;
    if (c->stage) (void)hipFree(c->stage);
    c->stage = nullptr;
    c->stage_bytes = 0;
    COMM_HIP_TRY(hipMalloc(&c->stage, need));
    c->stage_bytes = need;
    return 0;
}

int use_device(const trmc_comm *c)
{
    if (c->device >= 0) COMM_HIP_TRY(hipSetDevice(c->device));
    return 0;
}

// out[i][0..row_words) = src[index[i]][0..row_words), 4-byte words
__global__ void __launch_bounds__(256)
k_gather_rows_indexed(const uint32_t *__restrict__ src, const int64_t *__restrict__ index, uint32_t *__restrict__ dst,
                      int64_t nrows, int64_t row_words)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrows * row_words) return;
    const int64_t r = i / row_words, j = i - r * row_words;
    dst[i] = src[index[r] * row_words + j];
}

} // namespace

extern "C" {

int trmc_comm_unique_id(void *id_out)
{
    if (!id_out) return fail_with(TRMC_EINVAL, "id_out is NULL");
    Rccl &r = rccl();
    if (!r.error.empty() || !r.GetUniqueId) return fail_with(TRMC_ENODEVICE, r.error.empty() ? "RCCL unavailable" : r.error);
    static_assert(TRMC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "trmc.h promises RCCL's id size");
    ncclUniqueId id;
    const ncclResult_t rc = r.GetUniqueId(&id);
    if (rc != ncclSuccess) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id_out, &id, sizeof id);
    return 0;
}

int trmc_comm_init(int rank, int world, const void *id, int device, trmc_comm **out)
{
    if (!out) return fail_with(TRMC_EINVAL, "out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail_with(TRMC_EINVAL, "rank/world out of range");
    if (!id) return fail_with(TRMC_EINVAL, "id is NULL");
    Rccl &r = rccl();
    if (!r.error.empty() || !r.CommInitRank) return fail_with(TRMC_ENODEVICE, r.error.empty() ? "RCCL unavailable" : r.error);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail_with(TRMC_ENODEVICE, "no HIP device available");
    if (device < 0 || device >= count) return fail_with(TRMC_EINVAL, "device ordinal out of range");
    COMM_HIP_TRY(hipSetDevice(device));
    trmc_comm *c = new (std::nothrow) trmc_comm();
    if (!c) return fail_with(TRMC_ENOMEM, "out of host memory");
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->use_rccl = true;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    const ncclResult_t rc = r.CommInitRank(&c->nccl, world, uid, rank);
    if (rc != ncclSuccess) {
        delete c;
        return rccl_fail("ncclCommInitRank", rc);
    }
    *out = c;
    return 0;
}

int trmc_comm_init_shm(int rank, int world, const char *name, int device, int64_t capacity_bytes, trmc_comm **out)
{
    if (!out) return fail_with(TRMC_EINVAL, "out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail_with(TRMC_EINVAL, "rank/world out of range");
    if (!name || name[0] != '/' || std::strlen(name) > 200) return fail_with(TRMC_EINVAL, "name must look like \"/something\"");
    if (device >= 0) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail_with(TRMC_ENODEVICE, "no HIP device available");
        if (device >= count) return fail_with(TRMC_EINVAL, "device ordinal out of range");
    }
    trmc_comm *c = new (std::nothrow) trmc_comm();
    if (!c) return fail_with(TRMC_ENOMEM, "out of host memory");
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->shm_name = name;
    c->capacity = capacity_bytes > 0 ? (size_t)capacity_bytes : (size_t)64 << 20;
    c->capacity = (c->capacity + 4095) / 4096 * 4096;
    c->bytes = kShmHeader + c->capacity;
    if (const char *t = std::getenv("TRMC_COMM_TIMEOUT_S")) c->timeout_s = std::max(1.0, std::atof(t));
    auto bail = [&](const std::string &msg) {
        if (c->base) munmap(c->base, c->bytes);
        if (c->fd >= 0) close(c->fd);
        delete c;
        return fail_with(TRMC_EHIP, msg);
    };
    if (world > kShmMaxWorld) return bail("the shared-memory transport serves at most " + std::to_string(kShmMaxWorld) + " ranks");
    const auto t0 = std::chrono::steady_clock::now();
    auto timed_out = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s; };
    auto unmap = [&] {
        if (c->base) munmap(c->base, c->bytes);
        c->base = nullptr;
        if (c->fd >= 0) close(c->fd);
        c->fd = -1;
    };
    if (rank == 0) {
        // whatever is there under this name belongs to an earlier launch: remove it, then create the name exclusively
        for (int attempt = 0; c->fd < 0; ++attempt) {
            (void)shm_unlink(name);
            c->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (c->fd < 0 && (errno != EEXIST || attempt >= 8)) return bail(std::string("shm_open(") + name + "): " + std::strerror(errno));
        }
        if (ftruncate(c->fd, (off_t)c->bytes) != 0) {
            (void)shm_unlink(name);
            return bail(std::string("ftruncate: ") + std::strerror(errno));
        }
        void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
        if (p == MAP_FAILED) {
            (void)shm_unlink(name);
            return bail(std::string("mmap: ") + std::strerror(errno));
        }
        c->base = (uint8_t *)p;
        ShmHeader *h = header(c); // (a fresh segment is zero-filled: counters, tokens and echoes all start at 0)
        h->world = (uint32_t)world;
        h->magic.store(kShmMagic, std::memory_order_release);
        // answer every rank's token until all of them have been answered
        for (;;) {
            int answered = 1;
            for (int r = 1; r < world; ++r) {
                const uint64_t t = h->hello[r].load(std::memory_order_acquire);
                if (t != 0 && h->ack[r].load(std::memory_order_relaxed) != t) h->ack[r].store(t, std::memory_order_release);
                answered += t != 0;
            }
            if (answered == world) break;
            sched_yield();
            if (timed_out()) {
                (void)shm_unlink(name);
                return bail("communicator: not every rank attached to " + c->shm_name + " within the time-out");
            }
        }
    } else {
        uint64_t token = 0;
        {
            FILE *f = std::fopen("/dev/urandom", "rb");
            if (f) {
                if (std::fread(&token, sizeof token, 1, f) != 1) token = 0;
                std::fclose(f);
            }
            if (token == 0)
                token = ((uint64_t)getpid() << 32) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ (uint64_t)(uintptr_t)c;
            token |= 1; // never 0
        }
        for (bool done = false; !done;) {
            unmap();
            if (timed_out()) return bail("communicator: " + c->shm_name + " did not appear (or its rank 0 never answered) within the time-out");
            c->fd = shm_open(name, O_RDWR, 0600);
            struct stat st_fd;
            if (c->fd < 0 || fstat(c->fd, &st_fd) != 0 || (size_t)st_fd.st_size < c->bytes) { // not there yet, or not sized yet
                sched_yield();
                continue;
            }
            void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
            if (p == MAP_FAILED) return bail(std::string("mmap: ") + std::strerror(errno));
            c->base = (uint8_t *)p;
            ShmHeader *h = header(c);
            bool said_hello = false;
            uint32_t spins = 0;
            for (;;) {
                if (!said_hello && h->magic.load(std::memory_order_acquire) == kShmMagic && h->world == (uint32_t)world) {
                    h->hello[rank].store(token, std::memory_order_release);
                    said_hello = true;
                }
                if (said_hello && h->ack[rank].load(std::memory_order_acquire) == token) {
                    done = true;
                    break;
                }
                sched_yield();
                if ((++spins & 255u) == 0) {
                    // does the name still lead to the segment that is mapped here?  (rank 0 of THIS launch replaces a stale one)
                    const int fd2 = shm_open(name, O_RDWR, 0600);
                    struct stat st2;
                    const bool same = fd2 >= 0 && fstat(fd2, &st2) == 0 && st2.st_ino == st_fd.st_ino && st2.st_dev == st_fd.st_dev;
                    if (fd2 >= 0) close(fd2);
                    if (!same || timed_out()) break; // start over on whatever the name leads to now (or give up above)
                }
            }
        }
    }
    header(c)->attached.fetch_add(1, std::memory_order_acq_rel);
    *out = c;
    return 0;
}

int trmc_comm_info(const trmc_comm *c, int32_t *rank, int32_t *world, int32_t *is_rccl)
{
    if (!c) return fail_with(TRMC_EINVAL, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (is_rccl) *is_rccl = c->use_rccl ? 1 : 0;
    return 0;
}

int trmc_comm_barrier(trmc_comm *c)
{
    if (!c) return fail_with(TRMC_EINVAL, "comm is NULL");
    if (c->world == 1) return 0;
    if (!c->use_rccl) return shm_barrier(c);
    uint8_t mine = 1;
    uint8_t all[1024];
    if (c->world > 1024) return fail_with(TRMC_EINVAL, "world too large");
    return trmc_comm_all_gather_host(c, &mine, all, 1);
}

int trmc_comm_all_gather(trmc_comm *c, const void *send_dev, void *recv_dev, int64_t bytes, void *stream)
{
    if (!c) return fail_with(TRMC_EINVAL, "comm is NULL");
    if (bytes < 0) return fail_with(TRMC_EINVAL, "bytes < 0");
    if (bytes == 0) return 0;
    if (!send_dev || !recv_dev) return fail_with(TRMC_EINVAL, "send/recv is NULL");
    if (c->device < 0) return fail_with(TRMC_ESTATE, "this communicator has no device (host-pointer collectives only)");
    if (int rc = use_device(c)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (c->use_rccl) {
        const ncclResult_t rc = rccl().AllGather(send_dev, recv_dev, (size_t)bytes, ncclChar, c->nccl, st);
        if (rc != ncclSuccess) return rccl_fail("ncclAllGather", rc);
        return 0;
    }
    // through host memory, a slice of every rank's block at a time when the blocks exceed the segment
    const size_t slice_max = c->capacity / (size_t)c->world / 256 * 256;
    if (slice_max == 0) return fail_with(TRMC_EINVAL, "shared segment too small for this world size");
    for (size_t off = 0; off < (size_t)bytes; off += slice_max) {
        const size_t n = std::min(slice_max, (size_t)bytes - off);
        COMM_HIP_TRY(hipMemcpyAsync(area(c) + (size_t)c->rank * slice_max, (const uint8_t *)send_dev + off, n, hipMemcpyDeviceToHost, st));
        COMM_HIP_TRY(hipStreamSynchronize(st));
        if (int rc = shm_barrier(c)) return rc;
        for (int r = 0; r < c->world; ++r)
            COMM_HIP_TRY(hipMemcpyAsync((uint8_t *)recv_dev + (size_t)r * (size_t)bytes + off, area(c) + (size_t)r * slice_max, n,
                                        hipMemcpyHostToDevice, st));
        COMM_HIP_TRY(hipStreamSynchronize(st));
        if (int rc = shm_barrier(c)) return rc; // nobody refills the area before everybody has read it
    }
    return 0;
}

int trmc_comm_all_gather_host(trmc_comm *c, const void *send, void *recv, int64_t bytes)
{
    if (!c) return fail_with(TRMC_EINVAL, "comm is NULL");
    if (bytes < 0) return fail_with(TRMC_EINVAL, "bytes < 0");
    if (bytes == 0) return 0;
    if (!send || !recv) return fail_with(TRMC_EINVAL, "send/recv is NULL");
    if (c->world == 1) {
        std::memmove(recv, send, (size_t)bytes);
        return 0;
    }
    if (c->use_rccl) {
        if (int rc = use_device(c)) return rc;
        const size_t b = (size_t)bytes, pad = (b + 255) / 256 * 256;
        if (int rc = stage_ensure(c, pad * (size_t)(c->world + 1))) return rc;
        if (!c->stage_stream) COMM_HIP_TRY(hipStreamCreateWithFlags(&c->stage_stream, hipStreamNonBlocking));
        uint8_t *s = (uint8_t *)c->stage, *r = s + pad;
        COMM_HIP_TRY(hipMemcpyAsync(s, send, b, hipMemcpyHostToDevice, c->stage_stream));
        const ncclResult_t rc = rccl().AllGather(s, r, b, ncclChar, c->nccl, c->stage_stream);
        if (rc != ncclSuccess) return rccl_fail("ncclAllGather", rc);
        COMM_HIP_TRY(hipMemcpyAsync(recv, r, b * (size_t)c->world, hipMemcpyDeviceToHost, c->stage_stream));
        COMM_HIP_TRY(hipStreamSynchronize(c->stage_stream));
        return 0;
    }
    const size_t slice_max = c->capacity / (size_t)c->world / 256 * 256;
    if (slice_max == 0) return fail_with(TRMC_EINVAL, "shared segment too small for this world size");
    for (size_t off = 0; off < (size_t)bytes; off += slice_max) {
        const size_t n = std::min(slice_max, (size_t)bytes - off);
        std::memcpy(area(c) + (size_t)c->rank * slice_max, (const uint8_t *)send + off, n);
        if (int rc = shm_barrier(c)) return rc;
        for (int r = 0; r < c->world; ++r)
            std::memcpy((uint8_t *)recv + (size_t)r * (size_t)bytes + off, area(c) + (size_t)r * slice_max, n);
        if (int rc = shm_barrier(c)) return rc;
    }
    return 0;
}

void trmc_comm_destroy(trmc_comm *c)
{
    if (!c) return;
    if (c->device >= 0) (void)hipSetDevice(c->device);
    if (c->stage) (void)hipFree(c->stage);
    if (c->stage_stream) (void)hipStreamDestroy(c->stage_stream);
    if (c->use_rccl) {
        if (c->nccl && rccl().CommDestroy) (void)rccl().CommDestroy(c->nccl);
    } else {
        if (c->base) {
            // the last rank to leave removes the name
            const uint32_t left = header(c)->attached.fetch_sub(1, std::memory_order_acq_rel) - 1;
            munmap(c->base, c->bytes);
            if (left == 0) (void)shm_unlink(c->shm_name.c_str());
        }
        if (c->fd >= 0) close(c->fd);
    }
    delete c;
}

// ---- device memory, streams, events: what the host side needs to order a hand-off against a plan's stream ----------
int trmc_dev_alloc(int device, int64_t bytes, void **ptr_out)
{
    if (!ptr_out) return fail_with(TRMC_EINVAL, "ptr_out is NULL");
    *ptr_out = nullptr;
    if (bytes < 0) return fail_with(TRMC_EINVAL, "bytes < 0");
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipMalloc(ptr_out, bytes > 0 ? (size_t)bytes : 1));
    COMM_HIP_TRY(hipMemset(*ptr_out, 0, bytes > 0 ? (size_t)bytes : 1));
    COMM_HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int trmc_dev_free(int device, void *ptr)
{
    if (!ptr) return 0;
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipFree(ptr));
    return 0;
}

int trmc_dev_upload(int device, void *dst_dev, const void *src_host, int64_t bytes)
{
    if (bytes == 0) return 0;
    if (!dst_dev || !src_host || bytes < 0) return fail_with(TRMC_EINVAL, "bad upload arguments");
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipMemcpy(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice));
    return 0;
}

int trmc_dev_download(int device, void *dst_host, const void *src_dev, int64_t bytes, void *stream)
{
    if (bytes == 0) return 0;
    if (!dst_host || !src_dev || bytes < 0) return fail_with(TRMC_EINVAL, "bad download arguments");
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    COMM_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int trmc_dev_copy(int device, void *dst_dev, const void *src_dev, int64_t bytes, void *stream)
{
    if (bytes == 0) return 0;
    if (!dst_dev || !src_dev || bytes < 0) return fail_with(TRMC_EINVAL, "bad copy arguments");
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipMemcpyAsync(dst_dev, src_dev, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int trmc_dev_download_async(int device, void *dst_host, const void *src_dev, int64_t bytes, void *stream)
{
    if (bytes == 0) return 0;
    if (!dst_host || !src_dev || bytes < 0) return fail_with(TRMC_EINVAL, "bad download arguments");
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}

int trmc_dev_gather_rows(int device, const void *src_dev, const int64_t *index_dev, int64_t nrows, int64_t row_bytes,
                         void *dst_dev, void *stream)
{
    if (nrows == 0 || row_bytes == 0) return 0;
    if (!src_dev || !index_dev || !dst_dev || nrows < 0 || row_bytes < 0 || row_bytes % 4 != 0)
        return fail_with(TRMC_EINVAL, "bad gather arguments (row_bytes must be a multiple of 4)");
    COMM_HIP_TRY(hipSetDevice(device));
    const int64_t words = row_bytes / 4, work = nrows * words;
    hipLaunchKernelGGL(k_gather_rows_indexed, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t *)src_dev, index_dev, (uint32_t *)dst_dev, nrows, words);
    COMM_HIP_TRY(hipGetLastError());
    return 0;
}

int trmc_stream_create(int device, void **stream_out)
{
    if (!stream_out) return fail_with(TRMC_EINVAL, "stream_out is NULL");
    *stream_out = nullptr;
    COMM_HIP_TRY(hipSetDevice(device));
    hipStream_t s = nullptr;
    COMM_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = (void *)s;
    return 0;
}

int trmc_stream_create_prio(int device, int priority, void **stream_out)
{
    if (!stream_out) return fail_with(TRMC_EINVAL, "stream_out is NULL");
    *stream_out = nullptr;
    COMM_HIP_TRY(hipSetDevice(device));
    int lo = 0, hi = 0; // (numerically: lo is the LEAST urgent, hi the most)
    COMM_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s = nullptr;
    COMM_HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority < 0 ? lo : priority > 0 ? hi : (lo + hi) / 2));
    *stream_out = (void *)s;
    return 0;
}

int trmc_stream_destroy(int device, void *stream)
{
    if (!stream) return 0;
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

int trmc_stream_synchronize(int device, void *stream)
{
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int trmc_device_synchronize(int device)
{
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int trmc_event_create(int device, void **event_out)
{
    if (!event_out) return fail_with(TRMC_EINVAL, "event_out is NULL");
    *event_out = nullptr;
    COMM_HIP_TRY(hipSetDevice(device));
    hipEvent_t e = nullptr;
    COMM_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event_out = (void *)e;
    return 0;
}

int trmc_event_destroy(int device, void *event)
{
    if (!event) return 0;
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipEventDestroy((hipEvent_t)event));
    return 0;
}

int trmc_event_record(int device, void *event, void *stream)
{
    if (!event) return fail_with(TRMC_EINVAL, "event is NULL");
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return 0;
}

int trmc_stream_wait_event(int device, void *stream, void *event)
{
    if (!event) return fail_with(TRMC_EINVAL, "event is NULL");
    COMM_HIP_TRY(hipSetDevice(device));
    COMM_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return 0;
}

} // extern "C"
