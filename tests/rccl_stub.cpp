// rccl_stub.cpp -- TEST INFRASTRUCTURE ONLY.  A stand-in for the eight RCCL entry points csrc/trt_dist.hip binds by name
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGroupStart/End, ncclSend, ncclRecv, ncclGetErrorString), so that
// the library's multi-GPU path -- the send branch, the root's receive offsets for every peer, unequal shard heights, slot
// re-use while a gather is outstanding -- can be EXECUTED by several processes that share the ONE GPU a test box has.  Real
// RCCL refuses a communicator with two ranks on one device; nothing here measures anything, and nothing of the product
// links or loads it: trt_dist.hip opens the library named by the environment variable TRT_RCCL_LIB instead of librccl.so.1,
// and only the tests set that variable (tests/test_gpu_parity.py::test_trt_dist_with_several_ranks_on_one_gpu).
//
// Semantics kept: point-to-point send/recv matched by (source, destination) in call order, STREAM-ORDERED like RCCL's: a
// transfer happens when the caller's stream reaches it.  Mechanism: a POSIX shared-memory segment per communicator, named
// after the 128-byte id; per ordered pair of ranks a staging slot and two counters.  A send enqueues on the stream, per
// chunk: a host function that waits until the slot is free, a device-to-host copy into the (page-locked) slot, a host function that
// publishes the chunk; a receive enqueues: wait until a chunk is published, host-to-device copy out of the slot, release.
// Every wait gives up after a deadline and poisons the communicator, so a broken test fails instead of hanging the box.
// TRT_RCCL_STUB_RING = R (default 1): R staging slots per ordered pair, used round robin, so that a sender may be R chunks ahead of
// its receiver.  With several ranks as THREADS of one process (the 8-rank tests: a GPU box admits six processes) the HIP runtime may
// run every stream's host functions on one thread, and a sender that blocked there would keep its process's other ranks from
// publishing: such tests choose slot size and R so that no sender ever has to wait (and keep the receiving root in a process of
// its own).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <new>
#include <string>
#include <thread>

#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

extern "C"
{
    typedef enum
    {
        ncclSuccess = 0,
        ncclUnhandledCudaError = 1,
        ncclSystemError = 2,
        ncclInternalError = 3,
        ncclInvalidArgument = 4,
        ncclInvalidUsage = 5
    } ncclResult_t;
    typedef struct
    {
        char internal[128];
    } ncclUniqueId;
    typedef struct stub_comm *ncclComm_t;
    typedef int ncclDataType_t; // rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5, ncclFloat16 6, ncclFloat32 7, ncclFloat64 8
}

namespace
{

constexpr int kMaxWorld = 8;
constexpr uint64_t kMagic = 0x7472745f73747562ull; // "trt_stub"

struct Channel
{
    std::atomic<uint64_t> produced, consumed; // chunks published by the sender, released by the receiver
    char pad[48];
};

struct Shared
{
    std::atomic<uint64_t> magic;
    std::atomic<int> arrived, failed;
    int world;
    int ring; // staging slots per ordered pair
    size_t slot_bytes;
    Channel channel[kMaxWorld * kMaxWorld]; // [source * world + destination]
};

size_t header_bytes() { return (sizeof(Shared) + 4095) / 4096 * 4096; }

double deadline_seconds()
{
    const char *e = getenv("TRT_RCCL_STUB_DEADLINE");
    return e ? atof(e) : 90.0;
}

} // namespace

struct stub_comm
{
    Shared *shared = nullptr;
    char *slots = nullptr; // kMaxWorld^2 slots of slot_bytes behind the header
    size_t mapped = 0;
    int rank = 0, world = 1;
    bool pinned[kMaxWorld * kMaxWorld] = {false}; // staging slots this process has page-locked (on first use)
    uint64_t sent[kMaxWorld] = {0}, received[kMaxWorld] = {0}; // chunks enqueued so far, per peer
};

namespace
{

struct Wait
{
    Shared *shared;
    std::atomic<uint64_t> *counter;
    uint64_t at_least;
};

struct Bump
{
    std::atomic<uint64_t> *counter;
};

bool spin_until(Shared *sh, const std::atomic<uint64_t> &counter, uint64_t at_least)
{
    const auto give_up = std::chrono::steady_clock::now() + std::chrono::duration<double>(deadline_seconds());
    unsigned spins = 0;
    while (counter.load(std::memory_order_acquire) < at_least)
    {
        if (sh->failed.load(std::memory_order_relaxed))
            return false;
        if ((++spins & 1023u) == 0)
        {
            if (std::chrono::steady_clock::now() > give_up)
            {
                sh->failed.store(1);
                fprintf(stderr, "rccl_stub: gave up waiting for a peer after %.0f s\n", deadline_seconds());
                return false;
            }
            std::this_thread::yield();
        }
    }
    return true;
}

void wait_callback(void *p)
{
    Wait *w = (Wait *)p;
    (void)spin_until(w->shared, *w->counter, w->at_least);
    delete w;
}

void bump_callback(void *p)
{
    Bump *b = (Bump *)p;
    b->counter->fetch_add(1, std::memory_order_release);
    delete b;
}

size_t type_bytes(ncclDataType_t t)
{
    switch (t)
    {
    case 0:
    case 1:
        return 1;
    case 6:
        return 2;
    case 2:
    case 3:
    case 7:
        return 4;
    case 4:
    case 5:
    case 8:
        return 8;
    default:
        return 0;
    }
}

ncclResult_t transfer(stub_comm *c, bool sending, void *buffer, size_t count, ncclDataType_t type, int peer, hipStream_t stream)
{
    const size_t size = type_bytes(type);
    if (!c || !buffer || !size || peer < 0 || peer >= c->world || peer == c->rank)
        return ncclInvalidArgument;
    if (c->shared->failed.load())
        return ncclSystemError;
    const int source = sending ? c->rank : peer, destination = sending ? peer : c->rank;
    Channel &ch = c->shared->channel[source * c->world + destination];
    const size_t ring = (size_t)c->shared->ring;
    char *slots = c->slots + (size_t)(source * c->world + destination) * ring * c->shared->slot_bytes; // the pair's `ring` slots
    if (!c->pinned[source * c->world + destination])
    { // The slots must be page-locked: an asynchronous copy from or to PAGEABLE host memory may touch the host buffer when the
      // call is made, not when the stream gets there -- before the peer has written the slot, or while it still reads it.
        if (hipHostRegister(slots, ring * c->shared->slot_bytes, hipHostRegisterDefault) != hipSuccess)
            return ncclUnhandledCudaError;
        c->pinned[source * c->world + destination] = true;
    }
    uint64_t &mine = sending ? c->sent[peer] : c->received[peer];
    size_t left = count * size, at = 0;
    while (left)
    {
        const size_t chunk = left < c->shared->slot_bytes ? left : c->shared->slot_bytes;
        // sender: slot (mine mod ring) is free once chunk mine - ring has been released; receiver: chunk number `mine` is there once
        // the sender has published mine + 1 chunks
        char *slot = slots + (size_t)(mine % ring) * c->shared->slot_bytes;
        Wait *w = new Wait{c->shared, sending ? &ch.consumed : &ch.produced, sending ? (mine + 1 > ring ? mine + 1 - ring : 0) : mine + 1};
        if (hipLaunchHostFunc(stream, wait_callback, w) != hipSuccess)
            return ncclUnhandledCudaError;
        const hipError_t e = sending ? hipMemcpyAsync(slot, (const char *)buffer + at, chunk, hipMemcpyDeviceToHost, stream)
                                     : hipMemcpyAsync((char *)buffer + at, slot, chunk, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess)
            return ncclUnhandledCudaError;
        Bump *b = new Bump{sending ? &ch.produced : &ch.consumed};
        if (hipLaunchHostFunc(stream, bump_callback, b) != hipSuccess)
            return ncclUnhandledCudaError;
        mine++;
        at += chunk;
        left -= chunk;
    }
    return ncclSuccess;
}

} // namespace

extern "C" ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id)
        return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    unsigned long long salt = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof id->internal, "/trt_rccl_stub_%ld_%llx", (long)getpid(), salt);
    return ncclSuccess;
}

extern "C" ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank)
{
    if (!out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || id.internal[0] != '/' || id.internal[sizeof id.internal - 1])
        return ncclInvalidArgument;
    const char *mb = getenv("TRT_RCCL_STUB_SLOT_MB");
    const size_t slot_bytes = (size_t)(mb ? atoi(mb) : 32) << 20;
    const char *rg = getenv("TRT_RCCL_STUB_RING");
    const int ring = rg && atoi(rg) > 0 ? atoi(rg) : 1;
    const size_t total = header_bytes() + (size_t)kMaxWorld * kMaxWorld * (size_t)ring * slot_bytes; // sparse: only touched pages exist
    bool creator = true;
    int fd = shm_open(id.internal, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 && errno == EEXIST)
    {
        creator = false;
        fd = shm_open(id.internal, O_RDWR, 0600);
    }
    if (fd < 0)
        return ncclSystemError;
    if (creator && ftruncate(fd, (off_t)total) != 0)
    {
        close(fd);
        shm_unlink(id.internal);
        return ncclSystemError;
    }
    if (!creator)
    { // the creator sizes the segment before anybody maps it
        struct stat st;
        for (int tries = 0; tries < 3000; tries++)
        {
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= total)
                break;
            usleep(10000);
        }
    }
    void *base = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED)
        return ncclSystemError;
    stub_comm *c = new stub_comm();
    c->shared = (Shared *)base;
    c->slots = (char *)base + header_bytes();
    c->mapped = total;
    c->rank = rank;
    c->world = world;
    if (creator)
    {
        c->shared->world = world;
        c->shared->ring = ring;
        c->shared->slot_bytes = slot_bytes;
        c->shared->magic.store(kMagic, std::memory_order_release);
    }
    else
    {
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::duration<double>(deadline_seconds());
        while (c->shared->magic.load(std::memory_order_acquire) != kMagic)
        {
            if (std::chrono::steady_clock::now() > give_up)
                return ncclSystemError;
            usleep(1000);
        }
        if (c->shared->world != world || c->shared->slot_bytes != slot_bytes || c->shared->ring != ring)
            return ncclInvalidArgument;
    }
    // everybody meets here, like ncclCommInitRank; then the name can go (the mappings stay), so nothing is left behind
    c->shared->arrived.fetch_add(1);
    const auto give_up = std::chrono::steady_clock::now() + std::chrono::duration<double>(deadline_seconds());
    while (c->shared->arrived.load() < world)
    {
        if (std::chrono::steady_clock::now() > give_up)
        {
            c->shared->failed.store(1);
            if (creator)
                shm_unlink(id.internal);
            return ncclSystemError;
        }
        usleep(1000);
    }
    if (creator)
        shm_unlink(id.internal);
    *out = c;
    return ncclSuccess;
}

extern "C" ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c)
        return ncclSuccess;
    const bool failed = c->shared->failed.load() != 0;
    for (int i = 0; i < kMaxWorld * kMaxWorld; i++)
        if (c->pinned[i])
            (void)hipHostUnregister(c->slots + (size_t)i * (size_t)c->shared->ring * c->shared->slot_bytes);
    munmap((void *)c->shared, c->mapped);
    delete c;
    return failed ? ncclSystemError : ncclSuccess;
}

extern "C" ncclResult_t ncclCommCount(const ncclComm_t c, int *count)
{
    if (!c || !count)
        return ncclInvalidArgument;
    *count = c->shared->arrived.load(); // the ranks that actually met in ncclCommInitRank
    return ncclSuccess;
}

extern "C" ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
extern "C" ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }

extern "C" ncclResult_t ncclSend(const void *buffer, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream)
{
    return transfer(c, true, (void *)buffer, count, type, peer, stream);
}

extern "C" ncclResult_t ncclRecv(void *buffer, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream)
{
    return transfer(c, false, buffer, count, type, peer, stream);
}

extern "C" const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r)
    {
    case ncclSuccess:
        return "no error";
    case ncclUnhandledCudaError:
        return "rccl_stub: a HIP call failed";
    case ncclSystemError:
        return "rccl_stub: shared memory, or a peer that never came";
    case ncclInvalidArgument:
        return "rccl_stub: invalid argument";
    default:
        return "rccl_stub: error";
    }
}
