// trt_dist.hip -- one frame sharded over the GPUs of a node, behind the C-ABI (include/trt_hip.h, section 3).
//
// The frame producer is embarrassingly parallel over pixels; the only exchange is assembling the
// framebuffer.  One trt_dist per rank (one process -- or thread -- per GPU).  Rows are dealt in interleaved tiles of
// `tile_rows` rows (tile t -> rank t mod world, trt_rowset) so that sky rows and sphere / floor rows, which differ ~10x in
// cost, spread evenly.  Per frame every rank renders its tiles into a compact shard buffer on the stream of one of its
// `depth` renderer contexts, then ONE gather brings the shards to rank 0: ncclSend on the peers, `world - 1` ncclRecv on
// the root inside one ncclGroupStart/End, all on the communicator's own stream (RCCL over xGMI: each peer sends over its
// own direct link to the root, so the gather is not ring-bound), and one small kernel on the root puts the rows into frame
// order.  Frames are pipelined: frame f + 1 renders on the next context while frame f is gathered; a slot renders again
// only after its previous gather has been enqueued (events, no host waits).  The per-frame host work is this file's
// trt_dist_render: a handful of launches and event calls.
//
// RCCL is bound at run time (dlopen of librccl.so.1; a process that already carries one -- PyTorch bundles its own -- keeps
// using that one), so libtrt_hip.so itself does not depend on it and single-GPU hosts never load it.
#include "trt_hip.h"
#include "trt_hip_diag.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

namespace
{

thread_local char g_dist_error[512] = "";

int dist_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_dist_error, sizeof g_dist_error, fmt, ap);
    va_end(ap);
    return code;
}

#define DIST_HIP(expr)                                                                                             \
    do                                                                                                             \
    {                                                                                                              \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess)                                                                                      \
            return dist_fail(TRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// the nine RCCL entry points used, bound by name
struct Rccl
{
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    char why[256] = "";
};

// Bound once, by whichever thread comes first (one rank may be one THREAD per GPU: trt_hip.h); the struct is published only
// after every symbol is bound.  TRT_RCCL_LIB names another library with the same eight entry points (the tests' stand-in
// that lets several ranks share one GPU, tests/rccl_stub.cpp).  It is a TEST HOOK and is honoured only by a process that has
// asked for it BEFORE the first use of RCCL (trt_dist_allow_rccl_override(1)): a product process ignores the variable and binds
// RCCL itself.  trt_dist_rccl_library() says which library was bound.
Rccl g_rccl; // written inside the call_once below, read-only afterwards
std::atomic<int> g_override_allowed{0};
std::atomic<int> g_bind_started{0};  // the call_once has begun: the override can no longer be allowed or withdrawn
std::atomic<int> g_name_published{0}; // ... and g_bound_name is complete (set with release after the name is written: a rank may be a thread)
char g_bound_name[512] = "";          // written inside the call_once before g_name_published, read-only afterwards
const char *bound_name() { return g_name_published.load(std::memory_order_acquire) ? g_bound_name : ""; }

Rccl *rccl()
{
    Rccl &lib = g_rccl;
    static Rccl *published = nullptr;
    static std::once_flag once;
    std::call_once(once, [&lib] {
        g_bind_started.store(1);
        const char *override_name = g_override_allowed.load() ? getenv("TRT_RCCL_LIB") : nullptr;
        const char *names[] = {override_name && *override_name ? override_name : "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        const int count = override_name && *override_name ? 1 : 3; // an override that cannot be loaded is an error, not a reason to look elsewhere
        void *handle = nullptr;
        for (int i = 0; i < count && !handle; i++)
        {
            handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
            if (handle)
                snprintf(g_bound_name, sizeof g_bound_name, "%s%s", override_name && *override_name ? "STAND-IN (TRT_RCCL_LIB): " : "", names[i]);
        }
        g_name_published.store(1, std::memory_order_release); // "" if nothing could be loaded
        if (!handle)
        {
            const char *why = dlerror();
            snprintf(lib.why, sizeof lib.why, "%s", why ? why : "dlopen failed");
            return;
        }
        bool ok = true;
        auto bind = [&](const char *symbol) {
            void *p = dlsym(handle, symbol);
            ok = ok && p;
            return p;
        };
        lib.GetUniqueId = (decltype(lib.GetUniqueId))bind("ncclGetUniqueId");
        lib.CommInitRank = (decltype(lib.CommInitRank))bind("ncclCommInitRank");
        lib.CommDestroy = (decltype(lib.CommDestroy))bind("ncclCommDestroy");
        lib.GroupStart = (decltype(lib.GroupStart))bind("ncclGroupStart");
        lib.GroupEnd = (decltype(lib.GroupEnd))bind("ncclGroupEnd");
        lib.Send = (decltype(lib.Send))bind("ncclSend");
        lib.Recv = (decltype(lib.Recv))bind("ncclRecv");
        lib.GetErrorString = (decltype(lib.GetErrorString))bind("ncclGetErrorString");
        lib.CommCount = (decltype(lib.CommCount))bind("ncclCommCount");
        if (!ok)
        {
            snprintf(lib.why, sizeof lib.why, "the RCCL library lacks an entry point");
            return;
        }
        lib.handle = handle;
        published = &lib;
    });
    return published;
}

// after rccl() has returned nullptr: why
const char *rccl_why()
{
    static thread_local char text[400];
    snprintf(text, sizeof text, "the RCCL library could not be bound (multi-GPU needs librccl.so.1, or the library TRT_RCCL_LIB names): %s",
             g_rccl.why[0] ? g_rccl.why : "unknown reason");
    return text;
}

#define DIST_NCCL(R, expr)                                                                                          \
    do                                                                                                              \
    {                                                                                                               \
        ncclResult_t r_ = (expr);                                                                                   \
        if (r_ != ncclSuccess)                                                                                      \
            return dist_fail(TRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// rows of the rank-major, padded gather buffer -> frame order: one thread per element (a double of the Screen layout, or a
// byte of the emitter's (int)(c*255) triplets)
template <class T>
__global__ void assemble_rows_kernel(const T *gathered, const int *source_row, T *frame, long row_elements, long total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total)
        return;
    const long row = i / row_elements, at = i - row * row_elements;
    frame[i] = gathered[(long)source_row[row] * row_elements + at];
}

struct Slot
{
    trt_context *ctx = nullptr;
    hipStream_t stream = nullptr; // the context's own
    double *shard = nullptr;      // this rank's rows (padded to the largest shard); on the root a view into `gathered`
    double *gathered = nullptr;   // root: world x max_rows rows, rank-major
    double *frame = nullptr;      // root: height rows in frame order
    // the same three for the frame as the emitter's bytes, 3 per pixel (trt_dist_enable_rgb8)
    unsigned char *shard8 = nullptr, *gathered8 = nullptr, *frame8 = nullptr;
    hipEvent_t rendered = nullptr, consumed = nullptr;
    hipEvent_t gather_begin = nullptr; // on the communicator's stream once the shard is rendered; `consumed` ends the gather (both with timing)
    bool used = false, gathered_once = false;
};

} // namespace

struct trt_dist
{
    int device = 0, rank = 0, world = 1, root = 0;
    int width = 0, height = 0, tile_rows = 8;
    trt_rowset rows{};
    int local_rows = 0, max_rows = 0;
    std::vector<int> rows_of_rank; // rows owned by every rank
    std::vector<Slot> slots;
    hipStream_t comm_stream = nullptr;
    ncclComm_t comm = nullptr;
    int *d_source_row = nullptr; // root: frame row -> row of the gather buffer
    bool through_comm = false;   // world > 1, or world == 1 with an id given: the gather path is taken (with no peers to receive from)
    bool rgb8 = false;           // byte buffers allocated (trt_dist_enable_rgb8)
    int comm_ranks = 0;          // what ncclCommCount says of the communicator that was created (0: none, the gather path is not taken)
    bool poisoned = false;       // a collective failed on this rank: its peers may have gone on without it, nothing can be trusted any more
    long calls = 0;
};

extern "C" const char *trt_dist_last_error(void) { return g_dist_error; }

extern "C" int trt_dist_allow_rccl_override(int allow)
{
    if (g_bind_started.load())
        return dist_fail(TRT_ERR_NOT_INITIALISED, "RCCL has been bound already (%s): the override must be allowed before the first use", bound_name());
    g_override_allowed.store(allow ? 1 : 0);
    return TRT_OK;
}

extern "C" const char *trt_dist_rccl_library(void) { return bound_name(); }

// The root's assembly map (pure host arithmetic, no GPU): frame row -> row of the rank-major gather buffer in which rank r's
// shard starts at row r * max_rows.  Returns max_rows (the padded shard height), or a negative TRT_ERR_*.
extern "C" int trt_dist_source_rows(int width, int height, int tile_rows, int world, int *source_row)
{
    if (width < 1 || height < 1 || tile_rows < 1 || world < 1 || !source_row)
        return dist_fail(TRT_ERR_ARGUMENT, "bad argument");
    int max_rows = 0;
    for (int r = 0; r < world; r++)
    {
        const trt_rowset rs = {width, height, tile_rows, r, world};
        max_rows = std::max(max_rows, trt_rowset_rows(&rs));
    }
    for (int row = 0; row < height; row++)
        source_row[row] = -1;
    for (int r = 0; r < world; r++)
    {
        const trt_rowset rs = {width, height, tile_rows, r, world};
        const int rows = trt_rowset_rows(&rs);
        for (int i = 0; i < rows; i++)
        {
            const int at = trt_rowset_frame_row(&rs, i);
            if (at < 0 || at >= height || source_row[at] != -1)
                return dist_fail(TRT_ERR_ARGUMENT, "the row tiles do not partition the frame");
            source_row[at] = r * max_rows + i;
        }
    }
    for (int row = 0; row < height; row++)
        if (source_row[row] < 0)
            return dist_fail(TRT_ERR_ARGUMENT, "the row tiles do not cover the frame");
    return max_rows;
}

extern "C" int trt_dist_unique_id(void *id_out)
{
    if (!id_out)
        return dist_fail(TRT_ERR_ARGUMENT, "id_out is NULL");
    Rccl *R = rccl();
    if (!R)
        return dist_fail(TRT_ERR_NOT_INITIALISED, "%s", rccl_why());
    static_assert(sizeof(ncclUniqueId) == TRT_DIST_ID_BYTES, "trt_hip.h states the size of the id");
    ncclUniqueId id;
    DIST_NCCL(R, R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return TRT_OK;
}

extern "C" int trt_dist_comm_ranks(trt_dist *d)
{
    if (!d)
        return dist_fail(TRT_ERR_ARGUMENT, "d is NULL");
    return d->comm_ranks;
}

extern "C" int trt_dist_destroy(trt_dist *d)
{
    if (!d)
        return TRT_OK;
    (void)hipSetDevice(d->device);
    for (Slot &s : d->slots)
        if (s.ctx)
            (void)trt_synchronize(s.ctx);
    if (d->comm_stream)
        (void)hipStreamSynchronize(d->comm_stream);
    if (d->comm)
        if (Rccl *R = rccl())
            (void)R->CommDestroy(d->comm);
    for (Slot &s : d->slots)
    {
        if (s.rendered)
            (void)hipEventDestroy(s.rendered);
        if (s.consumed)
            (void)hipEventDestroy(s.consumed);
        if (s.gather_begin)
            (void)hipEventDestroy(s.gather_begin);
        if (s.gathered)
            (void)hipFree(s.gathered);
        else if (s.shard)
            (void)hipFree(s.shard);
        if (s.frame)
            (void)hipFree(s.frame);
        if (s.gathered8)
            (void)hipFree(s.gathered8);
        else if (s.shard8)
            (void)hipFree(s.shard8);
        if (s.frame8)
            (void)hipFree(s.frame8);
        if (s.ctx)
            (void)trt_destroy(s.ctx);
    }
    if (d->d_source_row)
        (void)hipFree(d->d_source_row);
    if (d->comm_stream)
        (void)hipStreamDestroy(d->comm_stream);
    delete d;
    return TRT_OK;
}

extern "C" int trt_dist_create(int device, const Scene *scene, const void *id, int rank, int world, int width, int height, int tile_rows,
                               int frames_in_flight, int reserved_cus, trt_dist **out)
{
    if (!out)
        return dist_fail(TRT_ERR_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!scene || world < 1 || rank < 0 || rank >= world || width < 1 || height < 1 || tile_rows < 1 || frames_in_flight < 1 ||
        frames_in_flight > 8 || reserved_cus < 0 || (world > 1 && !id))
        return dist_fail(TRT_ERR_ARGUMENT, "bad argument (rank %d of %d, %d x %d, tiles of %d rows, %d frames in flight)", rank, world, width,
                         height, tile_rows, frames_in_flight);
    trt_dist *d = new trt_dist();
    d->device = device, d->rank = rank, d->world = world;
    d->through_comm = world > 1 || id != nullptr;
    d->width = width, d->height = height, d->tile_rows = tile_rows;
    d->rows = trt_rowset{width, height, tile_rows, rank, world};
    d->local_rows = trt_rowset_rows(&d->rows);
    for (int r = 0; r < world; r++)
    {
        const trt_rowset rs = {width, height, tile_rows, r, world};
        d->rows_of_rank.push_back(trt_rowset_rows(&rs));
        d->max_rows = std::max(d->max_rows, d->rows_of_rank.back());
    }
    auto bail = [&](int rc) {
        char keep[sizeof g_dist_error];
        memcpy(keep, g_dist_error, sizeof keep);
        (void)trt_dist_destroy(d);
        memcpy(g_dist_error, keep, sizeof keep);
        return rc;
    };
#define DIST_TRY(expr)                \
    do                                \
    {                                 \
        const int rc_ = (expr);       \
        if (rc_ != TRT_OK)            \
            return bail(rc_);         \
    } while (0)
#define DIST_TRT(expr)                                                                     \
    do                                                                                     \
    {                                                                                      \
        const int rc_ = (expr);                                                            \
        if (rc_ != TRT_OK)                                                                 \
            return bail(dist_fail(rc_, "%s: %s", #expr, trt_last_error()));                \
    } while (0)
#define DIST_HIP_B(expr)                                                                                         \
    do                                                                                                           \
    {                                                                                                            \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
            return bail(dist_fail(TRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__)); \
    } while (0)

    DIST_HIP_B(hipSetDevice(device));
    const size_t row_doubles = (size_t)width * 3;
    d->slots.resize((size_t)frames_in_flight);
    for (Slot &s : d->slots)
    {
        DIST_TRT(trt_create(device, &s.ctx));
        // ONE copy of the scene and of its candidate tables per device: the first slot builds them, the others render from them
        // (trt_share_scene); only the eye's tables, the scratch and the frame buffers are a slot's own
        if (&s == &d->slots[0])
            DIST_TRT(trt_set_scene(s.ctx, scene));
        else
            DIST_TRT(trt_share_scene(s.ctx, d->slots[0].ctx));
        if (reserved_cus > 0)
            DIST_TRT(trt_reserve_cus(s.ctx, reserved_cus));
        void *stream = nullptr;
        DIST_TRT(trt_get_stream(s.ctx, &stream));
        s.stream = (hipStream_t)stream;
        if (rank == d->root && d->through_comm)
        { // the root renders straight into its part of the gather buffer
            DIST_HIP_B(hipMalloc((void **)&s.gathered, (size_t)world * d->max_rows * row_doubles * sizeof(double)));
            s.shard = s.gathered + (size_t)rank * d->max_rows * row_doubles;
            DIST_HIP_B(hipMalloc((void **)&s.frame, (size_t)height * row_doubles * sizeof(double)));
        }
        else
            DIST_HIP_B(hipMalloc((void **)&s.shard, (size_t)std::max(d->max_rows, 1) * row_doubles * sizeof(double)));
        DIST_HIP_B(hipEventCreateWithFlags(&s.rendered, hipEventDisableTiming));
        DIST_HIP_B(hipEventCreate(&s.consumed));     // with timing: trt_dist_frame_times
        DIST_HIP_B(hipEventCreate(&s.gather_begin));
    }
    if (d->through_comm)
    {
        Rccl *R = rccl();
        if (!R)
            return bail(dist_fail(TRT_ERR_NOT_INITIALISED, "%s", rccl_why()));
        DIST_HIP_B(hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking));
        ncclUniqueId uid;
        memcpy(&uid, id, sizeof uid);
        const ncclResult_t r = R->CommInitRank(&d->comm, world, uid, rank);
        if (r != ncclSuccess)
            return bail(dist_fail(TRT_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, R->GetErrorString(r)));
        // what the library itself says of the communicator: a record of "RCCL saw N ranks" for the caller (trt_dist_comm_ranks)
        const ncclResult_t rc = R->CommCount(d->comm, &d->comm_ranks);
        if (rc != ncclSuccess || d->comm_ranks != world)
            return bail(dist_fail(TRT_ERR_HIP, "the communicator has %d ranks, %d were asked for (%s)", d->comm_ranks, world, R->GetErrorString(rc)));
        if (rank == d->root)
        { // frame row -> row of the rank-major gather buffer (the tile map of trt_rowset_frame_row)
            std::vector<int> source((size_t)height, -1);
            if (trt_dist_source_rows(width, height, tile_rows, world, source.data()) != d->max_rows)
                return bail(dist_fail(TRT_ERR_ARGUMENT, "the row tiles do not cover the frame"));
            DIST_HIP_B(hipMalloc((void **)&d->d_source_row, source.size() * sizeof(int)));
            DIST_HIP_B(hipMemcpy(d->d_source_row, source.data(), source.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    *out = d;
    return TRT_OK;
#undef DIST_TRY
#undef DIST_TRT
#undef DIST_HIP_B
}

extern "C" int trt_dist_set_scene(trt_dist *d, const Scene *scene)
{
    if (!d || !scene)
        return dist_fail(TRT_ERR_ARGUMENT, "NULL argument");
    for (Slot &s : d->slots) // every slot's frames in flight still read the old tables
        if (trt_synchronize(s.ctx))
            return dist_fail(TRT_ERR_HIP, "trt_synchronize: %s", trt_last_error());
    for (Slot &s : d->slots)
    { // the first slot builds the new scene's tables (its old ones stay alive until the last sharer has let go), the others share them
        const int rc = &s == &d->slots[0] ? trt_set_scene(s.ctx, scene) : trt_share_scene(s.ctx, d->slots[0].ctx);
        if (rc)
        { // some slots may hold the new scene and others the old one: frames would alternate between the two without an error.
          // Nothing of this trt_dist can be trusted any more -- the same state a failed collective leaves (trt_hip.h).
            d->poisoned = true;
            return dist_fail(rc, "%s: %s (the renderer is unusable now: its frame slots may hold different scenes; destroy it)",
                             &s == &d->slots[0] ? "trt_set_scene" : "trt_share_scene", trt_last_error());
        }
    }
    return TRT_OK;
}

extern "C" int trt_dist_enable_rgb8(trt_dist *d)
{
    if (!d)
        return dist_fail(TRT_ERR_ARGUMENT, "d is NULL");
    if (d->rgb8)
        return TRT_OK;
    DIST_HIP(hipSetDevice(d->device));
    const size_t row_bytes = (size_t)d->width * 3;
    for (Slot &s : d->slots)
    { // a call that failed half-way may be repeated: what exists is kept
        if (d->rank == d->root && d->through_comm)
        {
            if (!s.gathered8)
                DIST_HIP(hipMalloc((void **)&s.gathered8, (size_t)d->world * d->max_rows * row_bytes));
            s.shard8 = s.gathered8 + (size_t)d->rank * d->max_rows * row_bytes;
            if (!s.frame8)
                DIST_HIP(hipMalloc((void **)&s.frame8, (size_t)d->height * row_bytes));
        }
        else if (!s.shard8)
            DIST_HIP(hipMalloc((void **)&s.shard8, (size_t)std::max(d->max_rows, 1) * row_bytes));
    }
    d->rgb8 = true;
    return TRT_OK;
}

// One frame: this rank's rows on the next slot's stream, then the gather on the communicator's.  `bytes`: what travels, and
// what the root assembles, are the emitter's (int)(c*255) triplets (TRT.c:1157-1163) instead of the doubles of the Screen layout.
static int render_and_gather(trt_dist *d, const Camera *camera, int bounce_limit, int rays_per_pixel, bool bytes, void **d_frame)
{
    if (!d || !camera)
        return dist_fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (d->poisoned)
        return dist_fail(TRT_ERR_NOT_INITIALISED, "an earlier collective failed on this rank: destroy the trt_dist on every rank");
    if (bytes && !d->rgb8)
        return dist_fail(TRT_ERR_NOT_INITIALISED, "trt_dist_enable_rgb8 has not been called");
    if (bounce_limit < 1 || rays_per_pixel < 1) // what trt_render_device would refuse: refuse it before anything is enqueued
        return dist_fail(TRT_ERR_ARGUMENT, "bounce_limit %d / rays_per_pixel %d", bounce_limit, rays_per_pixel);
    DIST_HIP(hipSetDevice(d->device));
    Slot &s = d->slots[(size_t)(d->calls % (long)d->slots.size())];
    const size_t row_doubles = (size_t)d->width * 3, row_elements = row_doubles; // 3 doubles or 3 bytes per pixel
    // From here on a failure leaves this rank out of step with its peers (they will issue a collective this rank has skipped, or
    // the other way round): the trt_dist is poisoned and every later call fails.
    auto poison = [&](int rc) {
        d->poisoned = true;
        return rc;
    };
    if (s.used && d->through_comm)
        if (hipStreamWaitEvent(s.stream, s.consumed, 0) != hipSuccess) // the slot's previous shard has left (or been assembled)
            return poison(dist_fail(TRT_ERR_HIP, "hipStreamWaitEvent failed"));
    d->calls++;
    s.used = true;
    if (d->local_rows > 0)
    {
        int rc = trt_render_device(s.ctx, camera, &d->rows, bounce_limit, rays_per_pixel, s.shard,
                                   (size_t)std::max(d->max_rows, 1) * row_doubles * sizeof(double));
        if (!rc && bytes)
            rc = trt_quantize_device(s.ctx, s.shard, (size_t)d->local_rows * d->width, s.shard8);
        if (rc)
            return poison(dist_fail(rc, "trt_render_device: %s", trt_last_error()));
    }
    if (!d->through_comm)
    {
        if (d_frame)
            *d_frame = bytes ? (void *)s.shard8 : (void *)s.shard; // a single renderer's rows are the frame, in order; valid in stream order of the slot's stream
        return TRT_OK;
    }
    Rccl *R = rccl();
    if (hipEventRecord(s.rendered, s.stream) != hipSuccess || hipStreamWaitEvent(d->comm_stream, s.rendered, 0) != hipSuccess)
        return poison(dist_fail(TRT_ERR_HIP, "event hand-over to the communicator's stream failed"));
    (void)hipEventRecord(s.gather_begin, d->comm_stream); // diagnostics only (trt_dist_frame_times)
    s.gathered_once = true;
    // the group is always closed, whatever happens inside it: the first error is kept and returned after ncclGroupEnd
    ncclResult_t first = R->GroupStart();
    const bool opened = first == ncclSuccess;
    const ncclDataType_t type = bytes ? ncclUint8 : ncclDouble;
    if (opened && d->rank == d->root)
    {
        for (int r = 0; r < d->world && first == ncclSuccess; r++)
            if (r != d->root && d->rows_of_rank[(size_t)r] > 0)
            {
                const size_t at = (size_t)r * d->max_rows * row_elements, count = (size_t)d->rows_of_rank[(size_t)r] * row_elements;
                first = R->Recv(bytes ? (void *)(s.gathered8 + at) : (void *)(s.gathered + at), count, type, r, d->comm, d->comm_stream);
            }
    }
    else if (opened && d->local_rows > 0)
        first = R->Send(bytes ? (const void *)s.shard8 : (const void *)s.shard, (size_t)d->local_rows * row_elements, type, d->root, d->comm, d->comm_stream);
    if (opened)
    {
        const ncclResult_t closed = R->GroupEnd();
        if (first == ncclSuccess)
            first = closed;
    }
    if (first != ncclSuccess)
        return poison(dist_fail(TRT_ERR_HIP, "the gather failed: %s", R->GetErrorString(first)));
    if (d->rank == d->root)
    {
        const long total = (long)d->height * (long)row_elements;
        if (bytes)
            hipLaunchKernelGGL(assemble_rows_kernel<unsigned char>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, d->comm_stream,
                               (const unsigned char *)s.gathered8, (const int *)d->d_source_row, s.frame8, (long)row_elements, total);
        else
            hipLaunchKernelGGL(assemble_rows_kernel<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, d->comm_stream,
                               (const double *)s.gathered, (const int *)d->d_source_row, s.frame, (long)row_elements, total);
        if (hipGetLastError() != hipSuccess)
            return poison(dist_fail(TRT_ERR_HIP, "the assembly kernel could not be launched"));
    }
    if (hipEventRecord(s.consumed, d->comm_stream) != hipSuccess)
        return poison(dist_fail(TRT_ERR_HIP, "hipEventRecord failed"));
    if (d_frame)
        *d_frame = d->rank == d->root ? (bytes ? (void *)s.frame8 : (void *)s.frame) : nullptr;
    return TRT_OK;
}

extern "C" int trt_dist_render(trt_dist *d, const Camera *camera, int bounce_limit, int rays_per_pixel, void **d_frame)
{
    return render_and_gather(d, camera, bounce_limit, rays_per_pixel, false, d_frame);
}

extern "C" int trt_dist_render_rgb8(trt_dist *d, const Camera *camera, int bounce_limit, int rays_per_pixel, void **d_frame_rgb8)
{
    return render_and_gather(d, camera, bounce_limit, rays_per_pixel, true, d_frame_rgb8);
}

extern "C" int trt_dist_synchronize(trt_dist *d)
{
    if (!d)
        return dist_fail(TRT_ERR_ARGUMENT, "d is NULL");
    if (d->poisoned)
        return dist_fail(TRT_ERR_NOT_INITIALISED, "an earlier collective failed on this rank: destroy the trt_dist on every rank");
    DIST_HIP(hipSetDevice(d->device));
    for (Slot &s : d->slots)
        DIST_HIP(hipStreamSynchronize(s.stream));
    if (d->comm_stream)
        DIST_HIP(hipStreamSynchronize(d->comm_stream));
    return TRT_OK;
}

// Diagnostics of the most recent frame of every slot, averaged over the slots: this rank's render time (render kernel + ordered
// mean, HIP events on the slot's stream) and the time its part of the gather took on the communicator's stream (from the moment
// the shard was rendered AND the stream was free, to the end of the send / of the last receive and the assembly): what makes a
// multi-GPU run slower than its slowest shard shows here, per rank.  Synchronises.  gather_ms = 0 without a communicator.
extern "C" int trt_dist_frame_times(trt_dist *d, float *render_ms, float *gather_ms)
{
    if (!d)
        return dist_fail(TRT_ERR_ARGUMENT, "d is NULL");
    const int rc = trt_dist_synchronize(d);
    if (rc)
        return rc;
    double render = 0.0, gather = 0.0;
    int renders = 0, gathers = 0;
    for (Slot &s : d->slots)
    {
        float r = 0.0f, m = 0.0f;
        if (s.used && d->local_rows > 0 && trt_render_kernel_times(s.ctx, &r, &m, 1) == 1)
            render += r + m, renders++;
        float g = 0.0f;
        if (s.gathered_once && hipEventElapsedTime(&g, s.gather_begin, s.consumed) == hipSuccess)
            gather += g, gathers++;
    }
    (void)hipGetLastError();
    if (render_ms)
        *render_ms = renders ? (float)(render / renders) : 0.0f;
    if (gather_ms)
        *gather_ms = gathers ? (float)(gather / gathers) : 0.0f;
    return TRT_OK;
}

extern "C" int trt_dist_info(const trt_dist *d, int *rank, int *world, int *local_rows, int *max_rows, int *frames_in_flight)
{
    if (!d)
        return dist_fail(TRT_ERR_ARGUMENT, "d is NULL");
    if (rank)
        *rank = d->rank;
    if (world)
        *world = d->world;
    if (local_rows)
        *local_rows = d->local_rows;
    if (max_rows)
        *max_rows = d->max_rows;
    if (frames_in_flight)
        *frames_in_flight = (int)d->slots.size();
    return TRT_OK;
}

extern "C" int trt_dist_fetch(trt_dist *d, const void *d_frame, Vector *pixels)
{
    if (!d || !d_frame || !pixels)
        return dist_fail(TRT_ERR_ARGUMENT, "NULL argument");
    const int rc = trt_dist_synchronize(d);
    if (rc)
        return rc;
    DIST_HIP(hipMemcpy(pixels, d_frame, (size_t)d->width * d->height * sizeof(Vector), hipMemcpyDeviceToHost));
    return TRT_OK;
}

extern "C" int trt_dist_fetch_rgb8(trt_dist *d, const void *d_frame_rgb8, unsigned char *rgb)
{
    if (!d || !d_frame_rgb8 || !rgb)
        return dist_fail(TRT_ERR_ARGUMENT, "NULL argument");
    const int rc = trt_dist_synchronize(d);
    if (rc)
        return rc;
    DIST_HIP(hipMemcpy(rgb, d_frame_rgb8, (size_t)d->width * d->height * 3, hipMemcpyDeviceToHost));
    return TRT_OK;
}

extern "C" trt_context *trt_dist_context(trt_dist *d, int slot)
{
    if (!d || slot < 0 || slot >= (int)d->slots.size())
        return nullptr;
    return d->slots[(size_t)slot].ctx;
}
