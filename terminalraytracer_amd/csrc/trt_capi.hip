// trt_capi.hip -- host side of libtrt_hip.so: contexts, uploads, launches.  C-ABI of include/trt_hip.h.
// Compiled for gfx950 only, with -ffp-contract=off (see trt_device.hpp).
#include "trt_hip.h"
#include "trt_hip_diag.h"

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

// light-space candidate masks: cells per side of a directional light's grid / of a point light's cube-map face
#ifndef TRT_DIRGRID_CELLS
#define TRT_DIRGRID_CELLS 128
#endif
#ifndef TRT_POINTGRID_CELLS
#define TRT_POINTGRID_CELLS 64
#endif
// ... and their third coordinate (trt_lightgrid.h (5)): slabs of depth along a directional light, shells of distance from a point light
#ifndef TRT_DIRGRID_SLABS
#define TRT_DIRGRID_SLABS 16
#endif
#ifndef TRT_POINTGRID_SHELLS
#define TRT_POINTGRID_SHELLS 16
#endif

// candidate tables of the path rays' families (trt_raygrid.h): cells per side of a cube-map face for the two families of
// the eye / for the 2N families of the spheres
#ifndef TRT_PATHGRID_EYE
#define TRT_PATHGRID_EYE 64
#endif
#ifndef TRT_PATHGRID_SPHERE
#define TRT_PATHGRID_SPHERE 32
#endif
// below this many spheres the wave-uniform sweep (9 VALU per sphere) is cheaper than a table look-up with its membership
// test: measured 1.097 against 1.122 ms at 8 spheres (BASELINE config 2), 0.159 against 0.169 ms at 6 (the demo scene)
#ifndef TRT_PATHGRID_MIN_SPHERES
#define TRT_PATHGRID_MIN_SPHERES 12
#endif
// Sub-families of the spheres (trt_raygrid.h): the surface of every sphere is cut into 6 m^2 patches with a family each.
// -1: by the number of spheres (dense scenes pay for the larger tables with much shorter candidate lists), 0: one family
// per sphere, 1..4: m.
#ifndef TRT_PATHGRID_PATCHES
#define TRT_PATHGRID_PATCHES -1
#endif
// TRT_PATHGRID_PATCHES = -1: scenes of at least this many spheres get m = 2 (24 patches per sphere)
#ifndef TRT_PATCHES_FROM_SPHERES
#define TRT_PATCHES_FROM_SPHERES 128
#endif

#include "trt_common.hpp"
#include "trt_rounds.hpp"
#include "trt_simple.hpp"

namespace
{

thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

 double host_seconds()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define HIP_TRY(expr)                                                                                      \
    do                                                                                                     \
    {                                                                                                      \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(TRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#if defined(TRT_MARKS) && TRT_MARKS == 2
constexpr int kCounterSlots = trt::kProfileAt + 64 * trt::kProfileKinds; // + the ISA profile's sums (tools/isa_profile.py)
#else
constexpr int kCounterSlots = 40;
#endif
// [path, shadow, rounds, phase-2 rounds, 24 stage stamps of the diagnostic build, swept, passes, loop diagnostics 30..36]
constexpr int kEventRing = 256;
constexpr double kPi = 3.14159265358979323846; // TRT.c:43

// TRT.c:225-228
double triangle_wave(double t)
{
    double m = fmod(t, 2 * kPi);
    return (m < kPi) ? (m / kPi) : (2 - (m / kPi));
}

template <typename T>
struct DeviceBuffer
{
    T *ptr = nullptr;
    size_t capacity = 0; // elements
    hipError_t reserve(size_t n)
    {
        if (n <= capacity && ptr)
            return hipSuccess;
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        capacity = 0;
        hipError_t e = hipMalloc((void **)&ptr, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess)
            capacity = std::max<size_t>(n, 1);
        return e;
    }
    void release()
    {
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        capacity = 0;
    }
};

} // namespace

// Everything on the device that depends on the SCENE only (primitives, cubemap, every candidate table but the eye's two): built by
// trt_set_scene, read-only afterwards, and shareable between the contexts of one device (trt_share_scene): the frame slots of a
// trt_dist render different cameras of ONE scene at the same time.  What depends on the camera -- the two tables of the eye's
// families and their part of the pool of long lists -- has kEyeSlots places in the same allocations, one per sharing context, so
// that the kernels keep reading ONE table base and ONE pool base whoever built what.
constexpr int kEyeSlots = 8; // = the most frames a trt_dist keeps in flight

struct SceneTables
{
    int device = 0;
    DeviceBuffer<double> d_spheres, d_dir, d_point;
    DeviceBuffer<float> d_cull;
    // light-space candidate masks (trt_lightgrid.h) and the host copy of the primitives they were built from
    DeviceBuffer<unsigned long long> d_dir_masks, d_point_masks;
    DeviceBuffer<trt_dirgrid> d_dirgrids;
    DeviceBuffer<trt_pointgrid> d_pointgrids;
    DeviceBuffer<trt_dirgrid_disc> d_discs;   // per directional light and sphere: what the marking kernel reads
    DeviceBuffer<trt_pointgrid_cone> d_cones; // per point light and sphere
    // the tables as LIST CELLS (trt_raygrid.h), which is what the kernel reads: one 64-bit word per cell, long lists in d_pool
    DeviceBuffer<unsigned long long> d_dir_lists, d_point_lists, d_path_lists, d_pool;
    // 64-bit counters of pool words taken: [0] by the scene's tables, [16 (1 + s)] by the eye's tables of slot s (a cache line apart).
    // 64 bits: a 32-bit counter that keeps counting after the pool is exhausted wraps, and lists would overwrite one another.
    DeviceBuffer<unsigned long long> d_pool_used;
    DeviceBuffer<trt_rayfamily> d_families;    // the 2NP families of the spheres, for the marking kernel
    DeviceBuffer<double> d_sphere_fam;         // per sphere {mirror centre, |r|}: what the render kernel keeps in LDS
    DeviceBuffer<double> d_patch_rec;          // per patch {t, rho, mirrored t, rho}: likewise
    DeviceBuffer<uint32_t> d_sky;
    int path_built_for[4] = {-1, -1, -1, -2};
    int grids_built_for[4] = {-1, -1, -1, -1};
    size_t pool_scene_words = 0, pool_eye_words = 0; // capacities: the scene's part of d_pool, then kEyeSlots parts of pool_eye_words
    trt_cull_scene cull_scene{};                      // of the spheres the tables were built from
    double ground_built[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    std::vector<double> h_spheres, h_dir, h_point; // what the tables on the device were built from
    unsigned eye_slots_taken = 0;                  // bit s: a context renders with the eye tables of slot s
    bool built_for_moving_scene = false;           // the cheap tables of a scene that changes from call to call
    double build_seconds = 0.0;                    // host time of the last table build (trt_scene_info)
    ~SceneTables()
    {
        (void)hipSetDevice(device);
        d_spheres.release(), d_dir.release(), d_point.release(), d_cull.release(), d_dir_masks.release(), d_point_masks.release();
        d_dirgrids.release(), d_pointgrids.release(), d_discs.release(), d_cones.release(), d_dir_lists.release(), d_point_lists.release();
        d_path_lists.release(), d_pool.release(), d_pool_used.release(), d_families.release(), d_sphere_fam.release(), d_patch_rec.release();
        d_sky.release();
    }
};

struct trt_context
{
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int compute_units = 0;
    int reserved_cus = 0; // CUs the context's own stream may not use (trt_reserve_cus)
    int lds_limit = 0;

    bool have_scene = false;
    std::shared_ptr<SceneTables> T; // never null after init_context; shared after trt_share_scene
    int eye_slot = 0;               // which of T's kEyeSlots places this context's eye tables live in
    trt::SceneView scene{};
    trt::CullView cull{};
    DeviceBuffer<double> d_jitter, d_fb, d_axes, d_samples, d_samples_alt;
    trt::GridView grids{};
    int dirgrid_cells = TRT_DIRGRID_CELLS, pointgrid_cells = TRT_POINTGRID_CELLS; // per side; 0 = no tables (sweep only)
    int dirgrid_slabs = TRT_DIRGRID_SLABS, pointgrid_shells = TRT_POINTGRID_SHELLS; // depth coordinate of the light tables (>= 1)
    int path_g_eye = TRT_PATHGRID_EYE, path_g_sph = TRT_PATHGRID_SPHERE; // 0 = no path tables (every path ray sweeps)
    int path_min_spheres = TRT_PATHGRID_MIN_SPHERES;                      // scenes with fewer spheres sweep
    int path_patches = TRT_PATHGRID_PATCHES;                              // m of the spheres' sub-families; -1: by the number of spheres
    size_t list_pool_cap = 0;                                             // trt_set_list_pool_words: cap on the scene's part of the pool (0 = automatic)
    // project_scene is a pure function of *scene (TRT.c:966): a caller of the drop-in entries may move a sphere before every call.
    // The drop-in layer counts consecutive calls whose primitives differ from the call before; from the second on the scene
    // counts as MOVING and its tables are built the cheap way (one family per sphere instead of 24 patches: 1/24 of the cells,
    // the dominant cost at 128+ spheres), and once it has been still for a few calls the full tables are built (trt_set_scene_policy).
    int scene_changes_in_a_row = 0, scene_still_calls = 0;
    bool moving_scene = false;
    double eye_built[3] = {0.0, 0.0, 0.0};
    bool eye_tables_valid = false;
    DeviceBuffer<double> d_ior; // refraction extension: per sphere, > 0 = index of refraction
    DeviceBuffer<unsigned char> d_rgb8; // trt_render_host_rgb8: the quantised frame before it crosses PCIe
    int ior_count = 0;          // 0 = off (the reference's path)
    DeviceBuffer<unsigned long long> d_counters;
    DeviceBuffer<unsigned int> d_queue;
    double *h_staging = nullptr; // pinned
    size_t h_staging_bytes = 0;

    // cache keys of the per-frame tables (jitter; per-column / per-row screen coordinates)
    int jit_spp = -1;
    double jit_pw = 0.0, jit_ph = 0.0;
    int axes_w = -1, axes_h = -1;
    double axes_sw = 0.0, axes_sh = 0.0;

    int kernel = 0; // 0 production (persistent waves, synchronous rounds), 1 reference-order
    int rounds_blocks_per_cu = 0;
    int compact_blocks_per_cu = 0; // the same for the kernel with shading rings in LDS
    long last_units = 0;           // samples of the most recent launch (trt_render_variant / trt_kernel_info describe that launch's kernel)
    bool last_compact = false;     // the most recent launch ran the kernel with the shading decoupled
    int compaction = -1;           // trt_set_compaction: -1 when it costs no occupancy, 0 never, 1 whenever the rings fit
    size_t occupancy_for_lds = (size_t)-1;
    hipEvent_t ev_chunk[16]; // hand-over of framebuffer chunks to the host copy threads (trt_render_host)
    hipEvent_t ev_band[8];   // a band of rows is rendered: its copy-out may start (trt_render_host)
    hipStream_t copy_stream = nullptr;
    hipStream_t alt_stream = nullptr; // second render stream of trt_render_host: odd bands (their tails overlap the next band)
    hipEvent_t ev_fork = nullptr;
    bool counters_enabled = false;
    unsigned long long last_trips = 0, last_phase2 = 0, last_swept = 0, last_passes = 0; // diagnostics of the counting kernel variant
    unsigned long long last_loops[8] = {0, 0, 0, 0, 0, 0, 0, 0};                        // trt_read_loop_diagnostics

    hipEvent_t ev_start[kEventRing], ev_mid[kEventRing], ev_stop[kEventRing]; // launch begins | render kernel done | reduction done
    long launches = 0;

    // skybox cache key of the default context
    const void *sky_faces[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int sky_dim = -1;
    unsigned long long sky_stamp = 0; // content stamp of the faces (sampled texels): a free-and-reload at the same addresses is noticed
};

namespace
{

constexpr int kCompactionMinLights = 2; // trt_set_compaction(-1): decouple the shading from two lights up (with one it is a wash)

// LDS image of the production kernel for the context's scene and tables
size_t image_lds_bytes(const trt_context *ctx, int spp)
{
    return trt::rounds_lds_bytes(ctx->scene, spp, ctx->grids.path_enabled ? ctx->grids.patch_count : 0);
}

// LDS of render_rounds_kernel<.., false, true>: the image, then one shading ring per wave of the workgroup
size_t compact_ring_at(const trt_context *ctx, int spp)
{
    return (image_lds_bytes(ctx, spp) / sizeof(double) + 1) & ~(size_t)1; // in doubles, on a 16-byte boundary
}

size_t compact_lds_bytes(const trt_context *ctx, int spp)
{
    return sizeof(double) * (compact_ring_at(ctx, spp) + (size_t)(trt::kCompactBlock / 64) * trt::kRingDoubles);
}

// Does a frame of `units` samples on this context run the kernel with the shading decoupled from the owning lane (COMPACT,
// trt_rounds.hpp)?  Measured (profiles/r02/n_compaction.md): 6 % faster with the two lights of the BASELINE scenes, 10 / 12 /
// 15 / 17 % with 3 / 4 / 6 / 8; the ring costs about what one light's idle lanes cost.  Its 1024-thread workgroups hold a
// whole CU until their last wave retires, which pipelined frames feel on SMALL launches (profiles/r02/t_shards.txt: a 1/8
// shard of the 1080p frame, three in flight, 0.249 ms decoupled against 0.218 plain; half a frame 0.884 against 0.871; the
// whole frame 1.630 against 1.685): by default only launches of 16 M samples or more are decoupled.
constexpr long kCompactionMinUnits = 16L << 20;

static bool renders_decoupled(const trt_context *ctx, long units)
{
    if (ctx->kernel != 0 || ctx->ior_count || ctx->compact_blocks_per_cu <= 0 || ctx->compaction == 0)
        return false;
    if (ctx->grids.path_enabled && ctx->grids.patch_m) // scenes whose spheres have patches (dense ones) run the plain rounds
        return false;
    // ... and only scenes whose path rays are served by tables: with the few spheres of a scene that sweeps (BASELINE configs[1]:
    // 8 spheres, most rays end on the ground or the sky) the ring costs more than the idle lanes (round 4, final kernel,
    // profiles/r04/i_all_configs_one_gpu.md: 43.3 G path rays/s plain against 40.8 decoupled; config 3 equal, config 4 +4 % decoupled)
    const bool pays = ctx->scene.num_dir + ctx->scene.num_point >= kCompactionMinLights && units >= kCompactionMinUnits && ctx->grids.path_enabled &&
                      ctx->compact_blocks_per_cu * trt::kCompactBlock >= ctx->rounds_blocks_per_cu * trt::kPersistentBlock;
    return ctx->compaction > 0 || pays;
}

size_t scene_lds_bytes(const trt::SceneView &s)
{
    return sizeof(double) * ((size_t)s.num_spheres * trt::kSphereDoubles + (size_t)s.num_dir * trt::kDirLightDoubles +
                             (size_t)s.num_point * trt::kPointLightDoubles);
}

// FNV-1a over 256 texels sampled at a fixed stride from every face: cheap enough for every frame of the drop-in call, and
// a different image loaded into the same allocation (free + malloc of the same size often returns the same pointers) shows
unsigned long long skybox_stamp(const Skybox *sky)
{
    unsigned long long h = 1469598103934665603ull;
    const size_t face = (size_t)sky->dim * sky->dim, step = face / 256 ? face / 256 : 1;
    for (int f = 0; f < 6; f++)
        for (size_t i = 0; i < face; i += step)
        {
            const Color c = sky->colors[f][i];
            h = (h ^ c.r) * 1099511628211ull;
            h = (h ^ c.g) * 1099511628211ull;
            h = (h ^ c.b) * 1099511628211ull;
        }
    const Color last = sky->colors[5][face - 1];
    return (h ^ ((unsigned)last.r << 16 | (unsigned)last.g << 8 | last.b)) * 1099511628211ull;
}

int upload_skybox(trt_context *ctx, const Skybox *sky)
{
    const int dim = sky->dim;
    if (dim <= 0)
        return fail(TRT_ERR_ARGUMENT, "skybox dim %d", dim);
    for (int f = 0; f < 6; f++)
        if (!sky->colors[f])
            return fail(TRT_ERR_ARGUMENT, "skybox face %d is NULL", f);
    const size_t face = (size_t)dim * dim;
    std::vector<uint32_t> packed(6 * face);
    for (int f = 0; f < 6; f++)
    {
        const Color *src = sky->colors[f];
        uint32_t *dst = packed.data() + f * face;
        for (size_t i = 0; i < face; i++)
            dst[i] = (uint32_t)src[i].r | ((uint32_t)src[i].g << 8) | ((uint32_t)src[i].b << 16);
    }
    HIP_TRY(ctx->T->d_sky.reserve(6 * face));
    HIP_TRY(hipMemcpy(ctx->T->d_sky.ptr, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    ctx->scene.sky = ctx->T->d_sky.ptr;
    ctx->scene.sky_dim = dim;
    ctx->scene.sky_dim_f = (double)dim;
    for (int f = 0; f < 6; f++)
        ctx->sky_faces[f] = sky->colors[f];
    ctx->sky_dim = dim;
    ctx->sky_stamp = skybox_stamp(sky);
    return TRT_OK;
}

// Marking kernels of the light-space tables: one thread per cell, every sphere tested with the predicates of
// trt_lightgrid.h (+ - * / sqrt only: the host reference builders in the tests produce the same bits).
__global__ void build_dirgrid_kernel(const trt_dirgrid_disc *discs, int n, int g, int slabs, int words, unsigned long long *masks)
{
    const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= (long)slabs * g * g)
        return;
    const int c = (int)(cell % g), j = (int)((cell / g) % g), slab = (int)(cell / ((long)g * g));
    for (int w = 0; w < words; w++)
    {
        unsigned long long m = 0;
        for (int b = 0; b < 64 && w * 64 + b < n; b++)
            if (trt_dirgrid_in_slab(discs + w * 64 + b, slab) && trt_dirgrid_reaches(discs + w * 64 + b, c, j))
                m |= 0x8000000000000000ull >> b;
        masks[cell * words + w] = m;
    }
}

__global__ void build_pointgrid_kernel(const trt_pointgrid_cone *cones, int n, int g, int shells, int words, unsigned long long *masks)
{
    const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= 6L * shells * g * g)
        return;
    const int shell = (int)(cell / (6L * g * g)), face = (int)((cell / ((long)g * g)) % 6), j = (int)((cell / g) % g), c = (int)(cell % g);
    for (int w = 0; w < words; w++)
    {
        unsigned long long m = 0;
        for (int b = 0; b < 64 && w * 64 + b < n; b++)
            if (trt_pointgrid_in_shell(cones + w * 64 + b, shell, shells) && trt_pointgrid_reaches(cones + w * 64 + b, face, c, j, g))
                m |= 0x8000000000000000ull >> b;
        masks[cell * words + w] = m;
    }
}

// Mask words of a table -> list cells (trt_raygrid.h).  Lists longer than seven entries take words from the pool; when
// the pool's part is exhausted the cell says TRT_LIST_NONE and its rays sweep.
// The counter has 64 bits: it keeps counting after the pool is exhausted (exhaustion is a normal mode: the cell then says "no
// list" and its rays sweep), and a 32-bit one would wrap after 2^32 words' worth of requests and hand out words that earlier
// cells already point to.
// Called by EVERY lane of a wave (`valid`: the lane has a cell): the lanes' requests are summed and the wave takes its words
// with ONE atomic -- a request per cell on the one counter serialises in L2 (the eye's tables at 256 spheres: 49 152 cells
// with pooled lists, ~0.1 ms of nothing but that).
__device__ unsigned long long pack_cell(const unsigned long long *mask, int words, unsigned long long *pool, unsigned long long *pool_used,
                                        unsigned pool_limit, int bits, bool valid = true)
{
    const int count = valid ? trt_list_count(mask, words) : 0;
    const unsigned need = valid ? trt_list_pool_words(count, bits) : 0u;
    const int lane = (int)(threadIdx.x & 63);
    unsigned upto = need; // inclusive prefix sum over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const unsigned below = __shfl_up(upto, d);
        upto += lane >= d ? below : 0u;
    }
    const unsigned total = __shfl(upto, 63);
    unsigned long long base = 0;
    if (total)
    {
        if (lane == 63)
            base = atomicAdd(pool_used, (unsigned long long)total);
        base = __shfl(base, 63);
    }
    if (!valid)
        return 0ull;
    if (need == 0)
        return trt_list_pack(mask, words, count, nullptr, 0u, bits);
    const unsigned long long at = base + (upto - need);
    if (count > 0xffff || at + need > (unsigned long long)pool_limit)
        return (unsigned long long)TRT_LIST_NONE << 56;
    return trt_list_pack(mask, words, count, pool, (unsigned)at, bits);
}

__global__ void set_pool_counter_kernel(unsigned long long *counter, unsigned long long value) { *counter = value; }

__global__ void pack_lists_kernel(const unsigned long long *masks, long cells, int words, unsigned long long *lists, unsigned long long *pool,
                                  unsigned long long *pool_used, unsigned pool_limit, int bits)
{
    const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = cell < cells;
    const unsigned long long packed = pack_cell(masks + (valid ? cell : 0) * words, words, pool, pool_used, pool_limit, bits, valid);
    if (valid)
        lists[cell] = packed;
}

// Direction tables of path-ray families (trt_raygrid.h): blockIdx.y = family.  When the table's side is a multiple of 8 a
// workgroup builds one TILE of 8 x 8 cells: every block first forms the cones of its family's apex (the same + - * / sqrt as
// the host reference builder: the same bits), thread i asks whether sphere i's cone reaches the TILE at all (trt_raygrid.h
// "marking the cells": the table's bit is tile AND cell), and then the four waves share the spheres that do -- wave q takes
// every fourth one -- each lane marking ITS cell with the cone / cell predicate of trt_lightgrid.h; the waves' masks are
// OR-ed in LDS and the first wave packs the lists.  (Every cell asking every sphere, 256 cells per block, took 0.95 ms per
// camera for the eye's two tables at 256 spheres: 192 blocks of long dependent FP64 chains.  Tiles of 16 x 16: 0.16 ms.)
// Otherwise: one thread per cell, 256 consecutive cells per block, every sphere.  `by_value`: the two families of the eye come
// as kernel arguments (they change with the camera), otherwise family blockIdx.y of `families`.
static_assert(TRT_FAMILY_TILE * TRT_FAMILY_TILE == 64, "a tile per wave");
constexpr int kSceneTilesPerBlock = 16; // the scene's sphere families: thousands of tables, 16 tiles from one set of cones
unsigned family_grid_blocks(int g, int tiles_per_block)
{
    const unsigned cells = 6u * (unsigned)g * (unsigned)g;
    return trt_family_tiled(g) ? (cells / 64u + (unsigned)tiles_per_block - 1u) / (unsigned)tiles_per_block : (cells + 255u) / 256u;
}

// kWords: 64-sphere mask words a cell can have -- 4 for scenes of up to 256 spheres (8-bit list entries), 16 for scenes of up to
// TRT_PATH_MAX_SPHERES = 1024 (16-bit entries; round 5: before, the path rays of a scene of more than 256 spheres swept).  The LDS
// is the launch's dynamic allocation (family_lds_bytes): the family's cones, the tile's reach words, the four waves' marks.
template <int kWords>
constexpr size_t family_lds_bytes()
{
    return sizeof(trt_pointgrid_cone) * 64 * kWords + sizeof(unsigned long long) * kWords + sizeof(unsigned long long) * 4 * 64 * kWords;
}

template <int kWords>
__global__ __launch_bounds__(256) void build_family_lists_kernel(const double *spheres, int n, const trt_rayfamily *families, trt_rayfamily f0,
                                                                 trt_rayfamily f1, int by_value, int g, unsigned long long *lists,
                                                                 unsigned long long *pool, unsigned long long *pool_used, unsigned pool_limit,
                                                                 int tiles_per_block)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char family_lds[];
    trt_pointgrid_cone *const cones = (trt_pointgrid_cone *)family_lds;                  // [64 kWords]
    unsigned long long *const reach = (unsigned long long *)(cones + 64 * kWords);       // [kWords]: bit k of word w: sphere 64 w + k reaches this block's tile
    unsigned long long(*const part)[64][kWords] = (unsigned long long(*)[64][kWords])(reach + kWords); // [4]: wave q's marks of the tile's cells
    const int bits = kWords > 4 ? 16 : 8; // entry width of the list cells (trt_raygrid.h)
    const trt_rayfamily F = by_value ? (blockIdx.y == 0 ? f0 : f1) : families[blockIdx.y];
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        trt_rayfamily_cone(&F, spheres + 9 * i, &cones[i]);
    __syncthreads();
    const unsigned cells = 6u * (unsigned)g * (unsigned)g;
    const int words = (n + 63) / 64 > 0 ? (n + 63) / 64 : 1;
    if (!trt_family_tiled(g))
    {
        const unsigned cell = blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = cell < cells;
        const int face = (int)(cell / ((unsigned)g * (unsigned)g)), j = (int)((cell / (unsigned)g) % (unsigned)g), c = (int)(cell % (unsigned)g);
        unsigned long long m[kWords];
#pragma unroll
        for (int w = 0; w < kWords; w++)
            m[w] = 0;
        for (int i = 0; valid && i < n; i++)
            if (trt_pointgrid_reaches(&cones[i], face, c, j, g))
                m[i >> 6] |= 0x8000000000000000ull >> (i & 63);
        const unsigned long long packed = pack_cell(m, words, pool, pool_used, pool_limit, bits, valid);
        if (valid)
            lists[(size_t)blockIdx.y * cells + cell] = packed;
        return;
    }
    // block -> tiles_per_block consecutive tiles (the scene's tables: thousands of families, a block builds a face's worth of
    // tiles from ONE set of cones; the eye's two tables: a tile per block, for the parallelism)
    const unsigned gt = (unsigned)g / TRT_FAMILY_TILE, tiles = 6u * gt * gt;
    const int lane = (int)(threadIdx.x & 63), q = (int)(threadIdx.x >> 6);
    for (unsigned tile = blockIdx.x * (unsigned)tiles_per_block; tile < tiles && tile < (blockIdx.x + 1u) * (unsigned)tiles_per_block; tile++)
    {
        const int face = (int)(tile / (gt * gt)), tj = (int)((tile / gt) % gt), tc = (int)(tile % gt);
        __syncthreads(); // the previous tile's reach[] and part[] have been read
#pragma unroll
        for (int chunk = 0; chunk < kWords / 4; chunk++) // thread t asks for spheres t, 256 + t, ...: wave q's ballot is word 4 chunk + q
        {
            const int i = 256 * chunk + (int)threadIdx.x;
            const bool reaches = i < n && trt_pointgrid_reaches(&cones[i], face, tc, tj, (int)gt);
            const unsigned long long word = __ballot(reaches); // the 64 spheres of this wave
            if (lane == 0)
                reach[4 * chunk + q] = word;
        }
        __syncthreads();
        const int j = tj * TRT_FAMILY_TILE + lane / TRT_FAMILY_TILE, c = tc * TRT_FAMILY_TILE + lane % TRT_FAMILY_TILE;
#pragma unroll
        for (int w = 0; w < kWords; w++)
        {
            unsigned long long m = 0;
            unsigned long long todo = 64 * w < n ? reach[w] & (0x1111111111111111ull << q) : 0ull; // the same in every lane of the wave: a scalar loop
            while (todo)
            {
                const int k = __builtin_ctzll(todo);
                todo &= todo - 1;
                if (trt_pointgrid_reaches(&cones[64 * w + k], face, c, j, g))
                    m |= 0x8000000000000000ull >> k;
            }
            part[q][lane][w] = m;
        }
        __syncthreads();
        if (q == 0)
        {
            unsigned long long m[kWords];
#pragma unroll
            for (int w = 0; w < kWords; w++)
                m[w] = part[0][lane][w] | part[1][lane][w] | part[2][lane][w] | part[3][lane][w];
            const unsigned cell = ((unsigned)face * (unsigned)g + (unsigned)j) * (unsigned)g + (unsigned)c;
            lists[(size_t)blockIdx.y * cells + cell] = pack_cell(m, words, pool, pool_used, pool_limit, bits);
        }
    }
}

// the builder for a scene of n spheres: 4 mask words (8-bit entries) up to 256 spheres, 16 (16-bit entries) up to 1024
void launch_family_builder(int n, dim3 grid, hipStream_t stream, const double *spheres, const trt_rayfamily *families, trt_rayfamily f0, trt_rayfamily f1,
                           int by_value, int g, unsigned long long *lists, unsigned long long *pool, unsigned long long *pool_used, unsigned pool_limit,
                           int tiles_per_block)
{
    if (n <= TRT_LIST_MAX_SPHERES)
        hipLaunchKernelGGL(build_family_lists_kernel<4>, grid, dim3(256), family_lds_bytes<4>(), stream, spheres, n, families, f0, f1, by_value, g, lists, pool,
                           pool_used, pool_limit, tiles_per_block);
    else
        hipLaunchKernelGGL(build_family_lists_kernel<TRT_PATH_MAX_SPHERES / 64>, grid, dim3(256), family_lds_bytes<TRT_PATH_MAX_SPHERES / 64>(), stream, spheres, n,
                           families, f0, f1, by_value, g, lists, pool, pool_used, pool_limit, tiles_per_block);
}

// Light-space candidate masks of every light (trt_lightgrid.h), from the context's host copy of the primitives: the
// host places each grid and prepares one small record per sphere and light, the device marks the cells.
int build_light_grids(trt_context *ctx, const trt_cull_scene &cs)
{
    const int n = (int)(ctx->T->h_spheres.size() / 9), nd = (int)(ctx->T->h_dir.size() / 6), np = (int)(ctx->T->h_point.size() / 7);
    const int gd = ctx->dirgrid_cells, gp = ctx->pointgrid_cells, sd = std::max(ctx->dirgrid_slabs, 1), sp = std::max(ctx->pointgrid_shells, 1);
    trt::GridView &g = ctx->grids;
    g.enabled = 0;
    ctx->T->grids_built_for[0] = gd;
    ctx->T->grids_built_for[1] = gp;
    ctx->T->grids_built_for[2] = ctx->dirgrid_slabs;
    ctx->T->grids_built_for[3] = ctx->pointgrid_shells;
    if (gd < 8 || gp < 2 || nd + np == 0 || n > TRT_LIST_MAX_SPHERES_WIDE)
        return TRT_OK; // enabled = 0: the kernel sweeps
    const int bits = n > TRT_LIST_MAX_SPHERES ? 16 : 8; // entry width of the list cells
    g.list_bits = bits;
    const size_t words = (size_t)std::max((n + 63) / 64, 1), slots = (size_t)std::max(n, 1);
    const size_t dir_stride = (size_t)sd * gd * gd * words, point_stride = 6 * (size_t)sp * gp * gp * words;
    std::vector<trt_dirgrid> dg(nd);
    std::vector<trt_pointgrid> pg(np);
    std::vector<trt_dirgrid_disc> discs(slots * nd);
    std::vector<trt_pointgrid_cone> cones(slots * np);
    for (int i = 0; i < nd; i++)
    {
        const double *li = ctx->T->h_dir.data() + 6 * i;
        const double to_light[3] = {-li[0], -li[1], -li[2]}; // TRT.c:903; prepare normalises
        const double len2 = to_light[0] * to_light[0] + to_light[1] * to_light[1] + to_light[2] * to_light[2];
        if (!(len2 > 0.0) || !(len2 < 1e300))
            return TRT_OK; // a light without a direction: leave the tables off
        trt_dirgrid_prepare(ctx->T->h_spheres.data(), n, &cs, to_light, gd, sd, &dg[i], discs.data() + slots * i);
    }
    for (int i = 0; i < np; i++)
        trt_pointgrid_prepare(ctx->T->h_spheres.data(), n, &cs, ctx->T->h_point.data() + 7 * i, gp, sp, &pg[i], cones.data() + slots * i);
    HIP_TRY(ctx->T->d_dir_masks.reserve(dir_stride * nd));
    HIP_TRY(ctx->T->d_point_masks.reserve(point_stride * np));
    HIP_TRY(ctx->T->d_dirgrids.reserve(nd));
    HIP_TRY(ctx->T->d_pointgrids.reserve(np));
    HIP_TRY(ctx->T->d_discs.reserve(discs.size()));
    HIP_TRY(ctx->T->d_cones.reserve(cones.size()));
    if (nd)
    {
        HIP_TRY(hipMemcpy(ctx->T->d_dirgrids.ptr, dg.data(), dg.size() * sizeof(trt_dirgrid), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->T->d_discs.ptr, discs.data(), discs.size() * sizeof(trt_dirgrid_disc), hipMemcpyHostToDevice));
    }
    if (np)
    {
        HIP_TRY(hipMemcpy(ctx->T->d_pointgrids.ptr, pg.data(), pg.size() * sizeof(trt_pointgrid), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->T->d_cones.ptr, cones.data(), cones.size() * sizeof(trt_pointgrid_cone), hipMemcpyHostToDevice));
    }
    const int block = 256;
    for (int i = 0; i < nd; i++)
        hipLaunchKernelGGL(build_dirgrid_kernel, dim3((unsigned)(((size_t)sd * gd * gd + block - 1) / block)), dim3(block), 0, ctx->stream,
                           ctx->T->d_discs.ptr + slots * i, n, gd, sd, (int)words, ctx->T->d_dir_masks.ptr + dir_stride * i);
    for (int i = 0; i < np; i++)
        hipLaunchKernelGGL(build_pointgrid_kernel, dim3((unsigned)((6 * (size_t)sp * gp * gp + block - 1) / block)), dim3(block), 0, ctx->stream,
                           ctx->T->d_cones.ptr + slots * i, n, gp, sp, (int)words, ctx->T->d_point_masks.ptr + point_stride * i);
    // the kernel reads list cells: pack every table (the mask words stay for trt_read_light_grid)
    const size_t dir_cells = (size_t)sd * gd * gd, point_cells = 6 * (size_t)sp * gp * gp;
    HIP_TRY(ctx->T->d_dir_lists.reserve(dir_cells * nd));
    HIP_TRY(ctx->T->d_point_lists.reserve(point_cells * np));
    if (nd)
        hipLaunchKernelGGL(pack_lists_kernel, dim3((unsigned)((dir_cells * nd + block - 1) / block)), dim3(block), 0, ctx->stream, ctx->T->d_dir_masks.ptr,
                           (long)(dir_cells * nd), (int)words, ctx->T->d_dir_lists.ptr, ctx->T->d_pool.ptr, ctx->T->d_pool_used.ptr, (unsigned)ctx->T->pool_scene_words, bits);
    if (np)
        hipLaunchKernelGGL(pack_lists_kernel, dim3((unsigned)((point_cells * np + block - 1) / block)), dim3(block), 0, ctx->stream, ctx->T->d_point_masks.ptr,
                           (long)(point_cells * np), (int)words, ctx->T->d_point_lists.ptr, ctx->T->d_pool.ptr, ctx->T->d_pool_used.ptr, (unsigned)ctx->T->pool_scene_words, bits);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // the caller may hand the context another stream before it renders
    g.dir = ctx->T->d_dirgrids.ptr;
    g.point = ctx->T->d_pointgrids.ptr;
    g.dir_lists = ctx->T->d_dir_lists.ptr;
    g.point_lists = ctx->T->d_point_lists.ptr;
    g.dir_stride = (unsigned)dir_cells;
    g.point_stride = (unsigned)point_cells;
    g.pool = ctx->T->d_pool.ptr;
    g.enabled = 1;
    return TRT_OK;
}

// m of the spheres' sub-families (trt_raygrid.h) for a scene of n spheres
// The automatic policy (path_patches = -1) also looks at what the tables would weigh: 6 m^2 tables per sphere and side -- 604 MB of
// cells and a 302 MB pool at 256 spheres, 32 cells, m = 2 -- and steps m down (2 -> 1 -> 0) until cells and pool fit a budget
// instead of failing in hipMalloc or on the 2^32-cell limit; an m asked for by number is taken as it is.
constexpr unsigned long long kAutoPatchBudgetBytes = 4ull << 30;
int patches_for(const trt_context *ctx, int n)
{
    if (ctx->path_patches >= 0)
        return std::min(ctx->path_patches, TRT_PATCH_MAX_M);
    if (ctx->moving_scene) // tables that live for one frame: the 24-fold cells of the patches cost more to build than they save
        return 0;
    int m = n >= TRT_PATCHES_FROM_SPHERES ? 2 : 0;
    const unsigned long long per_table = 6ull * (unsigned long long)ctx->path_g_sph * (unsigned long long)ctx->path_g_sph;
    while (m > 0 && 2ull * (unsigned long long)n * (6ull * m * m) * per_table * 12ull > kAutoPatchBudgetBytes) // 8 B a cell + half a pool word
        m--;
    return m;
}

// Direction tables of the 2NP families of the spheres of the path rays (trt_raygrid.h: P patches per sphere and their mirror
// images); the two families of the eye follow per camera (ensure_eye_tables).  The host places the families (O(NP)), the
// device forms the cones and marks and packs the cells.
int build_path_tables(trt_context *ctx, const trt_cull_scene &cs, const double *ground)
{
    const int n = (int)(ctx->T->h_spheres.size() / 9);
    trt::GridView &g = ctx->grids;
    g.path_enabled = 0;
    g.patch_m = g.patch_count = 0;
    ctx->eye_tables_valid = false;
    ctx->T->path_built_for[0] = ctx->path_g_eye;
    ctx->T->path_built_for[1] = ctx->path_g_sph;
    ctx->T->path_built_for[2] = ctx->path_min_spheres;
    ctx->T->path_built_for[3] = ctx->path_patches;
    ctx->T->cull_scene = cs;
    memcpy(ctx->T->ground_built, ground, sizeof ctx->T->ground_built);
    const int ge = ctx->path_g_eye, gs = ctx->path_g_sph;
    if (ge < 2 || gs < 2 || n > TRT_PATH_MAX_SPHERES || n < ctx->path_min_spheres)
        return TRT_OK; // path_enabled = 0: every path ray sweeps
    trt_patchset patches;
    trt_patchset_init(&patches, patches_for(ctx, n));
    const size_t P = (size_t)patches.count, families = 2 * (size_t)n * P;
    const size_t eye_cells = 6 * (size_t)ge * ge, sph_cells = 6 * (size_t)gs * gs;
    const size_t eye_part = (size_t)kEyeSlots * 2 * eye_cells; // the eye's two tables of every slot first
    if (eye_part + families * sph_cells >= 0xffffffffull)
        return fail(TRT_ERR_CAPACITY, "path tables of %zu families x %zu cells", families, sph_cells);
    HIP_TRY(ctx->T->d_path_lists.reserve(eye_part + families * sph_cells));
    HIP_TRY(ctx->T->d_families.reserve(std::max<size_t>(families, 1)));
    HIP_TRY(ctx->T->d_sphere_fam.reserve(4 * (size_t)std::max(n, 1)));
    HIP_TRY(ctx->T->d_patch_rec.reserve(P * TRT_PATCH_RECORD));
    std::vector<trt_rayfamily> fam(std::max<size_t>(families, 1));
    std::vector<double> rec(4 * (size_t)std::max(n, 1)), prec(P * TRT_PATCH_RECORD);
    trt_family_consts consts;
    trt_sphere_families(ctx->T->h_spheres.data(), n, ground, &cs, &patches, fam.data(), rec.data(), &consts);
    trt_patch_records(&patches, ground, prec.data());
    HIP_TRY(hipMemcpy(ctx->T->d_patch_rec.ptr, prec.data(), prec.size() * sizeof(double), hipMemcpyHostToDevice));
    if (n)
    {
        HIP_TRY(hipMemcpy(ctx->T->d_families.ptr, fam.data(), families * sizeof(trt_rayfamily), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->T->d_sphere_fam.ptr, rec.data(), 4 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        for (size_t first = 0; first < families; first += 32768) // grid.y is limited to 65535
        {
            const unsigned batch = (unsigned)std::min<size_t>(32768, families - first);
            launch_family_builder(n, dim3(family_grid_blocks(gs, kSceneTilesPerBlock), batch), ctx->stream, (const double *)ctx->T->d_spheres.ptr,
                                  (const trt_rayfamily *)ctx->T->d_families.ptr + first, trt_rayfamily{}, trt_rayfamily{}, 0, gs,
                                  ctx->T->d_path_lists.ptr + eye_part + first * sph_cells, ctx->T->d_pool.ptr, ctx->T->d_pool_used.ptr, (unsigned)ctx->T->pool_scene_words,
                                  kSceneTilesPerBlock);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    g.path_lists = ctx->T->d_path_lists.ptr;
    g.eye_at = (unsigned)((size_t)ctx->eye_slot * 2 * eye_cells);
    g.sph_at = (unsigned)eye_part;
    g.pool = ctx->T->d_pool.ptr;
    g.sphere_fam = ctx->T->d_sphere_fam.ptr;
    g.patch_rec = ctx->T->d_patch_rec.ptr;
    g.patch_m = patches.m;
    g.patch_count = patches.count;
    g.rg2_sph = consts.rg * consts.rg;
    g.slack0 = consts.slack;
    g.g_eye = ge;
    g.g_sph = gs;
    g.path_enabled = 1;
    return TRT_OK;
}

// Every candidate table of the scene: the pool of long lists is laid out first (one part for the scene's tables, one for
// the eye's, which are rebuilt per camera), then the light tables, then the sphere families.
int build_tables(trt_context *ctx, const trt_cull_scene &cs, const double *ground)
{
    const size_t n = ctx->T->h_spheres.size() / 9, nd = ctx->T->h_dir.size() / 6, np = ctx->T->h_point.size() / 7;
    const size_t gd = (size_t)ctx->dirgrid_cells, gp = (size_t)ctx->pointgrid_cells, ge = (size_t)ctx->path_g_eye, gs = (size_t)ctx->path_g_sph;
    const size_t sd = (size_t)std::max(ctx->dirgrid_slabs, 1), sp = (size_t)std::max(ctx->pointgrid_shells, 1);
    // one pool word per cell; the many small tables of sub-families (their lists are short: that is what they are for) get
    // half a word per cell -- a list that finds no room leaves its cell TRT_LIST_NONE and its rays sweep
    const int m = patches_for(ctx, (int)n);
    const size_t sphere_cells = 2 * n * (m ? 6 * (size_t)m * m : 1) * 6 * gs * gs;
    // (scenes of more than 256 spheres: 16-bit entries, a long list takes twice the words; beyond TRT_PATH_MAX_SPHERES no sphere families)
    const size_t wide = n > TRT_LIST_MAX_SPHERES ? 2 : 1, sphere_part = n > TRT_PATH_MAX_SPHERES ? 0 : (m ? sphere_cells / 2 : sphere_cells);
    ctx->T->pool_scene_words = std::max<size_t>(1024, wide * (nd * sd * gd * gd + np * 6 * sp * gp * gp + sphere_part));
    // The eye's two tables are rebuilt for every camera, on the frame's stream: nobody can look at their counter and grow their part
    // afterwards, so it holds the longest lists there can be -- every sphere in every cell, up to 64 words (512 / 256 entries) a cell.
    // (Round 4 gave them one word per cell: enough at 64 and 256 spheres, not in a scene of 700, whose primary rays then swept.)
    const size_t per_word = wide == 2 ? 4 : 8;
    ctx->T->pool_eye_words = std::max<size_t>(1024, 2 * 6 * ge * ge * std::min<size_t>((n + per_word - 1) / per_word, 64));
    if (ctx->list_pool_cap) // trt_set_list_pool_words (tests: the pool's exhaustion)
        ctx->T->pool_scene_words = std::min(ctx->T->pool_scene_words, std::max<size_t>(ctx->list_pool_cap, 1));
    const double t0 = host_seconds();
    int rc = TRT_OK;
    // The scene's part is sized by a guess (a word per cell) and GROWN to what the builders asked for if that was more: the
    // counter keeps counting after the part is exhausted (pack_cell), so one more pass with a part of that size has room for
    // every list.  Dense scenes of many spheres need it (700 spheres: ~7 words a cell); the BASELINE configs never do.
    for (int pass = 0; pass < 2 && !rc; pass++)
    {
        if (ctx->T->pool_scene_words + kEyeSlots * ctx->T->pool_eye_words >= 0xffffffffull)
            return fail(TRT_ERR_CAPACITY, "candidate tables too large");
        ctx->grids = trt::GridView{};
        ctx->grids.list_bits = n > TRT_LIST_MAX_SPHERES ? 16 : 8; // every table of the scene: the light tables' and the families' lists alike
        HIP_TRY(ctx->T->d_pool.reserve(ctx->T->pool_scene_words + kEyeSlots * ctx->T->pool_eye_words));
        HIP_TRY(ctx->T->d_pool_used.reserve(16 * (1 + kEyeSlots)));
        HIP_TRY(hipMemsetAsync(ctx->T->d_pool_used.ptr, 0, 16 * (1 + kEyeSlots) * sizeof(unsigned long long), ctx->stream));
        rc = build_light_grids(ctx, cs);
        if (!rc)
            rc = build_path_tables(ctx, cs, ground);
        if (rc || ctx->list_pool_cap)
            break; // a capped pool (tests) stays capped
        unsigned long long asked = 0;
        HIP_TRY(hipMemcpy(&asked, ctx->T->d_pool_used.ptr, sizeof asked, hipMemcpyDeviceToHost)); // the builders have been synchronised
        if (asked <= ctx->T->pool_scene_words)
            break;
        ctx->T->pool_scene_words = (size_t)asked + 1024;
    }
    ctx->T->build_seconds = host_seconds() - t0;
    return rc;
}

// The two families of the eye (trt_raygrid.h): rebuilt on `stream` whenever the eye (or the scene) changed since they were built.
int ensure_eye_tables(trt_context *ctx, const Camera *camera, hipStream_t stream)
{
    trt::GridView &g = ctx->grids;
    if (!g.path_enabled)
        return TRT_OK;
    const double eye[3] = {camera->frame.origin.x, camera->frame.origin.y, camera->frame.origin.z};
    if (ctx->eye_tables_valid && !memcmp(eye, ctx->eye_built, sizeof eye))
        return TRT_OK;
    trt_eye_families(eye, ctx->T->ground_built, &ctx->T->cull_scene, g.eye);
    const int n = (int)(ctx->T->h_spheres.size() / 9), ge = g.g_eye;
    const size_t eye_cells = 6 * (size_t)ge * ge;
    // this context's part of the pool: behind the scene's part and the parts of the slots before it
    const size_t pool_from = ctx->T->pool_scene_words + (size_t)ctx->eye_slot * ctx->T->pool_eye_words;
    unsigned long long *const counter = ctx->T->d_pool_used.ptr + 16 * (1 + ctx->eye_slot);
    hipLaunchKernelGGL(set_pool_counter_kernel, dim3(1), dim3(1), 0, stream, counter, (unsigned long long)pool_from);
    launch_family_builder(n, dim3(family_grid_blocks(ge, 1), 2u), stream, (const double *)ctx->T->d_spheres.ptr, (const trt_rayfamily *)nullptr, g.eye[0], g.eye[1], 1,
                          ge, ctx->T->d_path_lists.ptr + g.eye_at, ctx->T->d_pool.ptr, counter, (unsigned)(pool_from + ctx->T->pool_eye_words), 1);
    HIP_TRY(hipGetLastError());
    memcpy(ctx->eye_built, eye, sizeof eye);
    ctx->eye_tables_valid = true;
    return TRT_OK;
}

// The production kernel's occupancy depends on the scene and its tables only through the size of the LDS image: queried once
// per size, not once per frame.
int refresh_occupancy(trt_context *ctx)
{
    const trt::SceneView &v = ctx->scene;
    const size_t lds_need = std::max(scene_lds_bytes(v), image_lds_bytes(ctx, 64));
    if (lds_need > (size_t)ctx->lds_limit)
        return fail(TRT_ERR_CAPACITY, "scene needs %zu B of LDS staging, device offers %d", lds_need, ctx->lds_limit);
    if (ctx->occupancy_for_lds != image_lds_bytes(ctx, 64))
    {
        int blocks = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, trt::render_rounds_kernel<false>, trt::kPersistentBlock, image_lds_bytes(ctx, 64)));
        ctx->rounds_blocks_per_cu = std::max(blocks, 1);
        ctx->occupancy_for_lds = image_lds_bytes(ctx, 64);
        ctx->compact_blocks_per_cu = 0;
        if (compact_lds_bytes(ctx, 64) <= (size_t)ctx->lds_limit)
        {
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, trt::render_rounds_kernel<false, false, true>, trt::kCompactBlock,
                                                                 compact_lds_bytes(ctx, 64)));
            ctx->compact_blocks_per_cu = blocks;
        }
    }
    return TRT_OK;
}

// trt_set_scene_policy: a scene counts as moving from this many consecutive changed calls on, and as still again after this many unchanged ones
int g_moving_after = 2, g_still_after = 3;

// everything of the scene except camera and skybox.  per_call: the drop-in entries, which are handed the scene with every frame
int upload_primitives(trt_context *ctx, const Scene *scene, bool per_call = false)
{
    const int n = scene->num_spheres, nd = scene->num_directional_lights, np = scene->num_point_lights;
    if (n < 0 || nd < 0 || np < 0)
        return fail(TRT_ERR_ARGUMENT, "negative primitive count");
    if ((n && !scene->spheres) || (nd && !scene->directional_lights) || (np && !scene->point_lights))
        return fail(TRT_ERR_ARGUMENT, "NULL primitive array with a non-zero count");
    HIP_TRY(ctx->T->d_spheres.reserve((size_t)n * 9));
    HIP_TRY(ctx->T->d_dir.reserve((size_t)nd * 6));
    HIP_TRY(ctx->T->d_point.reserve((size_t)np * 7));
    if (n)
        HIP_TRY(hipMemcpy(ctx->T->d_spheres.ptr, scene->spheres, (size_t)n * sizeof(Sphere), hipMemcpyHostToDevice));
    if (nd)
        HIP_TRY(hipMemcpy(ctx->T->d_dir.ptr, scene->directional_lights, (size_t)nd * sizeof(DirectionalLight), hipMemcpyHostToDevice));
    if (np)
        HIP_TRY(hipMemcpy(ctx->T->d_point.ptr, scene->point_lights, (size_t)np * sizeof(PointLight), hipMemcpyHostToDevice));

    // FP32 culling table {Cx,Cy,Cz,kk} of trt_filter.h (filter only, never decides a result)
    const int padded = trt_cull_padded(n, trt::kCullGroup);
    std::vector<float> cull((size_t)padded * 4);
    trt_cull_scene cs;
    trt_cull_build((const double *)scene->spheres, n, trt::kCullGroup, cull.data(), &cs);
    HIP_TRY(ctx->T->d_cull.reserve(cull.size()));
    if (padded)
        HIP_TRY(hipMemcpy(ctx->T->d_cull.ptr, cull.data(), cull.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->cull.table = ctx->T->d_cull.ptr;
    ctx->cull.padded = padded;
    ctx->cull.c0x = cs.c0[0];
    ctx->cull.c0y = cs.c0[1];
    ctx->cull.c0z = cs.c0[2];
    ctx->cull.cn = cs.cn;
    ctx->cull.rm = cs.rm;

    // the light-space tables only change with the spheres and the lights (a render loop usually moves the camera only)
    const double *hs = (const double *)scene->spheres, *hd = (const double *)scene->directional_lights, *hp = (const double *)scene->point_lights;
    const bool same_primitives = ctx->T->h_spheres.size() == (size_t)n * 9 && ctx->T->h_dir.size() == (size_t)nd * 6 && ctx->T->h_point.size() == (size_t)np * 7 &&
                                 (!n || !memcmp(ctx->T->h_spheres.data(), hs, (size_t)n * sizeof(Sphere))) &&
                                 (!nd || !memcmp(ctx->T->h_dir.data(), hd, (size_t)nd * sizeof(DirectionalLight))) &&
                                 (!np || !memcmp(ctx->T->h_point.data(), hp, (size_t)np * sizeof(PointLight))) &&
                                 !memcmp(ctx->T->ground_built, &scene->ground, sizeof ctx->T->ground_built);
    if (!per_call) // trt_set_scene: "once per scene" -- always the full tables
        ctx->moving_scene = false, ctx->scene_changes_in_a_row = ctx->scene_still_calls = 0;
    else if (!same_primitives)
    {
        ctx->scene_still_calls = 0;
        if (++ctx->scene_changes_in_a_row >= g_moving_after && g_moving_after > 0)
            ctx->moving_scene = true;
    }
    else
    {
        ctx->scene_changes_in_a_row = 0;
        if (++ctx->scene_still_calls >= g_still_after)
            ctx->moving_scene = false; // still again: the tables below are promoted to the full ones
    }
    const bool same = same_primitives && ctx->T->built_for_moving_scene == ctx->moving_scene &&
                      ctx->T->grids_built_for[0] == ctx->dirgrid_cells && ctx->T->grids_built_for[1] == ctx->pointgrid_cells &&
                      ctx->T->grids_built_for[2] == ctx->dirgrid_slabs && ctx->T->grids_built_for[3] == ctx->pointgrid_shells &&
                      ctx->T->path_built_for[0] == ctx->path_g_eye && ctx->T->path_built_for[1] == ctx->path_g_sph &&
                      ctx->T->path_built_for[2] == ctx->path_min_spheres && ctx->T->path_built_for[3] == ctx->path_patches &&
                      !memcmp(ctx->T->ground_built, &scene->ground, sizeof ctx->T->ground_built);
    if (!same)
    {
        ctx->T->h_spheres.assign(hs, hs + (size_t)n * 9);
        ctx->T->h_dir.assign(hd, hd + (size_t)nd * 6);
        ctx->T->h_point.assign(hp, hp + (size_t)np * 7);
        ctx->T->built_for_moving_scene = ctx->moving_scene;
        const int rc = build_tables(ctx, cs, (const double *)&scene->ground);
        if (rc)
        {
            ctx->T->grids_built_for[0] = ctx->T->grids_built_for[1] = ctx->T->grids_built_for[2] = ctx->T->grids_built_for[3] = -1;
            return rc;
        }
    }

    trt::SceneView &v = ctx->scene;
    v.spheres = ctx->T->d_spheres.ptr;
    v.dir_lights = ctx->T->d_dir.ptr;
    v.point_lights = ctx->T->d_point.ptr;
    v.num_spheres = n;
    v.num_dir = nd;
    v.num_point = np;
    memcpy(v.ground, &scene->ground, sizeof(Plane));
    return refresh_occupancy(ctx);
}

int prepare_jitter(trt_context *ctx, const Camera *cam, int width, int height, int spp)
{
    // TRT.c:981-982, :992-993: triangle_wave(2*PI*k/spp)/2*pixel_width and triangle_wave(PI*k/spp)/2*pixel_height
    const double pw = cam->screen_width / width, ph = cam->screen_height / height;
    if (ctx->jit_spp == spp && ctx->jit_pw == pw && ctx->jit_ph == ph)
        return TRT_OK;
    std::vector<double> j(2 * (size_t)spp);
    for (int k = 0; k < spp; k++)
    {
        j[k] = triangle_wave(2 * kPi * k / spp) / 2 * pw;
        j[spp + k] = triangle_wave(kPi * k / spp) / 2 * ph;
    }
    HIP_TRY(ctx->d_jitter.reserve(j.size()));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // a frame in flight may still read the old table
    HIP_TRY(hipMemcpy(ctx->d_jitter.ptr, j.data(), j.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->jit_spp = spp;
    ctx->jit_pw = pw;
    ctx->jit_ph = ph;
    return TRT_OK;
}

// TRT.c:987-988 without the jitter: one value per column and one per frame row, formed on the host in the
// reference's operation order (this file is compiled with -ffp-contract=off for host and device alike)
int prepare_axes(trt_context *ctx, const Camera *cam, int width, int height)
{
    const double sw = cam->screen_width, sh = cam->screen_height;
    if (ctx->axes_w == width && ctx->axes_h == height && ctx->axes_sw == sw && ctx->axes_sh == sh)
        return TRT_OK;
    std::vector<double> t((size_t)width + height);
    for (int column = 0; column < width; column++)
        t[column] = (((double)column / (double)width) * sw - sw / 2.0);
    for (int row = 0; row < height; row++)
        t[(size_t)width + row] = -(((double)row / (double)height) * sh - sh / 2.0);
    HIP_TRY(ctx->d_axes.reserve(t.size()));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(ctx->d_axes.ptr, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->axes_w = width;
    ctx->axes_h = height;
    ctx->axes_sw = sw;
    ctx->axes_sh = sh;
    return TRT_OK;
}

bool rowset_valid(const trt_rowset *r)
{
    return r && r->width > 0 && r->height > 0 && r->tile_rows > 0 && r->tile_first >= 0 && r->tile_step > 0;
}

} // namespace

extern "C" int trt_rowset_rows(const trt_rowset *r)
{
    if (!rowset_valid(r))
        return 0;
    const int tiles = (r->height + r->tile_rows - 1) / r->tile_rows;
    long rows = 0;
    for (int t = r->tile_first; t < tiles; t += r->tile_step)
        rows += std::min(r->tile_rows, r->height - t * r->tile_rows);
    return (int)rows;
}

extern "C" int trt_rowset_frame_row(const trt_rowset *r, int local_row)
{
    if (!rowset_valid(r) || local_row < 0 || local_row >= trt_rowset_rows(r))
        return -1;
    const int t = local_row / r->tile_rows;
    return (r->tile_first + t * r->tile_step) * r->tile_rows + (local_row - t * r->tile_rows);
}

extern "C" const char *trt_last_error(void) { return g_error; }
extern "C" const char *trt_version(void) { return "trt-mi355x 0.1 (gfx950, fp64, contraction off)"; }

static int init_context(trt_context *ctx);

extern "C" int trt_create(int device, trt_context **out)
{
    if (!out)
        return fail(TRT_ERR_ARGUMENT, "out is NULL");
    *out = nullptr;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count)
        return fail(TRT_ERR_ARGUMENT, "device %d out of range (%d visible)", device, count);
    HIP_TRY(hipSetDevice(device));
    trt_context *ctx = new trt_context();
    ctx->device = device;
    const int rc = init_context(ctx);
    if (rc)
    { // hand nothing half-built to the caller, keep nothing behind
        (void)trt_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return TRT_OK;
}

static int init_context(trt_context *ctx)
{
    const int device = ctx->device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    ctx->compute_units = prop.multiProcessorCount;
    ctx->lds_limit = (int)prop.sharedMemPerBlock;
    ctx->T = std::make_shared<SceneTables>();
    ctx->T->device = device;
    ctx->T->eye_slots_taken = 1u;
    if (const char *e = getenv("TRT_LIGHTGRID"))
    {
        int gd = 0, gp = 0, sd = TRT_DIRGRID_SLABS, sp = TRT_POINTGRID_SHELLS;
        if (sscanf(e, "%d,%d,%d,%d", &gd, &gp, &sd, &sp) >= 2 && gd >= 0 && gp >= 0 && gd <= 2048 && gp <= 1024 && sd >= 1 && sp >= 1 && sd <= 64 && sp <= 64)
            ctx->dirgrid_cells = gd, ctx->pointgrid_cells = gp, ctx->dirgrid_slabs = sd, ctx->pointgrid_shells = sp;
    }
    if (const char *e = getenv("TRT_PATHGRID"))
    {
        int ge = 0, gs = 0, m = TRT_PATHGRID_PATCHES;
        if (sscanf(e, "%d,%d,%d", &ge, &gs, &m) >= 2 && ge >= 0 && gs >= 0 && ge <= 1024 && gs <= 256 && m >= -1 && m <= TRT_PATCH_MAX_M)
            ctx->path_g_eye = ge, ctx->path_g_sph = gs, ctx->path_patches = m;
    }
    if (const char *e = getenv("TRT_COMPACTION"))
    {
        int mode = -1;
        if (sscanf(e, "%d", &mode) == 1 && mode >= -1 && mode <= 1)
            ctx->compaction = mode;
    }
    HIP_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    for (int i = 0; i < kEventRing; i++)
    {
        HIP_TRY(hipEventCreate(&ctx->ev_start[i]));
        HIP_TRY(hipEventCreate(&ctx->ev_mid[i]));
        HIP_TRY(hipEventCreate(&ctx->ev_stop[i]));
    }
    for (int i = 0; i < 16; i++)
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_chunk[i], hipEventDisableTiming));
    for (int i = 0; i < 8; i++)
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_band[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIP_TRY(ctx->d_counters.reserve(kCounterSlots));
    HIP_TRY(ctx->d_queue.reserve(64));
    HIP_TRY(hipMemset(ctx->d_counters.ptr, 0, kCounterSlots * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(ctx->d_queue.ptr, 0, 64 * sizeof(unsigned int)));
    // dynamic LDS above the 64 KiB default needs the opt-in attribute
    (void)hipFuncSetAttribute((const void *)build_family_lists_kernel<TRT_PATH_MAX_SPHERES / 64>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_simple_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::probe_rays_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::probe_rounds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::probe_rounds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    return TRT_OK;
}

extern "C" int trt_destroy(trt_context *ctx)
{
    if (!ctx)
        return TRT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < kEventRing; i++)
    {
        if (ctx->ev_start[i])
            (void)hipEventDestroy(ctx->ev_start[i]);
        if (ctx->ev_mid[i])
            (void)hipEventDestroy(ctx->ev_mid[i]);
        if (ctx->ev_stop[i])
            (void)hipEventDestroy(ctx->ev_stop[i]);
    }
    for (int i = 0; i < 16; i++)
        if (ctx->ev_chunk[i])
            (void)hipEventDestroy(ctx->ev_chunk[i]);
    for (int i = 0; i < 8; i++)
        if (ctx->ev_band[i])
            (void)hipEventDestroy(ctx->ev_band[i]);
    (void)hipGetLastError();
    if (ctx->copy_stream)
        (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->alt_stream)
        (void)hipStreamDestroy(ctx->alt_stream);
    if (ctx->ev_fork)
        (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->T)
        ctx->T->eye_slots_taken &= ~(1u << ctx->eye_slot);
    ctx->T.reset(); // the tables go with their last context
    ctx->d_jitter.release();
    ctx->d_axes.release();
    ctx->d_samples.release();
    ctx->d_samples_alt.release();
    ctx->d_fb.release();
    ctx->d_rgb8.release();
    ctx->d_ior.release();
    ctx->d_counters.release();
    ctx->d_queue.release();
    if (ctx->h_staging)
        (void)hipHostFree(ctx->h_staging);
    if (ctx->own_stream)
        (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return TRT_OK;
}

extern "C" int trt_set_stream(trt_context *ctx, void *hip_stream)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return TRT_OK;
}

extern "C" int trt_reserve_cus(trt_context *ctx, int reserved)
{
    if (!ctx || reserved < 0)
        return fail(TRT_ERR_ARGUMENT, "bad argument");
    if (reserved > ctx->compute_units / 2)
        return fail(TRT_ERR_ARGUMENT, "%d of %d compute units", reserved, ctx->compute_units);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const bool was_own = ctx->stream == ctx->own_stream;
    hipStream_t fresh = nullptr;
    if (reserved == 0)
        HIP_TRY(hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
    else
    { // one bit per CU; the driver deals consecutive bits round the XCDs, so dropping the top bits thins every XCD alike
        std::vector<uint32_t> mask((size_t)(ctx->compute_units + 31) / 32, 0u);
        for (int cu = 0; cu < ctx->compute_units - reserved; cu++)
            mask[(size_t)cu / 32] |= 1u << (cu % 32);
        HIP_TRY(hipExtStreamCreateWithCUMask(&fresh, (uint32_t)mask.size(), mask.data()));
    }
    if (ctx->own_stream)
        (void)hipStreamDestroy(ctx->own_stream);
    ctx->own_stream = fresh;
    if (was_own)
        ctx->stream = fresh;
    ctx->reserved_cus = reserved;
    return TRT_OK;
}

extern "C" int trt_get_stream(trt_context *ctx, void **hip_stream)
{
    if (!ctx || !hip_stream)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    *hip_stream = (void *)ctx->stream;
    return TRT_OK;
}

// a context that shares its tables gets fresh, empty ones of its own (slot 0)
static void detach_tables(trt_context *ctx)
{
    if (ctx->T.use_count() <= 1)
        return;
    ctx->T->eye_slots_taken &= ~(1u << ctx->eye_slot);
    ctx->T = std::make_shared<SceneTables>();
    ctx->T->device = ctx->device;
    ctx->T->eye_slots_taken = 1u;
    ctx->eye_slot = 0;
    ctx->eye_tables_valid = false;
    ctx->grids = trt::GridView{};
    ctx->sky_dim = -1;
}

static int refuse_if_shared(const trt_context *ctx, const char *what)
{
    if (ctx->T.use_count() > 1)
        return fail(TRT_ERR_ARGUMENT, "%s: this context's scene tables are shared with %ld other context(s) (trt_share_scene); "
                                      "change them before sharing, or give the context a scene of its own (trt_set_scene)", what, ctx->T.use_count() - 1);
    return TRT_OK;
}

// dst renders the scene of src from src's tables (same device): nothing is uploaded or built again; only what depends on the
// camera -- the eye's two tables, in a slot of their own -- and the per-frame buffers stay dst's.  Up to kEyeSlots contexts per scene.
extern "C" int trt_share_scene(trt_context *dst, trt_context *src)
{
    if (!dst || !src || dst == src)
        return fail(TRT_ERR_ARGUMENT, "bad argument");
    if (!src->have_scene)
        return fail(TRT_ERR_NO_SCENE, "the source context has no scene");
    if (dst->device != src->device)
        return fail(TRT_ERR_ARGUMENT, "contexts of different devices (%d, %d) cannot share tables", dst->device, src->device);
    HIP_TRY(hipSetDevice(dst->device));
    HIP_TRY(hipStreamSynchronize(dst->stream));
    HIP_TRY(hipStreamSynchronize(src->stream));
    if (dst->T == src->T)
        return TRT_OK;
    int slot = -1;
    for (int k = 0; k < kEyeSlots && slot < 0; k++)
        if (!(src->T->eye_slots_taken & (1u << k)))
            slot = k;
    if (slot < 0)
        return fail(TRT_ERR_CAPACITY, "%d contexts share these tables already", kEyeSlots);
    dst->T->eye_slots_taken &= ~(1u << dst->eye_slot);
    dst->T = src->T;
    dst->T->eye_slots_taken |= 1u << slot;
    dst->eye_slot = slot;
    dst->scene = src->scene;
    dst->cull = src->cull;
    dst->grids = src->grids;
    dst->grids.eye_at = (unsigned)((size_t)slot * 2 * 6 * (size_t)src->grids.g_eye * (size_t)src->grids.g_eye);
    dst->eye_tables_valid = false;
    // the settings the tables were built with travel along (a later trt_set_scene on dst then builds alike)
    dst->dirgrid_cells = src->dirgrid_cells, dst->pointgrid_cells = src->pointgrid_cells;
    dst->dirgrid_slabs = src->dirgrid_slabs, dst->pointgrid_shells = src->pointgrid_shells;
    dst->path_g_eye = src->path_g_eye, dst->path_g_sph = src->path_g_sph, dst->path_min_spheres = src->path_min_spheres;
    dst->path_patches = src->path_patches, dst->list_pool_cap = src->list_pool_cap;
    memcpy(dst->sky_faces, src->sky_faces, sizeof dst->sky_faces);
    dst->sky_dim = src->sky_dim;
    dst->sky_stamp = src->sky_stamp;
    dst->ior_count = 0;
    dst->have_scene = true;
    dst->occupancy_for_lds = (size_t)-1;
    return refresh_occupancy(dst);
}

// {bytes of device memory held by the scene's tables and primitives, contexts sharing them, host seconds of the last table build}
extern "C" int trt_scene_info(trt_context *ctx, unsigned long long *table_bytes, int *sharers, double *build_seconds)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    const SceneTables &t = *ctx->T;
    if (table_bytes)
        *table_bytes = t.d_spheres.capacity * 8 + t.d_dir.capacity * 8 + t.d_point.capacity * 8 + t.d_cull.capacity * 4 + t.d_dir_masks.capacity * 8 +
                       t.d_point_masks.capacity * 8 + t.d_dirgrids.capacity * sizeof(trt_dirgrid) + t.d_pointgrids.capacity * sizeof(trt_pointgrid) +
                       t.d_discs.capacity * sizeof(trt_dirgrid_disc) + t.d_cones.capacity * sizeof(trt_pointgrid_cone) + t.d_dir_lists.capacity * 8 +
                       t.d_point_lists.capacity * 8 + t.d_path_lists.capacity * 8 + t.d_pool.capacity * 8 + t.d_pool_used.capacity * 8 +
                       t.d_families.capacity * sizeof(trt_rayfamily) + t.d_sphere_fam.capacity * 8 + t.d_patch_rec.capacity * 8 + t.d_sky.capacity * 4;
    if (sharers)
        *sharers = (int)ctx->T.use_count();
    if (build_seconds)
        *build_seconds = t.build_seconds;
    return TRT_OK;
}

extern "C" int trt_set_list_pool_words(trt_context *ctx, size_t words)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    const int rc = refuse_if_shared(ctx, "trt_set_list_pool_words");
    if (rc)
        return rc;
    ctx->list_pool_cap = words;
    ctx->T->grids_built_for[0] = -1; // the next trt_set_scene / table setter builds again
    return TRT_OK;
}

extern "C" int trt_set_scene(trt_context *ctx, const Scene *scene)
{
    if (!ctx || !scene)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->have_scene = false;
    detach_tables(ctx); // tables shared with other contexts (trt_share_scene) stay theirs: this context builds its own
    int rc = upload_primitives(ctx, scene);
    if (rc)
        return rc;
    rc = upload_skybox(ctx, &scene->skybox);
    if (rc)
        return rc;
    ctx->have_scene = true;
    return TRT_OK;
}

extern "C" int trt_read_diagnostics(trt_context *ctx, unsigned long long *wave_loop_trips, unsigned long long *phase2_rounds)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    if (wave_loop_trips)
        *wave_loop_trips = ctx->last_trips;
    if (phase2_rounds)
        *phase2_rounds = ctx->last_phase2;
    return TRT_OK;
}

extern "C" int trt_set_refraction(trt_context *ctx, const double *ior, int count)
{
    if (!ctx || count < 0 || (count > 0 && !ior))
        return fail(TRT_ERR_ARGUMENT, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->ior_count = 0;
    if (count == 0)
        return TRT_OK;
    for (int i = 0; i < count; i++)
        if (!(ior[i] >= 0.0) || !(ior[i] < 1e6))
            return fail(TRT_ERR_ARGUMENT, "index of refraction %g of sphere %d", ior[i], i);
    HIP_TRY(ctx->d_ior.reserve((size_t)count));
    HIP_TRY(hipMemcpy(ctx->d_ior.ptr, ior, (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    ctx->ior_count = count;
    return TRT_OK;
}

extern "C" int trt_read_sweep_fallbacks(trt_context *ctx, unsigned long long *swept_traces)
{
    if (!ctx || !swept_traces)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    *swept_traces = ctx->last_swept;
    return TRT_OK;
}

extern "C" int trt_render_variant(trt_context *ctx, int *decoupled, int *workgroup_threads)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    // the variant the most recent launch ran; before the first launch, what a whole large frame would run
    const bool d = ctx->have_scene && (ctx->last_units > 0 ? ctx->last_compact : renders_decoupled(ctx, kCompactionMinUnits));
    if (decoupled)
        *decoupled = d ? 1 : 0;
    if (workgroup_threads)
        *workgroup_threads = ctx->kernel == 1 ? 256 : (d ? trt::kCompactBlock : trt::kPersistentBlock);
    return TRT_OK;
}

extern "C" int trt_read_loop_diagnostics(trt_context *ctx, unsigned long long out[8])
{
    if (!ctx || !out)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    memcpy(out, ctx->last_loops, sizeof ctx->last_loops);
    return TRT_OK;
}

extern "C" int trt_read_shading_passes(trt_context *ctx, unsigned long long *passes)
{
    if (!ctx || !passes)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    *passes = ctx->last_passes;
    return TRT_OK;
}

extern "C" int trt_set_kernel(trt_context *ctx, int which)
{
    if (!ctx || which < 0 || which > 1)
        return fail(TRT_ERR_ARGUMENT, "kernel %d", which);
    ctx->kernel = which;
    return TRT_OK;
}

extern "C" int trt_set_compaction(trt_context *ctx, int mode)
{
    if (!ctx || mode < -1 || mode > 1)
        return fail(TRT_ERR_ARGUMENT, "compaction mode %d", mode);
    ctx->compaction = mode;
    return TRT_OK;
}

extern "C" int trt_set_light_grids(trt_context *ctx, int directional_cells, int point_cells)
{
    if (!ctx || directional_cells < 0 || point_cells < 0 || directional_cells > 2048 || point_cells > 1024)
        return fail(TRT_ERR_ARGUMENT, "light grids %d, %d", directional_cells, point_cells);
    if (const int shared = refuse_if_shared(ctx, "trt_set_light_grids"))
        return shared;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // a frame in flight may still read the old tables
    ctx->dirgrid_cells = directional_cells;
    ctx->pointgrid_cells = point_cells;
    if (!ctx->have_scene)
        return TRT_OK;
    const int n = (int)(ctx->T->h_spheres.size() / 9);
    std::vector<float> table((size_t)trt_cull_padded(n, trt::kCullGroup) * 4 + 4);
    trt_cull_scene cs;
    trt_cull_build(ctx->T->h_spheres.data(), n, trt::kCullGroup, table.data(), &cs);
    const int rc = build_tables(ctx, cs, ctx->scene.ground);
    return rc ? rc : refresh_occupancy(ctx);
}

extern "C" int trt_set_light_slabs(trt_context *ctx, int directional_slabs, int point_shells)
{
    if (!ctx || directional_slabs < 1 || point_shells < 1 || directional_slabs > 64 || point_shells > 64)
        return fail(TRT_ERR_ARGUMENT, "light slabs %d, %d", directional_slabs, point_shells);
    if (const int shared = refuse_if_shared(ctx, "trt_set_light_slabs"))
        return shared;
    ctx->dirgrid_slabs = directional_slabs;
    ctx->pointgrid_shells = point_shells;
    return trt_set_light_grids(ctx, ctx->dirgrid_cells, ctx->pointgrid_cells);
}

extern "C" int trt_set_path_grids(trt_context *ctx, int eye_cells, int sphere_cells)
{
    if (!ctx || eye_cells < 0 || sphere_cells < 0 || eye_cells > 1024 || sphere_cells > 256)
        return fail(TRT_ERR_ARGUMENT, "path grids %d, %d", eye_cells, sphere_cells);
    if (const int shared = refuse_if_shared(ctx, "trt_set_path_grids"))
        return shared;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // a frame in flight may still read the old tables
    ctx->path_g_eye = eye_cells;
    ctx->path_g_sph = sphere_cells;
    if (!ctx->have_scene)
        return TRT_OK;
    const int n = (int)(ctx->T->h_spheres.size() / 9);
    std::vector<float> table((size_t)trt_cull_padded(n, trt::kCullGroup) * 4 + 4);
    trt_cull_scene cs;
    trt_cull_build(ctx->T->h_spheres.data(), n, trt::kCullGroup, table.data(), &cs);
    const int rc = build_tables(ctx, cs, ctx->scene.ground);
    return rc ? rc : refresh_occupancy(ctx);
}

extern "C" int trt_set_path_patches(trt_context *ctx, int m)
{
    if (!ctx || m < -1 || m > TRT_PATCH_MAX_M)
        return fail(TRT_ERR_ARGUMENT, "patches %d", m);
    if (const int shared = refuse_if_shared(ctx, "trt_set_path_patches"))
        return shared;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // a frame in flight may still read the old tables
    ctx->path_patches = m;
    if (!ctx->have_scene)
        return TRT_OK;
    const int n = (int)(ctx->T->h_spheres.size() / 9);
    std::vector<float> table((size_t)trt_cull_padded(n, trt::kCullGroup) * 4 + 4);
    trt_cull_scene cs;
    trt_cull_build(ctx->T->h_spheres.data(), n, trt::kCullGroup, table.data(), &cs);
    const int rc = build_tables(ctx, cs, ctx->scene.ground);
    return rc ? rc : refresh_occupancy(ctx);
}

extern "C" int trt_get_path_patches(trt_context *ctx, int *m, int *patches_per_sphere)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    if (m)
        *m = ctx->grids.path_enabled ? ctx->grids.patch_m : 0;
    if (patches_per_sphere)
        *patches_per_sphere = ctx->grids.path_enabled ? ctx->grids.patch_count : 0;
    return TRT_OK;
}

extern "C" int trt_path_family_code(trt_context *ctx, int kind, int sphere, const double *parent_origin)
{
    if (!ctx || !ctx->have_scene || !ctx->grids.path_enabled || kind < 0 || kind > 3)
        return -1;
    if (kind < 2)
        return kind;
    const int n = (int)(ctx->T->h_spheres.size() / 9);
    if (sphere < 0 || sphere >= n)
        return -1;
    if (kind == 2)
        return 2 + sphere;
    if (!parent_origin)
        return -1;
    if (!ctx->grids.patch_m)
        return 2 + n + sphere; // one family per sphere
    const double *c = ctx->T->h_spheres.data() + 9 * (size_t)sphere;
    const int k = trt_patch_of(ctx->grids.patch_m, parent_origin[0] - c[0], parent_origin[1] - c[1], parent_origin[2] - c[2]);
    return 2 + n + ((sphere << TRT_PATCH_SHIFT) | k);
}

extern "C" int trt_set_path_grids_min_spheres(trt_context *ctx, int min_spheres)
{
    if (!ctx || min_spheres < 0)
        return fail(TRT_ERR_ARGUMENT, "min_spheres %d", min_spheres);
    if (const int shared = refuse_if_shared(ctx, "trt_set_path_grids_min_spheres"))
        return shared;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->path_min_spheres = min_spheres;
    if (!ctx->have_scene)
        return TRT_OK;
    const int n = (int)(ctx->T->h_spheres.size() / 9);
    std::vector<float> table((size_t)trt_cull_padded(n, trt::kCullGroup) * 4 + 4);
    trt_cull_scene cs;
    trt_cull_build(ctx->T->h_spheres.data(), n, trt::kCullGroup, table.data(), &cs);
    const int rc = build_tables(ctx, cs, ctx->scene.ground);
    return rc ? rc : refresh_occupancy(ctx);
}

extern "C" long trt_read_path_tables(trt_context *ctx, const Camera *camera, unsigned long long *cells, size_t capacity_cells,
                                     unsigned long long *pool, size_t capacity_pool, long info[8])
{
    if (!ctx || !camera || !cells || !pool || !info)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (!ctx->have_scene)
        return fail(TRT_ERR_NO_SCENE, "no scene");
    HIP_TRY(hipSetDevice(ctx->device));
    const int rc = ensure_eye_tables(ctx, camera, ctx->stream);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const trt::GridView &g = ctx->grids;
    const int n = (int)(ctx->T->h_spheres.size() / 9);
    unsigned long long used[16 * (1 + kEyeSlots)] = {0};
    HIP_TRY(hipMemcpy(used, ctx->T->d_pool_used.ptr, sizeof used, hipMemcpyDeviceToHost));
    const size_t eye_total = 2 * 6 * (size_t)g.g_eye * g.g_eye, sph_total = 2 * (size_t)n * (size_t)g.patch_count * 6 * (size_t)g.g_sph * g.g_sph;
    const size_t total = g.path_enabled ? eye_total + sph_total : 0;
    const size_t pool_words = ctx->T->pool_scene_words + kEyeSlots * ctx->T->pool_eye_words; // the whole pool: the cells' offsets are into it
    const size_t eye_from = ctx->T->pool_scene_words + (size_t)ctx->eye_slot * ctx->T->pool_eye_words;
    info[0] = g.path_enabled, info[1] = g.g_eye, info[2] = g.g_sph, info[3] = n, info[4] = (long)total;
    info[5] = (long)std::min<unsigned long long>(used[0], 1ull << 62), info[6] = (long)used[16 * (1 + ctx->eye_slot)] - (long)eye_from, info[7] = (long)pool_words;
    if (!g.path_enabled)
        return 0;
    if (capacity_cells < total || capacity_pool < pool_words)
        return fail(TRT_ERR_CAPACITY, "tables have %zu cells and %zu pool words", total, pool_words);
    HIP_TRY(hipMemcpy(cells, ctx->T->d_path_lists.ptr + g.eye_at, eye_total * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (sph_total)
        HIP_TRY(hipMemcpy(cells + eye_total, ctx->T->d_path_lists.ptr + g.sph_at, sph_total * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pool, ctx->T->d_pool.ptr, pool_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return (long)total;
}

extern "C" long trt_read_light_grid(trt_context *ctx, int point_light, int index, unsigned long long *masks, size_t capacity_words)
{
    if (!ctx || !masks || index < 0)
        return fail(TRT_ERR_ARGUMENT, "bad argument");
    if (!ctx->have_scene)
        return fail(TRT_ERR_NO_SCENE, "no scene");
    const trt::GridView &g = ctx->grids;
    if (!g.enabled)
        return 0;
    if (index >= (point_light ? ctx->scene.num_point : ctx->scene.num_dir))
        return fail(TRT_ERR_ARGUMENT, "light %d", index);
    const size_t words = (size_t)std::max((ctx->scene.num_spheres + 63) / 64, 1);
    const size_t stride = (point_light ? g.point_stride : g.dir_stride) * words; // mask words of one light's table
    if (capacity_words < stride)
        return fail(TRT_ERR_CAPACITY, "table has %zu words, buffer %zu", stride, capacity_words);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(masks, (point_light ? ctx->T->d_point_masks.ptr : ctx->T->d_dir_masks.ptr) + stride * (size_t)index, stride * sizeof(unsigned long long),
                      hipMemcpyDeviceToHost));
    return (long)stride;
}

extern "C" int trt_enable_counters(trt_context *ctx, int enable)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    ctx->counters_enabled = enable != 0;
    return TRT_OK;
}

extern "C" int trt_read_counters(trt_context *ctx, unsigned long long *path_rays, unsigned long long *shadow_rays)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long c[kCounterSlots];
    HIP_TRY(hipMemcpy(c, ctx->d_counters.ptr, sizeof c, hipMemcpyDeviceToHost));
    ctx->last_trips = c[2];
    ctx->last_phase2 = c[3];
    ctx->last_swept = c[28];
    ctx->last_passes = c[29];
    for (int k = 0; k < 7; k++)
        ctx->last_loops[k] = c[30 + k];
    if (getenv("TRT_PRINT_STAMPS"))
    { // diagnostic builds only (-DTRT_STAMP=1): per-stage wave-cycle sums
        static const char *const names[24] = {"units+primary", "unit(next_dir)", "P set-up", "P sweep", "P exact tests", "P plane",
                                              "P post: hit", "P post: sky", "Sd look-up", "Sd set-up/load", "Sd sweep", "Sd exact tests",
                                              "Sd plane", "Sd tail", "Sp unit/look-up", "Sp set-up/load", "Sp sweep", "Sp exact tests",
                                              "Sp plane", "Sp tail", "lit accumulate", "END", "loop edge", "-"};
        const int slots = 24;
        unsigned long long total = 0;
        for (int i = 0; i < slots; i++)
            total += c[4 + i];
        for (int i = 0; i < slots && total; i++)
            fprintf(stderr, "stamp %-16s %6.2f %%  %llu\n", names[i], 100.0 * c[4 + i] / total, c[4 + i]);
    }
#if defined(TRT_MARKS) && TRT_MARKS == 2
    if (getenv("TRT_PRINT_PROFILE"))
        for (int k = 0; k < trt::kProfileKinds; k++)
            for (int s = 0; s < 64; s++)
                if (c[trt::kProfileAt + 64 * k + s])
                    fprintf(stderr, "profile %d %d %llu\n", k, s, c[trt::kProfileAt + 64 * k + s]);
#endif
    if (path_rays)
        *path_rays = c[0];
    if (shadow_rays)
        *shadow_rays = c[1];
    return TRT_OK;
}

// `lane_set` 0: the context's stream, queue word 0, d_samples; 1: the alternate stream, its own queue word and scratch
// (trt_render_host renders odd bands there).
static int render_device_on(trt_context *ctx, const Camera *camera, const trt_rowset *rows, int bounce_limit, int rays_per_pixel,
                            void *d_pixels, size_t capacity_bytes, int lane_set);

extern "C" int trt_render_device(trt_context *ctx, const Camera *camera, const trt_rowset *rows, int bounce_limit,
                                 int rays_per_pixel, void *d_pixels, size_t capacity_bytes)
{
    return render_device_on(ctx, camera, rows, bounce_limit, rays_per_pixel, d_pixels, capacity_bytes, 0);
}

static int render_device_on(trt_context *ctx, const Camera *camera, const trt_rowset *rows, int bounce_limit, int rays_per_pixel,
                            void *d_pixels, size_t capacity_bytes, int lane_set)
{
    if (!ctx || !camera || !d_pixels)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (!rowset_valid(rows))
        return fail(TRT_ERR_ARGUMENT, "invalid rowset");
    if (bounce_limit < 1 || rays_per_pixel < 1) // bounce_limit 0 divides 0 by 0 in the reference (TRT.c:1061)
        return fail(TRT_ERR_ARGUMENT, "bounce_limit %d / rays_per_pixel %d", bounce_limit, rays_per_pixel);
    if (!ctx->have_scene)
        return fail(TRT_ERR_NO_SCENE, "trt_set_scene has not been called");
    const int local_rows = trt_rowset_rows(rows);
    const size_t need = (size_t)local_rows * rows->width * sizeof(Vector);
    if (capacity_bytes < need)
        return fail(TRT_ERR_CAPACITY, "framebuffer needs %zu B, %zu given", need, capacity_bytes);
    if (local_rows == 0)
        return TRT_OK;
    if ((unsigned long long)local_rows * rows->width >= 0x7fffffffull)
        return fail(TRT_ERR_ARGUMENT, "%d x %d pixels exceed the 2^31 pixel index range", local_rows, rows->width);
    HIP_TRY(hipSetDevice(ctx->device));
    const hipStream_t stream = lane_set ? ctx->alt_stream : ctx->stream;
    DeviceBuffer<double> &scratch = lane_set ? ctx->d_samples_alt : ctx->d_samples;
    int rc = prepare_jitter(ctx, camera, rows->width, rows->height, rays_per_pixel);
    if (rc)
        return rc;
    rc = prepare_axes(ctx, camera, rows->width, rows->height);
    if (rc)
        return rc;

    trt::FrameView f{};
    memcpy(f.cam, camera, sizeof(Camera));
    f.jitter = ctx->d_jitter.ptr;
    f.col_x = ctx->d_axes.ptr;
    f.row_y = ctx->d_axes.ptr + rows->width;
    f.inv_spp = 1.0 / rays_per_pixel;
    f.width_magic = (unsigned)std::min<unsigned long long>((0x100000000ull + (unsigned)rows->width - 1) / (unsigned)rows->width, 0xffffffffull);
    f.tile_magic = (unsigned)std::min<unsigned long long>((0x100000000ull + (unsigned)rows->tile_rows - 1) / (unsigned)rows->tile_rows, 0xffffffffull);
    f.out = (double *)d_pixels;
    f.counters = ctx->counters_enabled ? ctx->d_counters.ptr : nullptr;
#if defined(TRT_MARKS) && TRT_MARKS == 2
    f.counters = ctx->d_counters.ptr; // the ISA profile of the SHIPPING instantiations lands there
    HIP_TRY(hipMemsetAsync(ctx->d_counters.ptr, 0, kCounterSlots * sizeof(unsigned long long), lane_set ? ctx->alt_stream : ctx->stream));
#endif
    f.queue = ctx->d_queue.ptr + 16 * lane_set; // a cache line apart
    f.width = rows->width;
    f.height = rows->height;
    f.tile_rows = rows->tile_rows;
    f.tile_first = rows->tile_first;
    f.tile_step = rows->tile_step;
    f.local_rows = local_rows;
    f.bounce_limit = bounce_limit;
    f.spp = rays_per_pixel;

    rc = ensure_eye_tables(ctx, camera, stream); // no-op unless the eye moved (trt_render_host builds them before it forks its streams)
    if (rc)
        return rc;
    const long pixels = (long)local_rows * rows->width;
    const size_t lds = scene_lds_bytes(ctx->scene);
    if (ctx->counters_enabled)
        HIP_TRY(hipMemsetAsync(ctx->d_counters.ptr, 0, kCounterSlots * sizeof(unsigned long long), stream));
    const int slot = (int)(ctx->launches % kEventRing);
    if (ctx->kernel == 1)
    {
        const int block = 256;
        const unsigned grid = (unsigned)((pixels + block - 1) / block);
        HIP_TRY(hipEventRecord(ctx->ev_start[slot], stream));
        hipLaunchKernelGGL(trt::render_simple_kernel, dim3(grid), dim3(block), lds, stream, ctx->scene, f);
        HIP_TRY(hipEventRecord(ctx->ev_mid[slot], stream));
        HIP_TRY(hipEventRecord(ctx->ev_stop[slot], stream));
    }
    else
    {
        HIP_TRY(hipMemsetAsync(ctx->d_queue.ptr + 16 * lane_set, 0, 16 * sizeof(unsigned int), stream));
        // production (kernel 0): persistent waves, synchronous rounds over SAMPLE units, then the ordered mean per pixel
        const long units = pixels * rays_per_pixel;
        if ((unsigned long long)units >= 0x7fffffffull)
            return fail(TRT_ERR_ARGUMENT, "%ld work units exceed the 2^31 index range", units);
        if (scratch.capacity < (size_t)units * 3)
            HIP_TRY(hipStreamSynchronize(stream)); // a frame in flight may still use the old scratch
        HIP_TRY(scratch.reserve((size_t)units * 3));
        f.samples = scratch.ptr;
        f.spp_magic = (unsigned)std::min<unsigned long long>((0x100000000ull + (unsigned)rays_per_pixel - 1) / (unsigned)rays_per_pixel, 0xffffffffull);
        // shading decoupled from the owning lane (COMPACT, trt_rounds.hpp) when the rings fit in LDS: by default only if they
        // cost no resident wave and the scene has lights enough to pay for them.
        // (the occupancy figures were taken for 64 rays per pixel: with more, the jitter table may push the rings out of LDS)
        const bool compact = renders_decoupled(ctx, units) && compact_lds_bytes(ctx, rays_per_pixel) <= (size_t)ctx->lds_limit;
        ctx->last_units = units;
        if (image_lds_bytes(ctx, rays_per_pixel) > (size_t)ctx->lds_limit)
            return fail(TRT_ERR_CAPACITY, "scene and %d rays per pixel need %zu B of LDS staging, device offers %d", rays_per_pixel,
                        image_lds_bytes(ctx, rays_per_pixel), ctx->lds_limit);
        trt::PersistentLaunch pl = trt::persistent_launch_shape(ctx->compute_units - (ctx->stream == ctx->own_stream ? ctx->reserved_cus : 0),
                                                                ctx->rounds_blocks_per_cu, units);
        if (compact)
        {
            const long cap = (long)(ctx->compute_units - (ctx->stream == ctx->own_stream ? ctx->reserved_cus : 0)) * ctx->compact_blocks_per_cu;
            const long want = (units + trt::kCompactBlock - 1) / trt::kCompactBlock;
            pl = trt::PersistentLaunch{(unsigned)std::max(1L, std::min(want, cap)), (unsigned)trt::kCompactBlock};
        }
        const size_t plds = image_lds_bytes(ctx, rays_per_pixel);
        const dim3 grid(pl.grid), block(pl.block);
        if (ctx->ior_count && ctx->ior_count != ctx->scene.num_spheres) // before the first event of the launch is recorded
            return fail(TRT_ERR_ARGUMENT, "trt_set_refraction was given %d indices, the scene has %d spheres", ctx->ior_count, ctx->scene.num_spheres);
        ctx->last_compact = compact;
        HIP_TRY(hipEventRecord(ctx->ev_start[slot], stream));
        const bool patches = ctx->grids.path_enabled && ctx->grids.patch_m > 0; // a family per patch of a sphere: its own instantiations
        if (ctx->ior_count)
        { // the refraction extension (parity unpinned): its own instantiation, the reference's path is not touched
            f.ior = ctx->d_ior.ptr;
            if (patches && ctx->counters_enabled)
                hipLaunchKernelGGL((trt::render_rounds_kernel<true, true, false, true>), grid, block, plds, stream, ctx->scene, ctx->cull, f, ctx->grids);
            else if (patches)
                hipLaunchKernelGGL((trt::render_rounds_kernel<false, true, false, true>), grid, block, plds, stream, ctx->scene, ctx->cull, f, ctx->grids);
            else if (ctx->counters_enabled)
                hipLaunchKernelGGL((trt::render_rounds_kernel<true, true>), grid, block, plds, stream, ctx->scene, ctx->cull, f, ctx->grids);
            else
                hipLaunchKernelGGL((trt::render_rounds_kernel<false, true>), grid, block, plds, stream, ctx->scene, ctx->cull, f, ctx->grids);
        }
        else if (compact)
        {
            f.ring_at = (unsigned)compact_ring_at(ctx, rays_per_pixel);
            const size_t clds = compact_lds_bytes(ctx, rays_per_pixel);
            if (ctx->counters_enabled)
                hipLaunchKernelGGL((trt::render_rounds_kernel<true, false, true>), grid, block, clds, stream, ctx->scene, ctx->cull, f, ctx->grids);
            else
                hipLaunchKernelGGL((trt::render_rounds_kernel<false, false, true>), grid, block, clds, stream, ctx->scene, ctx->cull, f, ctx->grids);
        }
        else if (patches && ctx->counters_enabled)
            hipLaunchKernelGGL((trt::render_rounds_kernel<true, false, false, true>), grid, block, plds, stream, ctx->scene, ctx->cull, f, ctx->grids);
        else if (patches)
            hipLaunchKernelGGL((trt::render_rounds_kernel<false, false, false, true>), grid, block, plds, stream, ctx->scene, ctx->cull, f, ctx->grids);
        else if (ctx->counters_enabled)
            hipLaunchKernelGGL((trt::render_rounds_kernel<true>), grid, block, plds, stream, ctx->scene, ctx->cull, f, ctx->grids);
        else
            hipLaunchKernelGGL((trt::render_rounds_kernel<false>), grid, block, plds, stream, ctx->scene, ctx->cull, f, ctx->grids);
        HIP_TRY(hipEventRecord(ctx->ev_mid[slot], stream));
#if !TRT_AB_SKIP_REDUCE // diagnostic build (profiles/r03: what the ordered mean's streaming pass costs in the pipelined loop)
        { // TRT.c:1063-1065: the mean over each pixel's samples, in sample order
            const long values = pixels * 3;
            hipLaunchKernelGGL(trt::reduce_samples_kernel, dim3((unsigned)((values + 255) / 256)), dim3(256), 0, stream,
                               (const double *)scratch.ptr, (double *)d_pixels, values, rays_per_pixel, f.inv_spp);
        }
#endif
        HIP_TRY(hipEventRecord(ctx->ev_stop[slot], stream));
    }
    HIP_TRY(hipGetLastError());
    ctx->launches++;
    return TRT_OK;
}

extern "C" int trt_quantize_device(trt_context *ctx, const void *d_pixels, size_t num_pixels, void *d_rgb8)
{
    if (!ctx || !d_pixels || !d_rgb8)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (num_pixels == 0)
        return TRT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const long n = (long)num_pixels * 3;
    hipLaunchKernelGGL(trt::quantize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)d_pixels,
                       n, (unsigned char *)d_rgb8);
    HIP_TRY(hipGetLastError());
    return TRT_OK;
}

extern "C" int trt_synchronize(trt_context *ctx)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return TRT_OK;
}

namespace
{
double host_now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
bool print_host_times()
{
    static const bool on = getenv("TRT_PRINT_HOST_TIMES") != nullptr;
    return on;
}
} // namespace

extern "C" int trt_render_host(trt_context *ctx, const Camera *camera, const trt_rowset *rows, int bounce_limit,
                               int rays_per_pixel, Vector *pixels)
{
    if (!ctx || !pixels)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (!rowset_valid(rows))
        return fail(TRT_ERR_ARGUMENT, "invalid rowset");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t count = (size_t)trt_rowset_rows(rows) * rows->width;
    const size_t bytes = count * sizeof(Vector);
    HIP_TRY(ctx->d_fb.reserve(count * 3));
    if (ctx->h_staging_bytes < bytes)
    {
        if (ctx->h_staging)
            (void)hipHostFree(ctx->h_staging);
        ctx->h_staging = nullptr;
        ctx->h_staging_bytes = 0;
        HIP_TRY(hipHostMalloc((void **)&ctx->h_staging, std::max<size_t>(bytes, 1), hipHostMallocDefault));
        ctx->h_staging_bytes = std::max<size_t>(bytes, 1);
    }
    const double t_begin = host_now_ms();
    // A whole frame is rendered in up to four bands of rows: while band b+1 is being rendered, band b crosses PCIe on the
    // copy stream into pinned staging, chunk by chunk (an event per chunk), and a few host threads copy landed chunks
    // into the caller's (pageable) buffer.  Shards and small frames are one band.
    const int local_rows = trt_rowset_rows(rows);
    const bool whole = rows->tile_first == 0 && rows->tile_step == 1 && rows->tile_rows >= rows->height;
    static const int band_count = getenv("TRT_HOST_BANDS") ? std::min(8, std::max(1, atoi(getenv("TRT_HOST_BANDS")))) : 4;
#if defined(TRT_MARKS) && TRT_MARKS == 2
    const int bands = 1; // the ISA profile is of ONE launch
#else
    const int bands = whole && !ctx->counters_enabled && local_rows >= 256 && bytes >= (32u << 20) ? band_count : 1;
#endif
    const int band_rows = (local_rows + bands - 1) / bands;
    const size_t row_bytes = (size_t)rows->width * sizeof(Vector);
    const int chunks_per_band = (int)std::min<size_t>(16 / bands, std::max<size_t>(1, (size_t)band_rows * row_bytes / (4u << 20)));
    int chunks = 0;
    size_t chunk_at[16], chunk_len[16];
    if (bands > 1 && !ctx->copy_stream)
    { // created on first use: every stream of a process competes for a handful of hardware queues, and two streams that
      // land on one queue run one after the other (a renderer that never comes here keeps its streams to itself)
        HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&ctx->alt_stream, hipStreamNonBlocking));
    }
    const hipStream_t copy_stream = bands > 1 ? ctx->copy_stream : ctx->stream;
    if (ctx->have_scene && camera)
    { // both render streams read the eye's tables: build them before the fork
        const int rc = ensure_eye_tables(ctx, camera, ctx->stream);
        if (rc)
            return rc;
    }
    if (bands > 1)
    { // the alternate stream starts behind whatever the caller queued on the context's stream before this call
        HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->alt_stream, ctx->ev_fork, 0));
    }
    for (int b = 0; b < bands; b++)
    {
        trt_rowset band = *rows;
        if (bands > 1)
            band = trt_rowset{rows->width, rows->height, band_rows, b, bands};
        const int rows_here = trt_rowset_rows(&band);
        const size_t at = (size_t)b * band_rows * row_bytes, len = (size_t)rows_here * row_bytes;
        const int set = bands > 1 ? (b & 1) : 0; // odd bands on the alternate stream: a band's tail and reduction overlap the next band
        int rc = render_device_on(ctx, camera, &band, bounce_limit, rays_per_pixel, (char *)ctx->d_fb.ptr + at, len, set);
        if (rc)
            return rc;
        if (bands > 1)
        { // a second stream costs ~0.1 ms of cross-queue hand-over: only where there is something to overlap
            HIP_TRY(hipEventRecord(ctx->ev_band[b], set ? ctx->alt_stream : ctx->stream));
            HIP_TRY(hipStreamWaitEvent(copy_stream, ctx->ev_band[b], 0));
        }
        const size_t per = ((len + chunks_per_band - 1) / chunks_per_band + 63) / 64 * 64;
        for (int i = 0; i < chunks_per_band; i++, chunks++)
        {
            chunk_at[chunks] = at + (size_t)i * per;
            chunk_len[chunks] = (size_t)i * per < len ? std::min(per, len - (size_t)i * per) : 0;
            if (chunk_len[chunks])
                HIP_TRY(hipMemcpyAsync((char *)ctx->h_staging + chunk_at[chunks], (const char *)ctx->d_fb.ptr + chunk_at[chunks], chunk_len[chunks],
                                       hipMemcpyDeviceToHost, copy_stream));
            HIP_TRY(hipEventRecord(ctx->ev_chunk[chunks], copy_stream));
        }
    }
    const double t_enqueued = host_now_ms();
    const int workers = chunks >= 4 ? 4 : 1;
    hipError_t worker_error[4] = {hipSuccess, hipSuccess, hipSuccess, hipSuccess};
    auto drain = [&](int w) {
        (void)hipSetDevice(ctx->device);
        for (int i = w; i < chunks; i += workers)
        {
            const hipError_t e = hipEventSynchronize(ctx->ev_chunk[i]);
            if (e != hipSuccess)
            {
                worker_error[w] = e;
                return;
            }
            memcpy((char *)pixels + chunk_at[i], (const char *)ctx->h_staging + chunk_at[i], chunk_len[i]);
        }
    };
    if (workers == 1)
        drain(0);
    else
    {
        std::thread pool[3];
        for (int w = 1; w < workers; w++)
            pool[w - 1] = std::thread(drain, w);
        drain(0);
        for (int w = 1; w < workers; w++)
            pool[w - 1].join();
    }
    for (int w = 0; w < workers; w++)
        HIP_TRY(worker_error[w]);
    if (bands > 1)
    {
        HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
        HIP_TRY(hipStreamSynchronize(ctx->alt_stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (print_host_times())
        fprintf(stderr, "trt_render_host: %d band(s), enqueue %.3f ms, render + copy-out of %zu bytes %.3f ms\n", bands, t_enqueued - t_begin, bytes,
                host_now_ms() - t_enqueued);
    return TRT_OK;
}

extern "C" int trt_kernel_times(trt_context *ctx, float *ms, int max)
{
    if (!ctx || !ms || max < 0)
        return fail(TRT_ERR_ARGUMENT, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const long have = std::min<long>(ctx->launches, kEventRing);
    const long n = std::min<long>(have, max);
    for (long i = 0; i < n; i++)
    {
        const long launch = ctx->launches - n + i;
        const int slot = (int)(launch % kEventRing);
        HIP_TRY(hipEventElapsedTime(&ms[i], ctx->ev_start[slot], ctx->ev_stop[slot]));
    }
    return (int)n;
}

extern "C" int trt_render_kernel_times(trt_context *ctx, float *render_ms, float *reduce_ms, int max)
{
    if (!ctx || !render_ms || max < 0)
        return fail(TRT_ERR_ARGUMENT, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const long have = std::min<long>(ctx->launches, kEventRing);
    const long n = std::min<long>(have, max);
    for (long i = 0; i < n; i++)
    {
        const long launch = ctx->launches - n + i;
        const int slot = (int)(launch % kEventRing);
        HIP_TRY(hipEventElapsedTime(&render_ms[i], ctx->ev_start[slot], ctx->ev_mid[slot]));
        if (reduce_ms)
            HIP_TRY(hipEventElapsedTime(&reduce_ms[i], ctx->ev_mid[slot], ctx->ev_stop[slot]));
    }
    return (int)n;
}

extern "C" int trt_kernel_info(trt_context *ctx, int *vgprs, int *sgprs, int *static_lds_bytes, int *max_blocks_per_cu,
                               int *compute_units)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    const bool decoupled = ctx->have_scene && (ctx->last_units > 0 ? ctx->last_compact : renders_decoupled(ctx, kCompactionMinUnits));
    const bool patches = ctx->have_scene && ctx->grids.path_enabled && ctx->grids.patch_m > 0;
    const void *fn = ctx->kernel == 1 ? (const void *)trt::render_simple_kernel
                     : decoupled      ? (const void *)trt::render_rounds_kernel<false, false, true>
                     : patches        ? (const void *)trt::render_rounds_kernel<false, false, false, true>
                                      : (const void *)trt::render_rounds_kernel<false>;
    hipFuncAttributes attr;
    HIP_TRY(hipFuncGetAttributes(&attr, fn));
    if (vgprs)
        *vgprs = attr.numRegs;
    if (sgprs)
        *sgprs = 0; // not reported by hipFuncGetAttributes; see profiles/*resource_usage*.txt
    if (static_lds_bytes)
        *static_lds_bytes = (int)attr.sharedSizeBytes;
    if (max_blocks_per_cu)
    {
        int blocks = 0;
        const size_t lds = ctx->have_scene ? (ctx->kernel == 1 ? scene_lds_bytes(ctx->scene) : image_lds_bytes(ctx, 64)) : 0;
        if (ctx->kernel == 1)
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, trt::render_simple_kernel, 256, lds));
        else if (decoupled)
            blocks = ctx->compact_blocks_per_cu; // workgroups of kCompactBlock threads
        else
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, trt::render_rounds_kernel<false>, trt::kPersistentBlock, lds));
        *max_blocks_per_cu = blocks;
    }
    if (compute_units)
        *compute_units = ctx->compute_units;
    return TRT_OK;
}

extern "C" int trt_selftest_div_sqrt(trt_context *ctx, const double *a, const double *b, size_t n, double *quot, double *root)
{
    if (!ctx || !a || !b || !quot || !root)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (n == 0)
        return TRT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    DeviceBuffer<double> buf;
    HIP_TRY(buf.reserve(4 * n));
    double *da = buf.ptr, *db = buf.ptr + n, *dq = buf.ptr + 2 * n, *dr = buf.ptr + 3 * n;
    HIP_TRY(hipMemcpy(da, a, n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b, n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(trt::div_sqrt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, da, db, (long)n, dq, dr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(quot, dq, n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(root, dr, n * sizeof(double), hipMemcpyDeviceToHost));
    buf.release();
    return TRT_OK;
}

extern "C" int trt_selftest_unit(trt_context *ctx, const double *xyzw, size_t n, double *fast, double *reference)
{
    if (!ctx || !xyzw || !fast || !reference)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (n == 0)
        return TRT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    DeviceBuffer<double> buf;
    HIP_TRY(buf.reserve(12 * n));
    double *dv = buf.ptr, *df = buf.ptr + 4 * n, *dr = buf.ptr + 8 * n;
    HIP_TRY(hipMemcpy(dv, xyzw, 4 * n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(trt::unit_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dv, (long)n, df, dr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(fast, df, 4 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(reference, dr, 4 * n * sizeof(double), hipMemcpyDeviceToHost));
    buf.release();
    return TRT_OK;
}

namespace
{
// trt_cube_lookup as the DEVICE evaluates it (v_cubeid / v_cubesc / v_cubetc / v_cubema): {face, sc, tc, ma2} per direction
__global__ void cube_selftest_kernel(const float *xyz, long n, float *out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    int face;
    float sc, tc, ma2;
    trt_cube_lookup(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &face, &sc, &tc, &ma2);
    out[4 * i] = (float)face, out[4 * i + 1] = sc, out[4 * i + 2] = tc, out[4 * i + 3] = ma2;
}
} // namespace

namespace
{
// trt_selftest_sky: per direction the reference's texel index by the FP64 form, the FP32 estimate's, and whether the estimate
// calls itself ambiguous (the kernel then takes the FP64 form)
__global__ void sky_selftest_kernel(const double *dirs, long n, int dim, long *exact, long *estimate, int *ambiguous)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const trt::d3 d = trt::d3{dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
    bool amb;
    exact[i] = trt::sky_index_unit(dim, d, (double)dim);
    estimate[i] = trt::sky_index_estimate(dim, (float)dim, d, amb);
    ambiguous[i] = amb;
}
} // namespace

extern "C" int trt_selftest_sky(trt_context *ctx, const double *dirs, size_t n, int dim, long long *exact, long long *estimate, int *ambiguous)
{
    if (!ctx || !dirs || !exact || !estimate || !ambiguous || dim < 1)
        return fail(TRT_ERR_ARGUMENT, "bad argument");
    if (n == 0)
        return TRT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    DeviceBuffer<double> in;
    DeviceBuffer<long> out;
    DeviceBuffer<int> flags;
    HIP_TRY(in.reserve(3 * n));
    HIP_TRY(out.reserve(2 * n));
    HIP_TRY(flags.reserve(n));
    HIP_TRY(hipMemcpy(in.ptr, dirs, 3 * n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sky_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)in.ptr, (long)n, dim, out.ptr, out.ptr + n, flags.ptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(exact, out.ptr, n * sizeof(long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(estimate, out.ptr + n, n * sizeof(long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ambiguous, flags.ptr, n * sizeof(int), hipMemcpyDeviceToHost));
    in.release(), out.release(), flags.release();
    return TRT_OK;
}

extern "C" int trt_selftest_cube(trt_context *ctx, const float *xyz, size_t n, float *device_out, float *host_out)
{
    if (!ctx || !xyz || !device_out || !host_out)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (n == 0)
        return TRT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    DeviceBuffer<float> buf;
    HIP_TRY(buf.reserve(7 * n));
    HIP_TRY(hipMemcpy(buf.ptr, xyz, 3 * n * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cube_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const float *)buf.ptr, (long)n, buf.ptr + 3 * n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(device_out, buf.ptr + 3 * n, 4 * n * sizeof(float), hipMemcpyDeviceToHost));
    buf.release();
    for (size_t i = 0; i < n; i++)
    { // the same header compiled for the host: the C restatement of the four instructions
        int face;
        float sc, tc, ma2;
        trt_cube_lookup(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &face, &sc, &tc, &ma2);
        host_out[4 * i] = (float)face, host_out[4 * i + 1] = sc, host_out[4 * i + 2] = tc, host_out[4 * i + 3] = ma2;
    }
    return TRT_OK;
}

extern "C" int trt_probe_rays(trt_context *ctx, const Ray *rays, size_t n, int *obj, double *point, double *normal,
                              double *material, double *lit)
{
    if (!ctx || !rays || !obj || !point || !normal || !material || !lit)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (!ctx->have_scene)
        return fail(TRT_ERR_NO_SCENE, "trt_set_scene has not been called");
    if (n == 0)
        return TRT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    DeviceBuffer<double> buf;
    DeviceBuffer<int> dobj;
    HIP_TRY(buf.reserve(n * (6 + 3 + 3 + 5 + 3)));
    HIP_TRY(dobj.reserve(n));
    double *dr = buf.ptr, *dp = dr + 6 * n, *dn = dp + 3 * n, *dm = dn + 3 * n, *dl = dm + 5 * n;
    HIP_TRY(hipMemcpy(dr, rays, n * sizeof(Ray), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(trt::probe_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), scene_lds_bytes(ctx->scene), ctx->stream,
                       ctx->scene, dr, (long)n, dobj.ptr, dp, dn, dm, dl);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(obj, dobj.ptr, n * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(point, dp, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(normal, dn, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(material, dm, 5 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lit, dl, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    buf.release();
    dobj.release();
    return TRT_OK;
}

extern "C" int trt_probe_rays_production(trt_context *ctx, const Camera *camera, const Ray *rays, const int *families, size_t n, int *obj,
                                         double *point, double *normal, double *material, double *lit)
{
    if (!ctx || !camera || !rays || !obj || !point || !normal || !material || !lit)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (!ctx->have_scene)
        return fail(TRT_ERR_NO_SCENE, "trt_set_scene has not been called");
    if (n == 0)
        return TRT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const int rc = ensure_eye_tables(ctx, camera, ctx->stream);
    if (rc)
        return rc;
    DeviceBuffer<double> buf;
    DeviceBuffer<int> dobj;
    HIP_TRY(buf.reserve(n * (6 + 3 + 3 + 5 + 3)));
    HIP_TRY(dobj.reserve(2 * n));
    double *dr = buf.ptr, *dp = dr + 6 * n, *dn = dp + 3 * n, *dm = dn + 3 * n, *dl = dm + 5 * n;
    HIP_TRY(hipMemcpy(dr, rays, n * sizeof(Ray), hipMemcpyHostToDevice));
    if (families)
    { // only codes the kernel can decode reach it: 0, 1, 2 + i, and 2 + N + i (one family per sphere) or 2 + N + (i << 7 | k) with
      // k < patches (a patch number beyond the tables would index past the LDS image and the lists); anything else: no family
        const int ns = ctx->scene.num_spheres, pm = ctx->grids.path_enabled ? ctx->grids.patch_m : 0, pc = ctx->grids.path_enabled ? ctx->grids.patch_count : 0;
        std::vector<int> codes(families, families + n);
        for (int &c : codes)
        {
            bool ok = c == 0 || c == 1 || (c >= 2 && c < 2 + ns);
            if (!ok && c >= 2 + ns)
            {
                const int rest = c - 2 - ns;
                ok = pm ? ((rest >> TRT_PATCH_SHIFT) < ns && (rest & ((1 << TRT_PATCH_SHIFT) - 1)) < pc) : rest < ns;
            }
            if (!ok)
                c = -1;
        }
        HIP_TRY(hipMemcpy(dobj.ptr + n, codes.data(), n * sizeof(int), hipMemcpyHostToDevice));
    }
    trt::FrameView f{};
    memcpy(f.cam, camera, sizeof(Camera));
    f.jitter = ctx->d_jitter.ptr; // spp = 0: nothing is read through it
    if (ctx->grids.path_enabled && ctx->grids.patch_m > 0)
        hipLaunchKernelGGL(trt::probe_rounds_kernel<true>, dim3((unsigned)((n + trt::kPersistentBlock - 1) / trt::kPersistentBlock)), dim3(trt::kPersistentBlock),
                           image_lds_bytes(ctx, 0), ctx->stream, ctx->scene, ctx->cull, f, ctx->grids, (const double *)dr,
                           families ? (const int *)(dobj.ptr + n) : (const int *)nullptr, (long)n, dobj.ptr, dp, dn, dm, dl);
    else
        hipLaunchKernelGGL(trt::probe_rounds_kernel<false>, dim3((unsigned)((n + trt::kPersistentBlock - 1) / trt::kPersistentBlock)), dim3(trt::kPersistentBlock),
                           image_lds_bytes(ctx, 0), ctx->stream, ctx->scene, ctx->cull, f, ctx->grids, (const double *)dr,
                           families ? (const int *)(dobj.ptr + n) : (const int *)nullptr, (long)n, dobj.ptr, dp, dn, dm, dl);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(obj, dobj.ptr, n * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(point, dp, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(normal, dn, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(material, dm, 5 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lit, dl, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    buf.release();
    dobj.release();
    return TRT_OK;
}

// ---- default context: the drop-in layer ---------------------------------------------------------------------

namespace
{
trt_context *g_default = nullptr;
int g_default_device = 0;
// The reference's project_scene is a pure function of its arguments and may be called from several threads; the drop-in
// shares one device context, so calls on the default context take turns.
std::mutex g_default_mutex;

int default_context(trt_context **out)
{
    if (!g_default)
    {
        int rc = trt_create(g_default_device, &g_default);
        if (rc)
            return rc;
    }
    *out = g_default;
    return TRT_OK;
}
} // namespace

extern "C" int trt_init(int device)
{
    std::lock_guard<std::mutex> turn(g_default_mutex);
    if (g_default && g_default->device != device)
    {
        trt_destroy(g_default);
        g_default = nullptr;
    }
    g_default_device = device;
    trt_context *ctx;
    return default_context(&ctx);
}

extern "C" int trt_shutdown(void)
{
    std::lock_guard<std::mutex> turn(g_default_mutex);
    int rc = trt_destroy(g_default);
    g_default = nullptr;
    return rc;
}

extern "C" int trt_upload_skybox(const Skybox *skybox)
{
    std::lock_guard<std::mutex> turn(g_default_mutex);
    if (!skybox)
        return fail(TRT_ERR_ARGUMENT, "skybox is NULL");
    trt_context *ctx;
    int rc = default_context(&ctx);
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return upload_skybox(ctx, skybox);
}

extern "C" int trt_invalidate_skybox(void)
{
    std::lock_guard<std::mutex> turn(g_default_mutex);
    if (g_default)
        g_default->sky_dim = -1;
    return TRT_OK;
}

// The caller owns the scene and may have edited it since the last frame (main() rewrites the camera every frame,
// TRT.c:1327-1336): primitives are a few KB and are re-sent; the 6*dim*dim texels only when the face pointers, the dimension
// or the texel stamp changed.  With g_default_mutex held.
static int refresh_default_scene(trt_context *ctx, const Scene *scene)
{
    HIP_TRY(hipSetDevice(ctx->device));
    const double t_begin = host_now_ms();
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->have_scene = false;
    int rc = upload_primitives(ctx, scene, true);
    if (rc)
        return rc;
    bool same_sky = ctx->sky_dim == scene->skybox.dim;
    for (int f = 0; f < 6 && same_sky; f++)
        same_sky = scene->skybox.colors[f] && ctx->sky_faces[f] == scene->skybox.colors[f];
    same_sky = same_sky && scene->skybox.dim > 0 && ctx->sky_stamp == skybox_stamp(&scene->skybox);
    if (!same_sky)
    {
        rc = upload_skybox(ctx, &scene->skybox);
        if (rc)
            return rc;
    }
    ctx->have_scene = true;
    if (print_host_times())
        fprintf(stderr, "trt_render_frame: scene upload %.3f ms\n", host_now_ms() - t_begin);
    return TRT_OK;
}

// The drop-in entries (project_scene, trt_render_frame, trt_render_frame_rgb8) take the scene with every call.  A scene whose
// primitives differ from the previous call's on `moving_after` consecutive calls is treated as MOVING: its candidate tables are
// rebuilt per call the cheap way (one family per sphere, no patches); after `still_after` consecutive unchanged calls the full
// tables are built once.  moving_after = 0: never (every change builds the full tables).  Defaults 2 and 3.  Frames are
// bit-identical either way.  *moving (may be NULL): whether the default context currently treats its scene as moving.
extern "C" int trt_set_scene_policy(int moving_after, int still_after)
{
    if (moving_after < 0 || still_after < 1)
        return fail(TRT_ERR_ARGUMENT, "scene policy %d, %d", moving_after, still_after);
    std::lock_guard<std::mutex> turn(g_default_mutex);
    g_moving_after = moving_after;
    g_still_after = still_after;
    return TRT_OK;
}

extern "C" int trt_scene_is_moving(void)
{
    std::lock_guard<std::mutex> turn(g_default_mutex);
    return g_default && g_default->moving_scene ? 1 : 0;
}

extern "C" int trt_render_frame(const Scene *scene, Screen *screen, int bounce_limit, int rays_per_pixel)
{
    if (!scene || !screen || !screen->pixels)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (screen->width <= 0 || screen->height <= 0)
        return fail(TRT_ERR_ARGUMENT, "screen %d x %d", screen->width, screen->height);
    std::lock_guard<std::mutex> turn(g_default_mutex);
    trt_context *ctx;
    int rc = default_context(&ctx);
    if (rc)
        return rc;
    rc = refresh_default_scene(ctx, scene);
    if (rc)
        return rc;
    const trt_rowset whole = {screen->width, screen->height, screen->height, 0, 1};
    return trt_render_host(ctx, &scene->camera, &whole, bounce_limit, rays_per_pixel, screen->pixels);
}

// north_star's name for the entry: the frame producer with the two macros of TRT.c:54, :58 as run-time values
extern "C" int render_frame(const Scene *scene, Screen *screen, int bounce_limit, int rays_per_pixel)
{
    return trt_render_frame(scene, screen, bounce_limit, rays_per_pixel);
}

extern "C" int trt_render_host_rgb8(trt_context *ctx, const Camera *camera, const trt_rowset *rows, int bounce_limit, int rays_per_pixel,
                                    unsigned char *rgb)
{
    if (!ctx || !rgb)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (!rowset_valid(rows))
        return fail(TRT_ERR_ARGUMENT, "invalid rowset");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t count = (size_t)trt_rowset_rows(rows) * rows->width;
    if (count == 0)
        return TRT_OK;
    HIP_TRY(ctx->d_fb.reserve(count * 3));
    HIP_TRY(ctx->d_rgb8.reserve(count * 3));
    if (ctx->h_staging_bytes < count * 3)
    {
        if (ctx->h_staging)
            (void)hipHostFree(ctx->h_staging);
        ctx->h_staging = nullptr;
        ctx->h_staging_bytes = 0;
        HIP_TRY(hipHostMalloc((void **)&ctx->h_staging, count * 3, hipHostMallocDefault));
        ctx->h_staging_bytes = count * 3;
    }
    const double t_begin = host_now_ms();
    int rc = trt_render_device(ctx, camera, rows, bounce_limit, rays_per_pixel, ctx->d_fb.ptr, count * sizeof(Vector));
    if (rc)
        return rc;
    rc = trt_quantize_device(ctx, ctx->d_fb.ptr, count, ctx->d_rgb8.ptr); // (int)(c*255), TRT.c:1157-1163, on the device
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(ctx->h_staging, ctx->d_rgb8.ptr, count * 3, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(rgb, ctx->h_staging, count * 3);
    if (print_host_times())
        fprintf(stderr, "trt_render_host_rgb8: %.3f ms for %zu pixels\n", host_now_ms() - t_begin, count);
    return TRT_OK;
}

extern "C" int trt_render_frame_rgb8(const Scene *scene, int width, int height, int bounce_limit, int rays_per_pixel, unsigned char *rgb)
{
    if (!scene || !rgb)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    if (width <= 0 || height <= 0)
        return fail(TRT_ERR_ARGUMENT, "screen %d x %d", width, height);
    std::lock_guard<std::mutex> turn(g_default_mutex);
    trt_context *ctx;
    int rc = default_context(&ctx);
    if (rc)
        return rc;
    rc = refresh_default_scene(ctx, scene);
    if (rc)
        return rc;
    const trt_rowset whole = {width, height, height, 0, 1};
    return trt_render_host_rgb8(ctx, &scene->camera, &whole, bounce_limit, rays_per_pixel, rgb);
}

extern "C" void project_scene(Scene *scene, Screen *screen)
{
    const int rc = trt_render_frame(scene, screen, TRT_REF_BOUNCE_LIMIT, TRT_REF_RAYS_PER_PIXEL);
    if (rc != TRT_OK)
    {
        fprintf(stderr, "project_scene (libtrt_hip): %s\n", trt_last_error());
        abort();
    }
}
