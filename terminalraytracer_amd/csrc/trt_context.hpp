// trt_context.hpp -- INTERNAL to libtrt_hip.so: the renderer context, the scene tables it may share, and the helpers the
// translation units of the library call across one another (namespace trt_impl).  Nothing here is part of the C-ABI
// (include/trt_hip.h, include/trt_hip_diag.h).
//
//   trt_capi.hip    contexts, scene upload, settings                  (product API, section 2 of trt_hip.h)
//   trt_tables.hip  the candidate tables: marking and packing kernels, builders
//   trt_render.hip  render dispatch: the instantiations of the production kernel, occupancy, copy-out, kernel times
//   trt_diag.hip    counters read-out, table read-backs, self-tests, single-ray probes        (trt_hip_diag.h)
//   trt_dropin.hip  project_scene / render_frame and the default context                     (section 1 of trt_hip.h)
//   trt_dist.hip    one frame over the GPUs of a node                                        (section 3 of trt_hip.h)
#pragma once

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

namespace trt_impl
{

int fail(int code, const char *fmt, ...); // sets the thread's trt_last_error() text, returns code (trt_capi.hip)

inline double host_seconds()
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
inline double triangle_wave(double t)
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

} // namespace trt_impl

using trt_impl::DeviceBuffer;
using trt_impl::kEventRing;

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
    int big_blocks_per_cu = 0;     // ... and for the plain rounds in 1024-thread workgroups (render_rounds_kernel<.., BIG>; 0: not available)
    long last_units = 0;           // samples of the most recent launch (trt_render_variant / trt_kernel_info describe that launch's kernel)
    bool last_compact = false;     // the most recent launch ran the kernel with the shading decoupled
    bool last_big = false;         // ... the plain rounds in 1024-thread workgroups
    // what the queue of each lane set was last left ready for (workgroups, waves per workgroup, words' shift; 0 workgroups: nothing):
    // the ordered-mean pass of a frame starts the queue for the next one, which then needs no kernel of its own in front of it
    unsigned queue_ready[2][3] = {{0, 0, 0}, {0, 0, 0}};
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

namespace trt_impl
{

constexpr int kCompactionMinLights = 2; // trt_set_compaction(-1): decouple the shading from two lights up (with one it is a wash)

// LDS image of the production kernel for the context's scene and tables
inline size_t image_lds_bytes(const trt_context *ctx, int spp)
{
    return trt::rounds_lds_bytes(ctx->scene, spp, ctx->grids.path_enabled ? ctx->grids.patch_count : 0);
}

// LDS of render_rounds_kernel<.., false, true>: the image, then one shading ring per wave of the workgroup
inline size_t compact_ring_at(const trt_context *ctx, int spp)
{
    return (image_lds_bytes(ctx, spp) / sizeof(double) + 1) & ~(size_t)1; // in doubles, on a 16-byte boundary
}

inline size_t compact_lds_bytes(const trt_context *ctx, int spp)
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

inline bool renders_decoupled(const trt_context *ctx, long units)
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

inline size_t scene_lds_bytes(const trt::SceneView &s)
{
    return sizeof(double) * ((size_t)s.num_spheres * trt::kSphereDoubles + (size_t)s.num_dir * trt::kDirLightDoubles +
                             (size_t)s.num_point * trt::kPointLightDoubles);
}

inline double host_now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline bool print_host_times()
{
    static const bool on = getenv("TRT_PRINT_HOST_TIMES") != nullptr;
    return on;
}

// Does a frame run the plain rounds in 1024-thread workgroups (render_rounds_kernel<.., BIG>)?  When the scene has patches, no
// counters are asked for, and sixteen waves around ONE image are more than the 256-thread workgroups that fit the CU's LDS hold.
inline bool renders_big(const trt_context *ctx)
{
    return ctx->kernel == 0 && !ctx->ior_count && !ctx->counters_enabled && ctx->grids.path_enabled && ctx->grids.patch_m > 0 &&
           ctx->big_blocks_per_cu * trt::kBigBlock > ctx->rounds_blocks_per_cu * trt::kPersistentBlock;
}

inline bool rowset_valid(const trt_rowset *r)
{
    return r && r->width > 0 && r->height > 0 && r->tile_rows > 0 && r->tile_first >= 0 && r->tile_step > 0;
}

// ---- defined in trt_capi.hip
int upload_skybox(trt_context *ctx, const Skybox *sky);
unsigned long long skybox_stamp(const Skybox *sky);
// everything of the scene except camera and skybox.  per_call: the drop-in entries, which are handed the scene with every frame
int upload_primitives(trt_context *ctx, const Scene *scene, bool per_call = false);
int refuse_if_shared(const trt_context *ctx, const char *what);
extern int g_moving_after, g_still_after; // trt_set_scene_policy
// ---- defined in trt_tables.hip
int patches_for(const trt_context *ctx, int n);
int build_tables(trt_context *ctx, const trt_cull_scene &cs, const double *ground);
int ensure_eye_tables(trt_context *ctx, const Camera *camera, hipStream_t stream);
void allow_large_lds_tables(const trt_context *ctx); // dynamic LDS above the 64 KiB default needs the opt-in attribute, per kernel
// ---- defined in trt_render.hip
int refresh_occupancy(trt_context *ctx);
int prepare_jitter(trt_context *ctx, const Camera *cam, int width, int height, int spp);
int prepare_axes(trt_context *ctx, const Camera *cam, int width, int height);
void allow_large_lds_render(const trt_context *ctx);
// ---- defined in trt_diag.hip
void allow_large_lds_diag(const trt_context *ctx);

} // namespace trt_impl
