// trt_capi.hip -- contexts, scene upload, settings: section 2 of include/trt_hip.h (the render entries are in trt_render.hip).
// Compiled for gfx950 only, with -ffp-contract=off (see trt_device.hpp).
#include "trt_context.hpp"

using namespace trt_impl;

namespace trt_impl
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

// trt_set_scene_policy: a scene counts as moving from this many consecutive changed calls on, and as still again after this many unchanged ones
int g_moving_after = 2, g_still_after = 3;

// everything of the scene except camera and skybox.  per_call: the drop-in entries, which are handed the scene with every frame
int upload_primitives(trt_context *ctx, const Scene *scene, bool per_call)
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

} // namespace trt_impl

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
    HIP_TRY(ctx->d_queue.reserve(trt::kQueueWords));
    HIP_TRY(hipMemset(ctx->d_counters.ptr, 0, kCounterSlots * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(ctx->d_queue.ptr, 0, trt::kQueueWords * sizeof(unsigned int)));
    allow_large_lds_tables(ctx);
    allow_large_lds_render(ctx);
    allow_large_lds_diag(ctx);
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

namespace trt_impl
{
int refuse_if_shared(const trt_context *ctx, const char *what)
{
    if (ctx->T.use_count() > 1)
        return fail(TRT_ERR_ARGUMENT, "%s: this context's scene tables are shared with %ld other context(s) (trt_share_scene); "
                                      "change them before sharing, or give the context a scene of its own (trt_set_scene)", what, ctx->T.use_count() - 1);
    return TRT_OK;
}
} // namespace trt_impl

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

extern "C" int trt_render_variant(trt_context *ctx, int *decoupled, int *workgroup_threads)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    // the variant the most recent launch ran; before the first launch, what a whole large frame would run
    const bool d = ctx->have_scene && (ctx->last_units > 0 ? ctx->last_compact : renders_decoupled(ctx, kCompactionMinUnits));
    if (decoupled)
        *decoupled = d ? 1 : 0;
    if (workgroup_threads)
        *workgroup_threads = ctx->kernel == 1 ? 256 : (d ? trt::kCompactBlock : (ctx->last_units > 0 ? ctx->last_big : renders_big(ctx)) ? trt::kBigBlock : trt::kPersistentBlock);
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
    const int slabs_before = ctx->dirgrid_slabs, shells_before = ctx->pointgrid_shells;
    ctx->dirgrid_slabs = directional_slabs;
    ctx->pointgrid_shells = point_shells;
    const int rc = trt_set_light_grids(ctx, ctx->dirgrid_cells, ctx->pointgrid_cells);
    if (rc) // the tables were not rebuilt: the context keeps the settings its tables were built with
        ctx->dirgrid_slabs = slabs_before, ctx->pointgrid_shells = shells_before;
    return rc;
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
