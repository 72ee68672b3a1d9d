// trt_render.hip -- render dispatch: the instantiations of the production kernel (csrc/trt_rounds.hpp), occupancy, the copy-out to
// the host, kernel times and resource usage.
// Compiled for gfx950 only, with -ffp-contract=off (see trt_device.hpp).
#define TRT_UNIT_RENDER 1 // this unit is the home of the kernels that are not templates (trt_common.hpp, trt_simple.hpp)
#include "trt_context.hpp"
#ifndef TRT_QUEUE_PER_XCD
#define TRT_QUEUE_PER_XCD 1
#endif
#ifndef TRT_REDUCE_BLOCK
#define TRT_REDUCE_BLOCK 64 // one wave: fits beside the render kernels of the frames in flight wherever a wave retires (+0.8 % decoupled)
#endif
#include "trt_simple.hpp"

using namespace trt_impl;

namespace trt_impl
{

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
        ctx->big_blocks_per_cu = 0;
        if (ctx->rounds_blocks_per_cu < 4) // the image no longer fits four times: one image for sixteen waves instead
        {
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, trt::render_rounds_kernel<false, false, false, true, true>, trt::kBigBlock,
                                                                 image_lds_bytes(ctx, 64)));
            ctx->big_blocks_per_cu = blocks;
        }
    }
    return TRT_OK;
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

void allow_large_lds_render(const trt_context *ctx)
{
    (void)hipFuncSetAttribute((const void *)trt::render_simple_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<false, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::render_rounds_kernel<true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
}

} // namespace trt_impl

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
    f.queue = ctx->d_queue.ptr + trt::kQueueLaneWords * lane_set;
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
        const bool big = !compact && renders_big(ctx);
        if (big)
        {
            const long cap = (long)(ctx->compute_units - (ctx->stream == ctx->own_stream ? ctx->reserved_cus : 0)) * ctx->big_blocks_per_cu;
            const long want = (units + trt::kBigBlock - 1) / trt::kBigBlock;
            pl = trt::PersistentLaunch{(unsigned)std::max(1L, std::min(want, cap)), (unsigned)trt::kBigBlock};
        }
        ctx->last_big = big;
        const size_t plds = image_lds_bytes(ctx, rays_per_pixel);
        const dim3 grid(pl.grid), block(pl.block);
        if (ctx->ior_count && ctx->ior_count != ctx->scene.num_spheres) // before the first event of the launch is recorded
            return fail(TRT_ERR_ARGUMENT, "trt_set_refraction was given %d indices, the scene has %d spheres", ctx->ior_count, ctx->scene.num_spheres);
        ctx->last_compact = compact;
        // The queue (trt_common.hpp, kQueueStride): a word per XCD and chunks of half the size for scenes whose tables are small enough
        // that a wave may change its place in the image twice as often (no patches), when every word has workgroups; otherwise one
        // word.  Every wave owns its first chunk without asking.
        const bool per_xcd = TRT_QUEUE_PER_XCD && !(ctx->grids.path_enabled && ctx->grids.patch_m > 0) && pl.grid >= (1u << trt::kQueueXcdShift);
        f.queue_shift = per_xcd ? (unsigned)trt::kQueueXcdShift : 0u;
        f.chunk = per_xcd ? trt::kQueueChunkSmall : trt::kQueueChunkSamples;
        unsigned *const ready = ctx->queue_ready[lane_set];
        if (ready[0] != pl.grid || ready[1] != pl.block / 64 || ready[2] != f.queue_shift) // else the frame before left it ready (reduce_samples_kernel)
            hipLaunchKernelGGL(trt::start_queue_kernel, dim3(1), dim3(64), 0, stream, f.queue, pl.grid, pl.block / 64, f.queue_shift);
        ready[0] = 0; // the render kernel uses it up
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
        else if (patches && big)
            hipLaunchKernelGGL((trt::render_rounds_kernel<false, false, false, true, true>), grid, block, plds, stream, ctx->scene, ctx->cull, f, ctx->grids);
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
            hipLaunchKernelGGL(trt::reduce_samples_kernel, dim3((unsigned)((values + TRT_REDUCE_BLOCK - 1) / TRT_REDUCE_BLOCK)), dim3(TRT_REDUCE_BLOCK), 0, stream,
                               (const double *)scratch.ptr, (double *)d_pixels, values, rays_per_pixel, f.inv_spp, f.queue, pl.grid, pl.block / 64, f.queue_shift);
            ready[0] = pl.grid, ready[1] = pl.block / 64, ready[2] = f.queue_shift;
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

// trt_hip_diag.h: how many frames the context has launched, and the device time from the START of launch `first_launch` of context
// `first` to the END (render kernel and ordered mean) of launch `last_launch` of context `last` -- the span of a pipelined loop
// whose frames take turns over several contexts.  Launch numbers count from 0; the contexts keep the events of their last 256.
extern "C" long trt_launch_count(trt_context *ctx) { return ctx ? ctx->launches : -1; }

extern "C" int trt_launch_span_ms(trt_context *first, long first_launch, trt_context *last, long last_launch, float *ms)
{
    if (!first || !last || !ms || first->device != last->device)
        return fail(TRT_ERR_ARGUMENT, "bad argument");
    if (first_launch < 0 || first_launch >= first->launches || first->launches - first_launch > kEventRing || last_launch < 0 ||
        last_launch >= last->launches || last->launches - last_launch > kEventRing)
        return fail(TRT_ERR_ARGUMENT, "launches %ld / %ld are not among the last %d of their contexts (%ld / %ld launched)", first_launch, last_launch,
                    kEventRing, first->launches, last->launches);
    HIP_TRY(hipSetDevice(first->device));
    HIP_TRY(hipStreamSynchronize(first->stream));
    HIP_TRY(hipStreamSynchronize(last->stream));
    HIP_TRY(hipEventElapsedTime(ms, first->ev_start[first_launch % kEventRing], last->ev_stop[last_launch % kEventRing]));
    return TRT_OK;
}

extern "C" int trt_kernel_info(trt_context *ctx, int *vgprs, int *sgprs, int *static_lds_bytes, int *max_blocks_per_cu,
                               int *compute_units)
{
    if (!ctx)
        return fail(TRT_ERR_ARGUMENT, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    const bool decoupled = ctx->have_scene && (ctx->last_units > 0 ? ctx->last_compact : renders_decoupled(ctx, kCompactionMinUnits));
    const bool patches = ctx->have_scene && ctx->grids.path_enabled && ctx->grids.patch_m > 0;
    const bool big = !decoupled && renders_big(ctx);
    const void *fn = ctx->kernel == 1 ? (const void *)trt::render_simple_kernel
                     : decoupled      ? (const void *)trt::render_rounds_kernel<false, false, true>
                     : big            ? (const void *)trt::render_rounds_kernel<false, false, false, true, true>
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
        else if (big)
            blocks = ctx->big_blocks_per_cu; // likewise
        else
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, trt::render_rounds_kernel<false>, trt::kPersistentBlock, lds));
        *max_blocks_per_cu = blocks;
    }
    if (compute_units)
        *compute_units = ctx->compute_units;
    return TRT_OK;
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

