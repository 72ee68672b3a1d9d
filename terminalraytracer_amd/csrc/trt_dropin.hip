// trt_dropin.hip -- section 1 of include/trt_hip.h: project_scene (TRT.c:966) / render_frame and the default context behind them.
// Compiled for gfx950 only, with -ffp-contract=off (see trt_device.hpp).
#include "trt_context.hpp"

using namespace trt_impl;

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
// bit-identical either way.  trt_scene_is_moving() says whether the default context currently treats its scene as moving.
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

