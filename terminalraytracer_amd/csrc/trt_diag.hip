// trt_diag.hip -- include/trt_hip_diag.h: loop diagnostics, table read-backs, device self-tests, single-ray probes, test hooks.
// Nothing here is needed to produce a frame.
// Compiled for gfx950 only, with -ffp-contract=off (see trt_device.hpp).
#define TRT_UNIT_DIAG 1 // this unit is the home of the kernels that are not templates (trt_common.hpp, trt_simple.hpp)
#include "trt_context.hpp"
#include "trt_simple.hpp"

using namespace trt_impl;

namespace trt_impl
{
void allow_large_lds_diag(const trt_context *ctx)
{
    (void)hipFuncSetAttribute((const void *)trt::probe_rays_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::probe_rounds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
    (void)hipFuncSetAttribute((const void *)trt::probe_rounds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
}
} // namespace trt_impl

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

extern "C" int trt_read_sweep_fallbacks(trt_context *ctx, unsigned long long *swept_traces)
{
    if (!ctx || !swept_traces)
        return fail(TRT_ERR_ARGUMENT, "NULL argument");
    *swept_traces = ctx->last_swept;
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

