// trt_simple.hpp -- reference-order kernel: one lane per pixel, samples and bounces looped in the
// lane exactly as project_scene does (TRT.c:966-1069), every sphere tested exactly, no culling.
// It is the on-device parity anchor for the production kernel (trt_rounds.hpp), an independent implementation of the
// same path, and the code behind trt_probe_rays(..., production = 0).  Scene records are staged into LDS once per workgroup.
#pragma once

#include "trt_common.hpp"

namespace trt
{

struct Surface
{
    int what;     // ObjectType: 0 NONE, 1 SPHERE, 2 GROUND
    d3 point;     // nudged hit point, or the ray origin on a miss (TRT.c:860, :871-874)
    d3 normal;    // unit normal, or the unit ray direction on a miss (TRT.c:861, :878)
    d3 color;     // material colour (sky texel on a miss, TRT.c:866)
    double refl;  // material reflectivity (0 on a miss)
    double spec;  // carried only for trt_probe_rays
};

// LDS image of the scene: spheres (9 doubles each), then directional lights, then point lights
struct LdsScene
{
    const double *spheres;
    const double *dir_lights;
    const double *point_lights;
};

TRT_DEV LdsScene stage_scene(const SceneView &s, double *lds)
{
    const int ns = s.num_spheres * kSphereDoubles, nd = s.num_dir * kDirLightDoubles, np = s.num_point * kPointLightDoubles;
    for (int i = threadIdx.x; i < ns; i += blockDim.x)
        lds[i] = s.spheres[i];
    for (int i = threadIdx.x; i < nd; i += blockDim.x)
        lds[ns + i] = s.dir_lights[i];
    for (int i = threadIdx.x; i < np; i += blockDim.x)
        lds[ns + nd + i] = s.point_lights[i];
    __syncthreads();
    return LdsScene{lds, lds + ns, lds + ns + nd};
}

// TRT.c:793-889.  WANT_SURFACE=false is the shadow-ray form (normal/material NULL in the reference).
template <bool WANT_SURFACE>
TRT_DEV Surface closest_hit(const SceneView &s, const LdsScene &l, d3 o, d3 d)
{
    Surface best;
    best.what = 0;
    best.color = d3{0.0, 0.0, 0.0};
    best.refl = 0.0;
    best.spec = 0.0;
    double best_d2 = __builtin_inf();
    d3 best_point = o, best_normal = d;
    int best_index = -1;
    const double a = dot(d, d);

    for (int i = 0; i < s.num_spheres; i++)
    {
        const double *sp = l.spheres + i * kSphereDoubles;
        d3 c = load3(sp), p;
        if (hit_sphere(o, d, a, c, sp[3], p))
        {
            double d2 = dist2(o, p);
            if (d2 < best_d2) // strict: first index wins ties (TRT.c:816)
            {
                best.what = 1;
                best_d2 = d2;
                best_point = p;
                best_index = i;
            }
        }
    }
    if (best_index >= 0)
    {
        const double *sp = l.spheres + best_index * kSphereDoubles;
        best_normal = sub(best_point, load3(sp)); // TRT.c:824
        if (WANT_SURFACE)
        {
            best.color = load3(sp + 4);
            best.refl = sp[7];
            best.spec = sp[8];
        }
    }
    {
        d3 p;
        if (hit_plane(o, d, load3(s.ground), load3(s.ground + 3), p))
        {
            double d2 = dist2(o, p);
            if (d2 < best_d2)
            {
                best.what = 2;
                best_d2 = d2;
                best_point = p;
                best_normal = load3(s.ground + 3);
                if (WANT_SURFACE)
                {
                    const double *m = s.ground + (checker_odd(p) ? 11 : 6); // TRT.c:850-851
                    best.color = load3(m);
                    best.refl = m[3];
                    best.spec = m[4];
                }
            }
        }
    }
    if (best.what == 0)
    {
        if (WANT_SURFACE)
            best.color = texel_color(sky_texel(s.sky, s.sky_dim, d));
    }
    else
        best_point = nudge(o, best_point);
    best.point = best_point;
    best.normal = unit(best_normal);
    return best;
}

// TRT.c:894-963
TRT_DEV d3 lit_color(const SceneView &s, const LdsScene &l, d3 at, d3 normal, d3 albedo, unsigned &shadow_count)
{
    d3 out = d3{0.0, 0.0, 0.0};
    for (int i = 0; i < s.num_dir; i++)
    {
        const double *li = l.dir_lights + i * kDirLightDoubles;
        d3 to_light = unit(scale(load3(li), -1.0));
        shadow_count++;
        Surface blocker = closest_hit<false>(s, l, at, to_light);
        if (blocker.what == 0)
        {
            d3 diffuse = scale(load3(li + 3), min1(dot(normal, to_light)));
            out = add(out, mulc(diffuse, albedo));
        }
    }
    for (int i = 0; i < s.num_point; i++)
    {
        const double *li = l.point_lights + i * kPointLightDoubles;
        d3 to_light = sub(load3(li), at);
        double light_d2 = dot(to_light, to_light);
        double strength = clampd(li[6] / light_d2, 0.0, 1.0);
        to_light = unit(to_light);
        shadow_count++;
        Surface blocker = closest_hit<false>(s, l, at, to_light);
        d3 to_blocker = sub(blocker.point, at);
        double blocker_d2 = dot(to_blocker, to_blocker);
        if (blocker.what == 0 || light_d2 < blocker_d2)
        {
            d3 diffuse = scale(load3(li + 3), strength * min1(dot(normal, to_light)));
            out = add(out, mulc(diffuse, albedo));
        }
    }
    return d3{clampd(out.x, 0.0, 1.0), clampd(out.y, 0.0, 1.0), clampd(out.z, 0.0, 1.0)};
}

#ifdef TRT_UNIT_RENDER // a kernel that is not a template has ONE home among the library's translation units (trt_context.hpp)
__global__ __launch_bounds__(256) void render_simple_kernel(SceneView s, FrameView f)
{
    extern __shared__ double lds[];
    const LdsScene l = stage_scene(s, lds);

    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (long)f.local_rows * f.width)
        return;
    const int local_row = (int)(pix / f.width), column = (int)(pix - (long)local_row * f.width);
    const int row = frame_row_of(f, local_row);
    const d3 eye = load3(f.cam + 9);

    unsigned n_path = 0, n_shadow = 0;
    d3 mean = d3{0.0, 0.0, 0.0};
    for (int k = 0; k < f.spp; k++)
    {
        d3 dir = primary_direction(f, row, column, k);
        d3 org = eye;
        d3 sample = d3{0.0, 0.0, 0.0};
        int bounces = 0;
        double weight = 1.0, weight_sum = 0.0;
        bool going = true;
        while (going && bounces < f.bounce_limit && weight > 0.00001) // TRT.c:1018
        {
            n_path++;
            Surface hit = closest_hit<true>(s, l, org, dir);
            d3 color = hit.color;
            if (hit.what != 0)
                color = lit_color(s, l, hit.point, hit.normal, color, n_shadow);
            weight_sum += weight;
            color = scale(color, weight);
            if (hit.what != 0)
            {
                weight *= hit.refl;
                bounces++;
            }
            else
            {
                weight = 0.0;
                going = false;
            }
            sample = add(sample, color);
            dir = unit(reflect(dir, hit.normal));
            org = hit.point;
        }
        sample = scale(sample, 1.0 / weight_sum); // TRT.c:1061
        mean = add(mean, sample);
    }
    mean = scale(mean, 1.0 / f.spp); // TRT.c:1065
    double *o = f.out + pix * 3;
    o[0] = mean.x;
    o[1] = mean.y;
    o[2] = mean.z;
    if (f.counters)
    {
        atomicAdd(&f.counters[0], (unsigned long long)n_path);
        atomicAdd(&f.counters[1], (unsigned long long)n_shadow);
    }
}
#endif // TRT_UNIT_RENDER
#ifdef TRT_UNIT_DIAG

// trt_probe_rays: closest hit + lighting of arbitrary rays (tests)
__global__ __launch_bounds__(256) void probe_rays_kernel(SceneView s, const double *rays, long n, int *obj, double *point,
                                                          double *normal, double *material, double *lit)
{
    extern __shared__ double lds[];
    const LdsScene l = stage_scene(s, lds);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    d3 o = load3(rays + 6 * i), d = load3(rays + 6 * i + 3);
    Surface h = closest_hit<true>(s, l, o, d);
    obj[i] = h.what;
    point[3 * i + 0] = h.point.x, point[3 * i + 1] = h.point.y, point[3 * i + 2] = h.point.z;
    normal[3 * i + 0] = h.normal.x, normal[3 * i + 1] = h.normal.y, normal[3 * i + 2] = h.normal.z;
    material[5 * i + 0] = h.color.x, material[5 * i + 1] = h.color.y, material[5 * i + 2] = h.color.z;
    material[5 * i + 3] = h.refl, material[5 * i + 4] = h.spec;
    d3 c = d3{0.0, 0.0, 0.0};
    if (h.what != 0)
    {
        unsigned dummy = 0;
        c = lit_color(s, l, h.point, h.normal, h.color, dummy);
    }
    lit[3 * i + 0] = c.x, lit[3 * i + 1] = c.y, lit[3 * i + 2] = c.z;
}

__global__ void div_sqrt_kernel(const double *a, const double *b, long n, double *q, double *r)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        q[i] = a[i] / b[i];
        r[i] = __builtin_sqrt(a[i]);
    }
}

// unit() as the kernels use it (shared reciprocal, lean sqrt) next to the compiler's plain expansions; both also return
// the square root of the first component's magnitude
__global__ void unit_selftest_kernel(const double *v, long n, double *fast, double *reference)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        const d3 a = d3{v[4 * i], v[4 * i + 1], v[4 * i + 2]};
        const d3 f = unit(a), r = unit_reference(a);
        fast[4 * i] = f.x, fast[4 * i + 1] = f.y, fast[4 * i + 2] = f.z, fast[4 * i + 3] = sqrt_exact(v[4 * i + 3]);
        reference[4 * i] = r.x, reference[4 * i + 1] = r.y, reference[4 * i + 2] = r.z, reference[4 * i + 3] = __builtin_sqrt(v[4 * i + 3]);
    }
}

#endif // TRT_UNIT_DIAG

} // namespace trt
