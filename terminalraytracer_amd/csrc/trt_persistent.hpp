// trt_persistent.hpp -- production frame-producer kernel for gfx950.
//
// Shape of the work: per pixel `spp` samples, per sample a bounce loop, per bounce
// one closest-hit trace plus one shadow trace per light; a trace = N sphere tests + 1 plane test.
// >95 % of the time is sphere tests, and path length varies 1..bounce_limit per sample.
//
// Design:
//   * PERSISTENT waves, ONE LANE = ONE PIXEL AT A TIME.  A lane runs its pixel's samples in
//     order (the mean over samples must be accumulated in the reference's order, TRT.c:1063)
//     and, when the pixel is finished, pulls the next pixel index from a global queue
//     (wave-aggregated atomic).  Lanes therefore never idle while pixels remain, however
//     different the path lengths of neighbouring pixels are.
//   * ONE TRACE LOOP FOR EVERY KIND OF RAY.  Path rays (TRT.c:1024) and the shadow rays of
//     lighting (TRT.c:907, :937) are the same closest-hit search, so the lane keeps a tiny
//     state machine (mode = PATH | SHADOW(light i)) and all 64 lanes of a wave -- whatever
//     their mode -- go through the sphere sweep together.  The reference's recursion
//     (apply_lighting -> trace_ray) becomes "re-enter the loop with the shadow ray".
//   * TWO-PHASE SPHERE SWEEP.  Phase 1: wave-uniform loop over a FP32 culling table read with
//     scalar loads (sphere index is uniform, so the table rides in SGPRs), 9 FP32 VALU ops per
//     sphere, result = per-lane 64-bit candidate mask (trt_filter.h; conservative, never decides
//     a hit).  Phase 2: each lane pops its own candidates in ascending index order and runs the
//     EXACT FP64 test in the reference's operation order against sphere records in LDS
//     (per-lane index -> LDS gather).  The divergent, expensive part (sqrt, divide) thus runs
//     max-over-lanes(candidates) ~ 2-4 times per trace instead of once per sphere.
//   * Scene records (spheres SoA, materials, lights) are staged once per workgroup into LDS;
//     the cubemap (6*dim*dim texels, 1.5 MB at 256^2, does not fit the 160 KB LDS) stays in
//     global memory / L2 and is touched once per sample.
//   * FP64 throughout, contraction off: results are bit-identical to the reference.
#pragma once

#include "trt_device.hpp"
#include "trt_filter.h"

namespace trt
{

constexpr int kPersistentBlock = 256;
constexpr int kCullGroup = 16; // culling-table entries fetched per scalar-load batch (table padded to this)

struct PersistentLaunch
{
    unsigned grid, block;
};

inline PersistentLaunch persistent_launch_shape(int compute_units, int blocks_per_cu, long pixels)
{
    long want = (pixels + kPersistentBlock - 1) / kPersistentBlock;
    long cap = (long)compute_units * (blocks_per_cu > 0 ? blocks_per_cu : 1);
    return PersistentLaunch{(unsigned)(want < cap ? (want > 0 ? want : 1) : cap), (unsigned)kPersistentBlock};
}

// LDS image: cx[n] cy[n] cz[n] r2[n] | mat[(n+2)*5] (spheres, ground even, ground odd) |
//            dir lights: unit to-light(3) colour(3) | point lights: pos(3) colour(3) intensity
inline size_t persistent_lds_bytes(const SceneView &s)
{
    return sizeof(double) * ((size_t)s.num_spheres * 4 + ((size_t)s.num_spheres + 2) * 5 + (size_t)s.num_dir * 6 + (size_t)s.num_point * 7);
}

struct CullView
{
    const float *table; // padded to a multiple of kCullGroup entries of {Cx,Cy,Cz,kk}
    int padded;
    double c0x, c0y, c0z;
    float cn, rm;
};

enum : int
{
    kDone = 0,
    kNeedPixel = 1,
    kNeedPrimary = 2,
    kTrace = 3
};

typedef const float __attribute__((address_space(4))) *const_float_ptr;

template <bool COUNT>
__global__ __launch_bounds__(kPersistentBlock) void render_persistent_kernel(SceneView s, CullView cull, FrameView f)
{
    extern __shared__ double lds[];
    const int n = s.num_spheres, nd = s.num_dir, np = s.num_point, nl = nd + np;
    double *const l_cx = lds, *const l_cy = l_cx + n, *const l_cz = l_cy + n, *const l_r2 = l_cz + n;
    double *const l_mat = l_r2 + n;
    double *const l_dir = l_mat + (n + 2) * 5;
    double *const l_pt = l_dir + nd * 6;

    for (int i = threadIdx.x; i < n; i += blockDim.x)
    {
        const double *sp = s.spheres + (long)i * kSphereDoubles;
        l_cx[i] = sp[0];
        l_cy[i] = sp[1];
        l_cz[i] = sp[2];
        l_r2[i] = sp[3] * sp[3]; // radius*radius exactly as TRT.c:648 forms it
        for (int j = 0; j < 5; j++)
            l_mat[i * 5 + j] = sp[4 + j];
    }
    for (int i = threadIdx.x; i < 10; i += blockDim.x)
        l_mat[n * 5 + i] = s.ground[6 + i]; // even material, odd material
    for (int i = threadIdx.x; i < nd; i += blockDim.x)
    {
        const double *li = s.dir_lights + i * kDirLightDoubles;
        d3 tl = unit(scale(load3(li), -1.0)); // TRT.c:903-904, the same value for every hit point
        l_dir[i * 6 + 0] = tl.x, l_dir[i * 6 + 1] = tl.y, l_dir[i * 6 + 2] = tl.z;
        l_dir[i * 6 + 3] = li[3], l_dir[i * 6 + 4] = li[4], l_dir[i * 6 + 5] = li[5];
    }
    for (int i = threadIdx.x; i < np * 7; i += blockDim.x)
        l_pt[i] = s.point_lights[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const long total = (long)f.local_rows * f.width;
    const d3 eye = load3(f.cam + 9);
    const d3 gp = load3(s.ground), gn = load3(s.ground + 3);
    const const_float_ptr table = (const_float_ptr)(uintptr_t)cull.table;

    // ---- per-lane state ------------------------------------------------------------------------
    int state = kNeedPixel;
    long pix = 0;
    int k = 0;                    // sample index within the pixel
    d3 mean = d3{0.0, 0.0, 0.0};  // sum over samples (average_pixel_color, TRT.c:977)
    d3 sample = d3{0.0, 0.0, 0.0}; // pixel_color of the current sample (TRT.c:1012)
    double weight = 1.0, weight_sum = 0.0;
    int bounces = 0;
    d3 o = eye, d = d3{0.0, 0.0, 1.0}; // the ray being traced
    int mode = 0;                      // 0: path ray; 1+i: shadow ray of light i (directional lights first)
    // surface found by the path ray, kept while its shadow rays are traced
    d3 h_point = eye, h_normal = d, path_dir = d, lit = d3{0.0, 0.0, 0.0};
    int h_mat = 0;                     // index into l_mat (sphere i, n = ground even, n+1 = ground odd)
    double light_d2 = 0.0, strength = 0.0;
    unsigned n_path = 0, n_shadow = 0;

    for (;;)
    {
        // ---- take new pixels from the queue (wave-aggregated) --------------------------------------
        const unsigned long long need = __ballot(state == kNeedPixel);
        if (need)
        {
            unsigned base = 0;
            const int leader = __builtin_ctzll(need);
            if (lane == leader)
                base = atomicAdd(f.queue, (unsigned)__builtin_popcountll(need));
            base = __shfl(base, leader);
            if (state == kNeedPixel)
            {
                const long mine = (long)base + __builtin_popcountll(need & ((1ull << lane) - 1ull));
                if (mine < total)
                {
                    pix = mine;
                    k = 0;
                    mean = d3{0.0, 0.0, 0.0};
                    state = kNeedPrimary;
                }
                else
                    state = kDone;
            }
        }
        if (!__any(state != kDone))
            break;

        // ---- primary ray of sample k (TRT.c:981-1016) ------------------------------------------------
        if (state == kNeedPrimary)
        {
            const int local_row = (int)(pix / f.width), column = (int)(pix - (long)local_row * f.width);
            d = primary_direction(f, frame_row_of(f, local_row), column, k);
            o = eye;
            sample = d3{0.0, 0.0, 0.0};
            weight = 1.0;
            weight_sum = 0.0;
            bounces = 0;
            mode = 0;
            state = kTrace;
        }
        const bool active = state == kTrace;

        // ---- closest hit (TRT.c:793-856) -----------------------------------------------------------
        const double a = dot(d, d);
        double best_d2 = __builtin_inf();
        d3 best_p = o;
        int best_i = -1; // sphere index, or n for the ground
        if (COUNT && active)
        {
            if (mode == 0)
                n_path++;
            else
                n_shadow++;
        }
        // a directional-light shadow ray only asks "anything hit?" (TRT.c:908): any hit ends its search
        const bool any_hit_suffices = mode != 0 && mode <= nd;

        trt_ray_filter flt;
        trt_filter_setup(&flt, o.x, o.y, o.z, d.x, d.y, d.z, a, cull.c0x, cull.c0y, cull.c0z, cull.cn, cull.rm);

        for (int base = 0; base < cull.padded; base += 64)
        {
            // phase 1: uniform sweep of up to 64 table entries, scalar-loaded
            unsigned long long cand = 0;
            const int chunk = (cull.padded - base) < 64 ? (cull.padded - base) : 64;
            for (int g = 0; g < chunk; g += kCullGroup)
            {
                unsigned bits = 0;
#pragma unroll
                for (int j = 0; j < kCullGroup; j++)
                {
                    const const_float_ptr e = table + (long)(base + g + j) * 4;
                    if (trt_filter_pass(&flt, e[0], e[1], e[2], e[3]))
                        bits |= 1u << j;
                }
                cand |= (unsigned long long)bits << g;
            }
            if (!active)
                cand = 0;
            // phase 2: exact FP64 tests of this lane's candidates, ascending index (first index wins ties)
            while (__any(cand != 0))
            {
                if (cand != 0)
                {
                    const int i = base + __builtin_ctzll(cand);
                    cand &= cand - 1;
                    if (i < n)
                    {
                        const d3 c = d3{l_cx[i], l_cy[i], l_cz[i]};
                        const d3 oc = sub(o, c);
                        const double b = 2.0 * dot(oc, d);
                        const double cc = dot(oc, oc) - l_r2[i];
                        const double disc = b * b - 4.0 * a * cc;
                        if (!(disc < 0.0))
                        {
                            const double t0 = (-b - __builtin_sqrt(disc)) / (2.0 * a);
                            if (t0 > 0.0)
                            {
                                const d3 p = d3{o.x + t0 * d.x, o.y + t0 * d.y, o.z + t0 * d.z};
                                const double d2 = dist2(o, p);
                                if (d2 < best_d2)
                                {
                                    best_d2 = d2;
                                    best_p = p;
                                    best_i = i;
                                }
                                if (any_hit_suffices)
                                    cand = 0;
                            }
                        }
                    }
                }
            }
        }
        // ground plane (TRT.c:831-853)
        if (active && !(any_hit_suffices && best_i >= 0))
        {
            d3 p;
            if (hit_plane(o, d, gp, gn, p))
            {
                const double d2 = dist2(o, p);
                if (d2 < best_d2)
                {
                    best_d2 = d2;
                    best_p = p;
                    best_i = n;
                }
            }
        }

        // ---- what the lane does with the result -----------------------------------------------------
        bool sample_done = false, lighting_done = false, next_shadow = false;
        int li = 0;
        if (active && mode == 0)
        {
            if (best_i < 0)
            { // sky (TRT.c:858-867, :1044-1048): colour = texel, reflectivity 0, the sample ends
                const d3 color = texel_color(sky_texel(s.sky, s.sky_dim, d));
                weight_sum += weight;
                sample = add(sample, scale(color, weight));
                sample_done = true;
            }
            else
            {
                h_point = nudge(o, best_p); // TRT.c:871-874
                if (best_i < n)
                {
                    h_normal = unit(sub(best_p, d3{l_cx[best_i], l_cy[best_i], l_cz[best_i]})); // TRT.c:824, :878
                    h_mat = best_i;
                }
                else
                {
                    h_normal = unit(gn);
                    h_mat = n + checker_odd(best_p); // TRT.c:850-851
                }
                path_dir = d;
                lit = d3{0.0, 0.0, 0.0};
                if (nl > 0)
                    next_shadow = true;
                else
                    lighting_done = true;
            }
        }
        else if (active)
        {
            // shadow ray of light `mode-1` came back (TRT.c:908-922 / :939-956); d is the unit vector to the light
            li = mode - 1;
            bool is_lit;
            double factor;
            d3 lcolor;
            if (li < nd)
            {
                is_lit = best_i < 0;
                factor = min1(dot(h_normal, d));
                lcolor = load3(l_dir + li * 6 + 3);
            }
            else
            {
                is_lit = best_i < 0;
                if (!is_lit)
                { // blocker farther than the light?  distance to the NUDGED blocker point (TRT.c:939-942)
                    const d3 to_blocker = sub(nudge(o, best_p), o);
                    is_lit = light_d2 < dot(to_blocker, to_blocker);
                }
                factor = strength * min1(dot(h_normal, d));
                lcolor = load3(l_pt + (li - nd) * 7 + 3);
            }
            if (is_lit)
                lit = add(lit, mulc(scale(lcolor, factor), load3(l_mat + h_mat * 5)));
            li++;
            if (li < nl)
                next_shadow = true;
            else
                lighting_done = true;
        }

        if (lighting_done)
        { // TRT.c:960-962 then :1034-1056
            d3 color = d3{clampd(lit.x, 0.0, 1.0), clampd(lit.y, 0.0, 1.0), clampd(lit.z, 0.0, 1.0)};
            weight_sum += weight;
            color = scale(color, weight);
            weight *= l_mat[h_mat * 5 + 3];
            bounces++;
            sample = add(sample, color);
            d = unit(reflect(path_dir, h_normal));
            o = h_point;
            mode = 0;
            if (!(bounces < f.bounce_limit && weight > 0.00001)) // TRT.c:1018
                sample_done = true;
        }
        if (next_shadow)
        { // shadow ray towards light li (TRT.c:903-907 / :929-937)
            if (li < nd)
                d = load3(l_dir + li * 6);
            else
            {
                const double *pl = l_pt + (li - nd) * 7;
                d3 to_light = sub(load3(pl), h_point);
                light_d2 = dot(to_light, to_light);
                strength = clampd(pl[6] / light_d2, 0.0, 1.0);
                d = unit(to_light);
            }
            o = h_point;
            mode = li + 1;
        }
        if (sample_done)
        { // TRT.c:1061-1066
            sample = scale(sample, 1.0 / weight_sum);
            mean = add(mean, sample);
            k++;
            if (k < f.spp)
                state = kNeedPrimary;
            else
            {
                mean = scale(mean, 1.0 / f.spp);
                double *out = f.out + pix * 3;
                out[0] = mean.x;
                out[1] = mean.y;
                out[2] = mean.z;
                state = kNeedPixel;
            }
        }
    }

    if (COUNT && f.counters)
    {
        atomicAdd(&f.counters[0], (unsigned long long)n_path);
        atomicAdd(&f.counters[1], (unsigned long long)n_shadow);
    }
}

} // namespace trt
