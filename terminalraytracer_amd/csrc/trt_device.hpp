// trt_device.hpp -- device-side leaf math and scene views for the gfx950 frame producer.
//
// Every function here evaluates in IEEE binary64 in the SAME operation order as the reference
// (citations "TRT.c:N" = TerminalRayTracer.c line N).  The translation unit is compiled with
// -ffp-contract=off, so a*b+c is two roundings exactly as on the reference's x86-64 build;
// the only fused operations are the explicit __builtin_fmaf calls in the FP32 culling filter,
// which never decides a result (it only selects which spheres get the exact FP64 test).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace trt
{

struct d3
{
    double x, y, z;
};

#define TRT_DEV __device__ __forceinline__

TRT_DEV d3 make3(double x, double y, double z) { return d3{x, y, z}; }
TRT_DEV d3 load3(const double *p) { return d3{p[0], p[1], p[2]}; }
TRT_DEV double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // TRT.c:461 ((xx+yy)+zz)
TRT_DEV d3 sub(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
TRT_DEV d3 add(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
TRT_DEV d3 mulc(d3 a, d3 b) { return d3{a.x * b.x, a.y * b.y, a.z * b.z}; }
TRT_DEV d3 scale(d3 a, double s) { return d3{a.x * s, a.y * s, a.z * s}; }

// ---- square root and the three divisions of a normalisation, without the compiler's range handling ----
// hipcc expands an FP64 sqrt into v_rsq_f64 + 9 multiply/FMA steps wrapped in scaling for tiny arguments and a class test
// for 0/inf, and every FP64 division into two v_div_scale, v_rcp_f64, 7 FMA/mul steps, v_div_fmas and v_div_fixup.  For
// arguments whose exponents are nowhere near the ends of the range the wrappers do nothing: v_div_scale returns its
// operand and VCC = 0 (so v_div_fmas is a plain FMA), v_div_fixup passes a finite quotient through, the sqrt scaling is off.
// The functions below run the SAME arithmetic steps without the wrappers when EVERY active lane of the wave is inside
// that window, and the compiler's full sequence otherwise (one wave-uniform branch), so the results are those of
// `/` and __builtin_sqrt bit for bit; trt_selftest_unit compares them on the device.  TRT_LEAN_MATH=0 turns this off.
#ifndef TRT_LEAN_MATH
#define TRT_LEAN_MATH 1
#endif

// the short ways below are the ones taken: say so, so that they are laid out as the fall-through (two taken branches fewer each)
#ifndef TRT_OPT_LIKELY
#define TRT_OPT_LIKELY 1
#endif
#if TRT_OPT_LIKELY
#define TRT_LIKELY(c) __builtin_expect(!!(c), 1)
#else
#define TRT_LIKELY(c) (c)
#endif

TRT_DEV unsigned hi32(double x) { return (unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32); }
TRT_DEV unsigned lo32(double x) { return (unsigned)__builtin_bit_cast(unsigned long long, x); }

// positive, finite, biased exponent in [723, 1323): 2^-300 <= x < 2^300
TRT_DEV bool mid_range(double x) { return hi32(x) - (723u << 20) < (600u << 20); }

#ifndef TRT_LEAN_SQRT
#define TRT_LEAN_SQRT 1
#endif
#ifndef TRT_OPT_SQRT_FIXUP
#define TRT_OPT_SQRT_FIXUP 1 // the short way first, the compiler's sequence as a rare fix-up behind ONE forward branch (no diamond)
#endif
TRT_DEV double sqrt_exact(double x) // == __builtin_sqrt(x)
{
#if TRT_LEAN_MATH && TRT_LEAN_SQRT && TRT_OPT_SQRT_FIXUP
    // the steps of the compiler's expansion between its scaling and its 0/inf select, for every lane; a lane outside the window gets
    // garbage from them and, in the rare wave that has such a lane, the compiler's full sequence instead
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    double root = __builtin_fma(d, h, g);
    const bool outside = !mid_range(x);
    if (__builtin_expect(__any(outside), 0))
    {
        const double full = __builtin_sqrt(x);
        root = outside ? full : root;
    }
    return root;
#elif TRT_LEAN_MATH && TRT_LEAN_SQRT
    if (TRT_LIKELY(!__any(!mid_range(x))))
    { // the steps of the compiler's expansion between its scaling and its 0/inf select
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = y * 0.5;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        d = __builtin_fma(-g, g, x);
        return __builtin_fma(d, h, g);
    }
    return __builtin_sqrt(x);
#else
    return __builtin_sqrt(x);
#endif
}

// TRT.c:439-450: sqrt of the squared length, then THREE divisions, only when length > 1e-4
#ifndef TRT_LEAN_UNIT
#define TRT_LEAN_UNIT 2
#endif
TRT_DEV d3 unit(d3 a)
{
    const double len = sqrt_exact(a.x * a.x + a.y * a.y + a.z * a.z);
#if TRT_LEAN_MATH && TRT_LEAN_UNIT == 2
    // Round 4: no branch on the length and no copies.  A vector that is left alone (len <= 1e-4 or NaN) is "divided" by 1.0 --
    // x 1.0, a zero residual, + 0: the operand's own bits for every finite x -- so the three quotients are formed for every lane.
    // The short way is taken by the WAVE when, in every active lane, len < 2^300 (false for NaN / inf: then some component is
    // NaN / inf) and every component is zero or at least 2^-300 in magnitude (|a_k| <= len (1 + 2^-52) bounds them above); one
    // v_frexp_exp per component and a v_min3 decide that: the exponent of zero is 0, inside the window, and that of a denormal is
    // below it.  With len >= 1e-4 no quotient, product or residual leaves the normal range.
    const bool longer = len > 0.0001;
    const int ex = __builtin_amdgcn_frexp_exp(a.x), ey = __builtin_amdgcn_frexp_exp(a.y), ez = __builtin_amdgcn_frexp_exp(a.z);
    int lo = ex < ey ? ex : ey;
    lo = lo < ez ? lo : ez;
    const bool ok = len < 0x1p300 && lo > -300;
    if (TRT_LIKELY(!__any(!ok)))
    {
        const double den = longer ? len : 1.0;
        double r = __builtin_amdgcn_rcp(den);
        double e = __builtin_fma(-den, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-den, r, 1.0);
        r = __builtin_fma(r, e, r);
        double q[3] = {a.x, a.y, a.z};
#pragma unroll
        for (int k = 0; k < 3; k++)
        {
            const double num = q[k];
            const double q0 = num * r;
            const double err = __builtin_fma(-den, q0, num);
            const double quo = __builtin_fma(err, r, q0);
            q[k] = __builtin_copysign(quo, num); // v_div_fixup gives the quotient the sign of num/len; matters for +-0 only
        }
        return d3{q[0], q[1], q[2]};
    }
    if (longer)
    {
        a.x /= len;
        a.y /= len;
        a.z /= len;
    }
    return a;
#else
    if (len > 0.0001)
    {
#if TRT_LEAN_MATH
        // a numerator takes the short way if it is zero (the quotient is that zero, sign included) or mid-range in
        // magnitude; with len mid-range too, no quotient is near overflow or underflow
        const unsigned ax = hi32(a.x) & 0x7fffffffu, ay = hi32(a.y) & 0x7fffffffu, az = hi32(a.z) & 0x7fffffffu;
        const bool ok = mid_range(len) && (ax - (723u << 20) < (600u << 20) || (ax | lo32(a.x)) == 0) &&
                        (ay - (723u << 20) < (600u << 20) || (ay | lo32(a.y)) == 0) && (az - (723u << 20) < (600u << 20) || (az | lo32(a.z)) == 0);
        if (!__any(!ok))
        {
            // one reciprocal refinement for the three quotients (what the three expansions would each repeat)
            double r = __builtin_amdgcn_rcp(len);
            double e = __builtin_fma(-len, r, 1.0);
            r = __builtin_fma(r, e, r);
            e = __builtin_fma(-len, r, 1.0);
            r = __builtin_fma(r, e, r);
            double q[3] = {a.x, a.y, a.z};
#pragma unroll
            for (int k = 0; k < 3; k++)
            {
                const double num = q[k];
                const double q0 = num * r;
                const double err = __builtin_fma(-len, q0, num);
                const double quo = __builtin_fma(err, r, q0);
                q[k] = __builtin_copysign(quo, num); // v_div_fixup gives the quotient the sign of num/len; matters for +-0 only
            }
            return d3{q[0], q[1], q[2]};
        }
#endif
        a.x /= len;
        a.y /= len;
        a.z /= len;
    }
    return a;
#endif
}

// the compiler's own expansions, for trt_selftest_unit
TRT_DEV d3 unit_reference(d3 a)
{
    const double len = __builtin_sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    if (len > 0.0001)
    {
        a.x /= len;
        a.y /= len;
        a.z /= len;
    }
    return a;
}

// TRT.c:523-530 (NaN passes through, as in the reference)
TRT_DEV double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// TRT.c:627-633: v - ((2.0*dot)*n)
TRT_DEV d3 reflect(d3 v, d3 n)
{
    double d = dot(v, n);
    return d3{v.x - 2.0 * d * n.x, v.y - 2.0 * d * n.y, v.z - 2.0 * d * n.z};
}

// (int)x as the reference's x86-64 build computes it (cvttsd2si): truncation, and the
// "integer indefinite" 0x80000000 for NaN / out-of-range (gfx950's v_cvt_i32_f64 saturates instead).
TRT_DEV int d2i(double x)
{
    // |x| < 2^31: the truncation is exact.  Everything else -- NaN, x >= 2^31, x <= -2^31 -- is 0x80000000 on x86-64: the "integer
    // indefinite", which for -2^31 - 1 < x <= -2^31 is also the truncated value.  One compare (the absolute value is a free
    // operand modifier) instead of two.
    return __builtin_fabs(x) < 2147483648.0 ? (int)x : (int)0x80000000;
}

// fmin(x, 1.0) of TRT.c:911/945 (NaN -> 1.0)
TRT_DEV double min1(double x) { return __builtin_fmin(x, 1.0); }

// ---- scene views ---------------------------------------------------------------------------------
// Spheres/lights are the reference's AoS records viewed as doubles:
//   Sphere (TRT.c:161)  9 doubles: cx cy cz r | colour xyz | reflectivity | specularity
//   DirectionalLight (TRT.c:146) 6: dir xyz | colour xyz
//   PointLight (TRT.c:153) 7: pos xyz | colour xyz | intensity
//   Plane (TRT.c:169) 16: point | normal | even{colour,refl,spec} | odd{...}
constexpr int kSphereDoubles = 9;
constexpr int kDirLightDoubles = 6;
constexpr int kPointLightDoubles = 7;

struct SceneView
{
    const double *spheres; // global, AoS
    const double *dir_lights;
    const double *point_lights;
    const uint32_t *sky; // 6 faces x dim x dim texels, 0x00BBGGRR
    int num_spheres;
    int num_dir;
    int num_point;
    int sky_dim;
    double ground[16];
    double sky_dim_f; // (double)sky_dim, formed once on the host (TRT.c:727-728 convert it per look-up: the same value)
};

struct FrameView
{
    double cam[15];       // Camera (TRT.c:178): basis x,y,z | origin | screen_distance, screen_width, screen_height
    const double *jitter; // 2*spp doubles: x offsets then y offsets, already scaled by the pixel size (TRT.c:992-993)
    const double *col_x;  // [width]  (col/W)*screen_width - screen_width/2          (TRT.c:987), formed on the host
    const double *row_y;  // [height] -((row/H)*screen_height - screen_height/2)     (TRT.c:988), indexed by FRAME row
    double inv_spp;       // 1.0 / rays_per_pixel (TRT.c:1065)
    unsigned width_magic; // ceil(2^32 / width): pixel index -> row by multiply-high (persistent kernel)
    unsigned spp_magic;   // ceil(2^32 / spp): sample-unit index -> pixel
    unsigned tile_magic;  // ceil(2^32 / tile_rows): local row -> tile
    double *samples;      // [pixels*spp][3] per-sample colours when the work units are samples
    double *out;          // compact framebuffer of the owned rows
    const double *ior;    // refraction extension (render_rounds_kernel<.., true> only): per sphere, > 0 = index of refraction
    unsigned long long *counters; // [path, shadow] or nullptr
    unsigned int *queue;  // work-queue head for the persistent kernel
    unsigned ring_at;     // render_rounds_kernel<.., .., true>: where the waves' shading rings start in LDS, in doubles
    unsigned queue_shift; // render_rounds_kernel: the queue has 1 << queue_shift words (1, or one per XCD), see kQueueStride
    unsigned chunk;       // ... and hands out chunks of this many units
    int width, height;
    int tile_rows, tile_first, tile_step;
    int local_rows;
    int bounce_limit, spp;
};

// the same with the division as a multiply-high (f.tile_magic = min(ceil(2^32 / tile_rows), 2^32 - 1): off by at most one)
TRT_DEV int frame_row_of_magic(const FrameView &f, unsigned local_row)
{
    unsigned t = __umulhi(local_row, f.tile_magic);
    int within = (int)(local_row - t * (unsigned)f.tile_rows);
    const int under = within < 0, over = within >= f.tile_rows;
    t += (unsigned)(over - under);
    within += (under - over) * f.tile_rows;
    return (f.tile_first + (int)t * f.tile_step) * f.tile_rows + within;
}

TRT_DEV int frame_row_of(const FrameView &f, int local_row)
{
    int t = local_row / f.tile_rows; // tiles before the last are always full
    return (f.tile_first + t * f.tile_step) * f.tile_rows + (local_row - t * f.tile_rows);
}

// ---- exact primitive tests --------------------------------------------------------------------------
// TRT.c:638-672.  a = d.d is passed in (same value for every sphere of a ray).
TRT_DEV bool hit_sphere(d3 o, d3 d, double a, d3 c, double r, d3 &p)
{
    d3 oc = sub(o, c);
    double b = 2.0 * dot(oc, d);
    double cc = dot(oc, oc) - r * r;
    double disc = b * b - 4.0 * a * cc;
    if (disc < 0.0)
        return false;
    double t0 = (-b - __builtin_sqrt(disc)) / (2.0 * a);
    if (!(t0 > 0.0))
        return false;
    p = d3{o.x + t0 * d.x, o.y + t0 * d.y, o.z + t0 * d.z};
    return true;
}

// TRT.c:677-695
TRT_DEV bool hit_plane(d3 o, d3 d, d3 gp, d3 gn, d3 &p)
{
    double denom = dot(d, gn);
    if (!(__builtin_fabs(denom) > 0.00001))
        return false;
    const double num = dot(sub(gp, o), gn);
    // numerator and denominator of opposite sign: t <= 0 (or -0), a miss whatever the quotient's digits are.  Decided
    // from the sign bits so that a wave whose rays all head away from the plane skips the division (NaN passes on).
    if ((long long)(__builtin_bit_cast(unsigned long long, num) ^ __builtin_bit_cast(unsigned long long, denom)) < 0)
        return false;
    double t = num / denom;
    if (!(t > 0.00001))
        return false;
    p = d3{o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};
    return true;
}

// squared distance from the ray origin to a (rounded) hit point, TRT.c:810-815 / 834-839
TRT_DEV double dist2(d3 o, d3 p)
{
    d3 b = sub(o, p);
    return dot(b, b);
}

// TRT.c:871-874: p + unit(o - p)*1e-6
TRT_DEV d3 nudge(d3 o, d3 p) { return add(p, scale(unit(sub(o, p)), 0.000001)); }

// TRT.c:850: checker parity of a ground hit
TRT_DEV int checker_odd(d3 p) { return d2i(__builtin_floor(p.x) + __builtin_floor(p.z)) & 1; }

// ---- skybox, TRT.c:700-789 -----------------------------------------------------------------------------
// The reference evaluates everything through dot products with the axis table; multiplying by
// +-1/0 and adding zeros is kept (it can turn -0.0 into +0.0, and NaN/inf propagate the same way).
TRT_DEV d3 cube_axis(int f)
{
    double s = (f & 1) ? -1.0 : 1.0;
    int a = f >> 1;
    return d3{a == 0 ? s : 0.0, a == 1 ? s : 0.0, a == 2 ? s : 0.0};
}

TRT_DEV uint32_t sky_texel(const uint32_t *sky, int dim, d3 direction)
{
    d3 dir = unit(direction);
    int face = -1;
    double best = -1.0;
#pragma unroll
    for (int f = 0; f < 6; f++)
    {
        double t = dot(dir, cube_axis(f));
        if (t > best)
        {
            best = t;
            face = f;
        }
    }
    if (face < 0) // NaN direction: the reference indexes face -1 (undefined); defined here as face 0
        face = 0;
    d3 ax = cube_axis(face);
    d3 touching = mulc(dir, ax);
    double scale_by = touching.x + touching.y + touching.z;
    dir = scale(dir, 1.0 / scale_by);
    double along = dot(dir, ax);
    d3 in_plane = scale(sub(dir, scale(ax, along)), 0.5);
    double u = dot(in_plane, cube_axis((face + 2) % 6));
    double v = dot(in_plane, cube_axis((face + 4) % 6));
    if (face & 1)
        u *= -1.0;
    if (face < 2)
    {
        double t = u;
        u = v;
        v = -t;
    }
    else if (face < 4)
    {
        double t = u;
        u = -v;
        v = t;
    }
    else if (face == 4)
    {
        u *= -1.0;
        v *= -1.0;
    }
    u = clampd(u, -0.5, 0.5);
    v = clampd(v, -0.5, 0.5);
    int ui = d2i((u + 0.5) * dim);
    int vi = d2i((v + 0.5) * dim);
    long idx = (long)ui + (long)vi * dim; // same linear index as TRT.c:788 (ui == dim runs into the next row)
    long last = (long)dim * dim - 1;
    if (idx > last) // reference reads past the face here (undefined); defined as the last texel
        idx = last;
    if (idx < 0)
        idx = 0;
    return sky[(long)face * dim * dim + idx];
}

// Same lookup for an ALREADY normalised direction (the caller ran TRT.c:702's normalisation), with the
// axis-table algebra of TRT.c:705-727 carried out symbolically: multiplying by +-1 or 0 and adding the
// resulting zeros is exact, so t_f = +-component, scale_by = the winning t, and u, v are +-0.5 * a
// component of dir*(1/scale_by).  Only the sign of a zero can differ from the table form, and no
// later step (clamp, +0.5, *dim, truncation) can see it.  Assumes finite components.
TRT_DEV long sky_index_unit(int dim, d3 dir, double dim_f)
{
    int face = 0;
    double best = -1.0;
    const double t[6] = {dir.x, -dir.x, dir.y, -dir.y, dir.z, -dir.z};
#pragma unroll
    for (int f = 0; f < 6; f++)
        if (t[f] > best) // strict, first wins (TRT.c:708)
        {
            best = t[f];
            face = f;
        }
    const double inv = 1.0 / best; // TRT.c:718-719
    const d3 ds = scale(dir, inv);
    const int axis = face >> 1;
    const double sgn = (face & 1) ? -0.5 : 0.5;
    // u along axes[(face+2)%6], v along axes[(face+4)%6]: same parity of face, next two coordinate axes
    const double cu = axis == 0 ? ds.y : (axis == 1 ? ds.z : ds.x);
    const double cv = axis == 0 ? ds.z : (axis == 1 ? ds.x : ds.y);
    double u = sgn * cu, v = sgn * cv;
    if (face & 1)
        u = -u;
    if (face < 2)
    {
        const double w = u;
        u = v;
        v = -w;
    }
    else if (face < 4)
    {
        const double w = u;
        u = -v;
        v = w;
    }
    else if (face == 4)
    {
        u = -u;
        v = -v;
    }
    u = clampd(u, -0.5, 0.5);
    v = clampd(v, -0.5, 0.5);
    const int ui = d2i((u + 0.5) * dim_f);
    const int vi = d2i((v + 0.5) * dim_f);
    long idx = (long)ui + (long)vi * dim;
    const long last = (long)dim * dim - 1;
    idx = idx > last ? last : (idx < 0 ? 0 : idx);
    return (long)face * dim * dim + idx;
}

TRT_DEV uint32_t sky_texel_unit(const uint32_t *sky, int dim, d3 dir, double dim_f) { return sky[sky_index_unit(dim, dir, dim_f)]; }

TRT_DEV uint32_t sky_texel_unit(const uint32_t *sky, int dim, d3 dir)
{
    return sky_texel_unit(sky, dim, dir, (double)dim);
}

// The linear index (face dim^2 + vi dim + ui) of the same look-up from an FP32 ESTIMATE of the texel coordinates, with a proof
// that it is the reference's index -- or `ambiguous`, and then the caller takes the FP64 path above.
//   The reference forms U = (u + 0.5) dim, u = -+0.5 c / best (best = the largest |component|, c one of the other two; u and v
// per face as derived in sky_texel_unit), and truncates.  Here the face and the two in-plane components come from the cube
// instructions on the direction rounded to FP32 (v_cubeid / v_cubesc / v_cubetc / v_cubema: trt_cube_lookup in
// csrc/trt_lightgrid.h restates them; in the reference's frames u = -sc / |ma|, v = tc / |ma| on every face but -Y, where both
// signs flip), U' = fma(u', dim, dim / 2) in FP32.  Errors of U' against the real value: the roundings of the components
// (2^-24 each, relative to |u| <= 1/2), the reciprocal (1 ulp, 2^-23), a product (2^-24) and the fma (2^-24 dim): below
// 2^-21.9 dim; the reference's own U is within 2^-50 dim of the real value.  So with E = 2^-20 dim (3.7 x head-room) a U' that
// lies in (E, dim - E) and whose fractional part lies in (E, 1 - E) truncates to the reference's ui; everything else -- a
// coordinate next to a texel's edge, next to the face's edge (where the clamp of TRT.c:760 acts) or NaN (a direction that is
// not finite or not of FP32's range) -- is ambiguous.  A WRONG FACE needs two components whose magnitudes agree to 2^-23: then
// |c| / best is within 2^-22 of 1 and U' within E of 0 or dim: ambiguous.  dim >= 2^20 makes everything ambiguous, and so does a
// direction whose largest component is below 2^-100 in magnitude (the instructions flush FP32 denormals).
TRT_DEV long sky_index_estimate(int dim, float dim_f, d3 dir, bool &ambiguous)
{
    const float x = (float)dir.x, y = (float)dir.y, z = (float)dir.z;
    const int face = (int)__builtin_amdgcn_cubeid(x, y, z);
    const float sc = __builtin_amdgcn_cubesc(x, y, z), tc = __builtin_amdgcn_cubetc(x, y, z), ma = __builtin_amdgcn_cubema(x, y, z);
    const float inv = __builtin_amdgcn_rcpf(__builtin_fabsf(ma)); // 1 / (2 best)
    const unsigned flip = face == 3 ? 0x80000000u : 0u;
    const float un = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, sc) ^ flip ^ 0x80000000u) * inv; // u = -+sc / |ma|
    const float vn = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, tc) ^ flip) * inv;
    const float half = 0.5f * dim_f, e = 0x1p-20f * dim_f;
    const float U = __builtin_fmaf(un, dim_f, half), V = __builtin_fmaf(vn, dim_f, half);
    const float fu = __builtin_amdgcn_fractf(U), fv = __builtin_amdgcn_fractf(V);
    const float lo = __builtin_fminf(__builtin_fminf(U, V), __builtin_fminf(fu, fv)), hi = __builtin_fmaxf(U - dim_f + 1.0f, __builtin_fmaxf(V - dim_f + 1.0f, __builtin_fmaxf(fu, fv)));
    // The cube instructions flush FP32 denormals: a direction whose LARGEST component is below 2^-100 (an un-normalised vector that
    // TRT.c:444 left alone) may have lost a smaller one altogether while the reference's FP64 ratio keeps it -- ambiguous.  With the
    // largest component above that, a component that flushes is below 2^-25 of it: inside E.
    ambiguous = !(lo > e && hi < 1.0f - e) || !(__builtin_fabsf(ma) > 0x1p-100f); // NaN: ambiguous
    return ((long)face * dim + (long)(int)V) * dim + (long)(int)U;
}

#ifndef TRT_SKY_ESTIMATE
#define TRT_SKY_ESTIMATE 1 // 0: always the FP64 form (A/B)
#endif
// the reference's index for the lanes with `active`: the estimate, and the FP64 form for a wave in which some lane's is ambiguous
TRT_DEV long sky_index_unit(int dim, d3 dir, double dim_f);
TRT_DEV uint32_t sky_texel_wave(const uint32_t *sky, int dim, d3 dir, double dim_f, bool active)
{
#if TRT_SKY_ESTIMATE
    bool ambiguous;
    long idx = sky_index_estimate(dim, (float)dim_f, dir, ambiguous);
    if (__any(active && ambiguous))
    {
        const long exact = sky_index_unit(dim, dir, dim_f);
        idx = ambiguous ? exact : idx;
    }
    return active ? sky[idx] : 0u;
#else
    return active ? sky[sky_index_unit(dim, dir, dim_f)] : 0u;
#endif
}

TRT_DEV d3 texel_color(uint32_t t) // TRT.c:866: byte / 255.0
{
    return d3{(double)(t & 0xFF) / 255.0, (double)((t >> 8) & 0xFF) / 255.0, (double)((t >> 16) & 0xFF) / 255.0};
}

// ---- primary rays, TRT.c:981-1011 -------------------------------------------------------------------------
TRT_DEV d3 primary_direction(const FrameView &f, int row, int column, int k)
{
    const double sw = f.cam[13], sh = f.cam[14], sd = f.cam[12];
    double sx = (((double)column / (double)f.width) * sw - sw / 2.0);
    double sy = -(((double)row / (double)f.height) * sh - sh / 2.0);
    double sz = -sd;
    sx += f.jitter[k];
    sy += f.jitter[f.spp + k];
    d3 bx = load3(f.cam + 0), by = load3(f.cam + 3), bz = load3(f.cam + 6), eye = load3(f.cam + 9);
    d3 dir = d3{0.0, 0.0, 0.0};
    dir = add(dir, scale(bx, sx));
    dir = add(dir, scale(by, sy));
    dir = add(dir, scale(bz, sz));
    dir = sub(dir, eye); // sic, TRT.c:1005
    return unit(dir);
}

} // namespace trt
