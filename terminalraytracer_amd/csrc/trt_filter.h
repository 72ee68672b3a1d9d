/*
 * trt_filter.h -- conservative FP32 ray/sphere culling filter (host + device, plain C).
 *
 * The reference tests every sphere against every ray in FP64 (ray_intersects_sphere,
 * TRT.c:638-672): a sphere is hit iff  disc = b*b - 4*a*c >= 0  and  t0 = (-b - sqrt(disc))/(2a) > 0.
 * The production kernel first runs this cheap FP32 filter over all spheres and then performs
 * the EXACT FP64 test (reference operation order) only on the spheres the filter lets through,
 * in ascending index order.  The filter NEVER decides a hit: it may only discard spheres that
 * the exact test is guaranteed to reject, so the rendered image is bit-identical with or
 * without it.  (tests/test_filter.py checks "exact hit => filter passes" on millions of rays
 * with this very code compiled for the host.)
 *
 * Formulation.  With C = c - c0 (c0 = a per-scene shift: the centre of the sphere centres'
 * bounding box), O = o - c0, a = d.d and O' = O - (O.d) d, which is a point of the ray's LINE for
 * any a (the closest one to c0 when a = 1); kk = |C|^2 - r^2:
 *
 *      disc / (4a)  =  (b' - cd)^2 + 2 C.O' - kk - |O'|^2 ,   cd = (C.d)/sqrt(a),  b' = (O'.d)/sqrt(a)
 *      (o-c).d      =  O.d - C.d                                                   (exact algebra)
 *
 * Every ray the renderer traces has been normalised (TRT.c:1008, :1055, :904, :933), so
 * a = 1 +- a few ulp.  The filter therefore evaluates with a := 1 and b' := 0,
 *
 *      m = fma(cd,cd, C.W - thr) - kk,   n = cd - cd_min,     cd = C.d (FP32), W = 2 O'
 *      reject  <=>  m < 0  or  n < 0     (taken from the SIGN BITS of m and n: on gfx950 a compare
 *                                         costs as much as three VOP2 operations, a subtract does not)
 *
 * and a ray that is not "ok" -- |a - 1| > 2^-40 (a degenerate, un-normalisable direction), or bounds
 * that overflowed FP32 -- passes every sphere on to the exact test; for an ok ray no intermediate
 * can overflow (each is bounded by (Cn+Wn+Rm)^2, whose finiteness is part of ok), so m and n are
 * never NaN.  Per ray the set-up is 19 FP64 add/mul (no division, no sqrt),
 * 8 conversions and a dozen FP32 ops; per sphere 7 FP32 FMA/mul and 2 compares on a
 * scalar-loaded table {Cx,Cy,Cz,kk}.
 *
 * thr = |O'|^2 - E and cd_min = O.d - Eb, where E and Eb bound the total rounding error of the
 * FP32 evaluation PLUS the reference's own FP64 rounding PLUS the a := 1 approximation
 * (eps = 2^-24, u = 2^-53; Cn >= max|C|, Rm >= max r, Wn = |W|_1 >= |W|, On = |O|_1 >= |O|):
 *
 *      FP32 evaluation:   <= eps (20 Cn^2 + 8 Cn Wn + 6 Rm^2 + 2 Wn^2)   <=  20 eps (Cn + Wn + Rm)^2
 *                         (thr = |O'|^2 - E <= Wn^2/4 rides through the C.W chain: 5 roundings of it)
 *      a := 1, b' := 0:   <= 2^-39 (On + Cn)^2  =  2^14 u (On + Cn)^2
 *      FP64 (set-up, and the reference's own disc):  <= 64 u (On + Cn + Rm)^2
 *      E  = 40 eps (Cn + Wn + Rm)^2 + 2^16 u (On + Cn + Rm)^2          (>= 2x head-room on each)
 *      Eb = 32 eps (Cn + |O.d|) + 2^16 u (On + Cn + Rm)
 *
 * The second condition uses: b = 2 (o-c).d >= 0  =>  -b - sqrt(disc) <= 0  =>  t0 <= 0  =>  miss.
 */
#ifndef TRT_FILTER_H
#define TRT_FILTER_H

#if defined(__HIPCC__) || defined(__HIP__)
#define TRT_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define TRT_HD static inline
#endif

typedef struct
{
    float dx, dy, dz; /* d */
    float wx, wy, wz; /* W = 2 O' */
    float neg_thr;    /* -(|O'|^2 - E): start value of the C.W chain */
    float cd_min;     /* O.d - Eb */
    int ok;           /* 0: pass every sphere (degenerate ray or overflowed bounds) */
} trt_ray_filter;

/* per-scene constants of the culling table */
typedef struct
{
    double c0[3]; /* shift */
    float cn;     /* >= max_i |c_i - c0| (Euclidean), rounded up */
    float rm;     /* >= max_i r_i, rounded up */
} trt_cull_scene;

/* Build the culling table for n spheres given as the reference's 9-double Sphere records
 * (TRT.c:161-166: centre xyz, radius, material...).  table must hold 4*padded floats where
 * padded = n rounded up to a multiple of `group`; pad entries carry kk = 3e38 and never pass.
 * C = c - c0 with c0 the centre of the centres' bounding box; kk = |C|^2 - r^2; both are formed
 * in FP64 and rounded once. */
/* kk of a padding entry: large and FINITE (never passes; 0 * kk stays 0 in the MFMA form of the sweep) */
#define TRT_CULL_PAD_KK 3.0e38f

static inline int trt_cull_padded(int n, int group) { return (n + group - 1) / group * group; }

static inline void trt_cull_build(const double *spheres, int n, int group, float *table, trt_cull_scene *cs)
{
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++)
        {
            const double c = spheres[9 * i + j];
            lo[j] = (i == 0 || c < lo[j]) ? c : lo[j];
            hi[j] = (i == 0 || c > hi[j]) ? c : hi[j];
        }
    for (int j = 0; j < 3; j++)
        cs->c0[j] = 0.5 * (lo[j] + hi[j]);
    double cn = 0.0, rm = 0.0;
    for (int i = 0; i < n; i++)
    {
        const double C0 = spheres[9 * i] - cs->c0[0], C1 = spheres[9 * i + 1] - cs->c0[1], C2 = spheres[9 * i + 2] - cs->c0[2];
        const double r = spheres[9 * i + 3];
        const double c2 = C0 * C0 + C1 * C1 + C2 * C2;
        table[4 * i + 0] = (float)C0;
        table[4 * i + 1] = (float)C1;
        table[4 * i + 2] = (float)C2;
        table[4 * i + 3] = (float)(c2 - r * r);
        const double cl = __builtin_sqrt(c2), ra = __builtin_fabs(r);
        cn = cl > cn ? cl : cn;
        rm = ra > rm ? ra : rm;
    }
    const int padded = trt_cull_padded(n, group);
    for (int i = n; i < padded; i++)
    {
        table[4 * i + 0] = table[4 * i + 1] = table[4 * i + 2] = 0.0f;
        table[4 * i + 3] = TRT_CULL_PAD_KK;
    }
    /* rounded UP: the bounds must not be under-estimated */
    cs->cn = __builtin_nextafterf((float)(cn * (1.0 + 1e-6)), __builtin_inff());
    cs->rm = __builtin_nextafterf((float)(rm * (1.0 + 1e-6)), __builtin_inff());
}

#define TRT_EPS32 5.9604644775390625e-08f /* 2^-24 */
#define TRT_U64F 1.1102230246251565e-16f  /* 2^-53 as float */

TRT_HD void trt_filter_setup(trt_ray_filter *f, double ox, double oy, double oz, double dx, double dy, double dz, double a,
                             double c0x, double c0y, double c0z, float cn, float rm)
{
    const double Ox = ox - c0x, Oy = oy - c0y, Oz = oz - c0z;
    const double od = Ox * dx + Oy * dy + Oz * dz;
    const double Px = Ox - od * dx, Py = Oy - od * dy, Pz = Oz - od * dz; /* O' */
    f->dx = (float)dx;
    f->dy = (float)dy;
    f->dz = (float)dz;
    f->wx = (float)(2.0 * Px);
    f->wy = (float)(2.0 * Py);
    f->wz = (float)(2.0 * Pz);
    const float kc = (float)(Px * Px + Py * Py + Pz * Pz); /* |O'|^2 */
    const float wn = __builtin_fabsf(f->wx) + __builtin_fabsf(f->wy) + __builtin_fabsf(f->wz);
    const float on = (float)(__builtin_fabs(Ox) + __builtin_fabs(Oy) + __builtin_fabs(Oz));
    const float m32 = cn + wn + rm;
    const float m64 = on + cn + rm;
    const float E = 40.0f * TRT_EPS32 * m32 * m32 + 65536.0f * TRT_U64F * m64 * m64;
    const float odf = (float)od;
    const float Eb = 32.0f * TRT_EPS32 * (cn + __builtin_fabsf(odf)) + 65536.0f * TRT_U64F * m64;
    const int unit_ray = __builtin_fabs(a - 1.0) <= 9.094947017729282e-13; /* 2^-40; false for NaN */
    /* thr is pushed down by the rounding of kc itself and of the subtraction */
    f->neg_thr = -(kc - E - 4.0f * TRT_EPS32 * kc);
    f->cd_min = odf - Eb;
    f->ok = unit_ray && E < __builtin_inff() && Eb < __builtin_inff() && __builtin_fabsf(f->neg_thr) < __builtin_inff() &&
            __builtin_fabsf(f->cd_min) < __builtin_inff();
}

/* TRT_FILTER_BEHIND_TEST: 1 = the sweep also drops spheres whose centre lies behind the ray origin by more than
 * the error bound (n < 0; 11 VALU per sphere); 0 = only the discriminant test (9 VALU per sphere) and the exact
 * stage rejects those spheres (about twice the candidates, ~0.8 per ray instead of ~0.4). */
#ifndef TRT_FILTER_BEHIND_TEST
#define TRT_FILTER_BEHIND_TEST 0
#endif

/* sign word of one sphere for an ok ray: bit 31 set = reject */
TRT_HD unsigned trt_filter_sign(const trt_ray_filter *f, float cx, float cy, float cz, float kk)
{
    const float cd = __builtin_fmaf(cz, f->dz, __builtin_fmaf(cy, f->dy, cx * f->dx));
    const float cw = __builtin_fmaf(cz, f->wz, __builtin_fmaf(cy, f->wy, __builtin_fmaf(cx, f->wx, f->neg_thr)));
    const float m = __builtin_fmaf(cd, cd, cw) - kk;
    unsigned mb;
    __builtin_memcpy(&mb, &m, 4);
#if TRT_FILTER_BEHIND_TEST
    const float n = cd - f->cd_min;
    unsigned nb;
    __builtin_memcpy(&nb, &n, 4);
    mb |= nb;
#endif
    return mb;
}

/* Rays of ONE fixed direction (the shadow rays of a directional light, TRT.c:903-907): cd = C.d is the same for
 * every such ray, so it is folded into the table once, kk' = kk - cd*cd (FP32, two roundings), and the test
 * shrinks to m = (C.W - thr) - kk'.  Compared with trt_filter_sign this only moves cd^2 from one FMA into a
 * separately rounded product and difference: <= 3 eps Cn^2 more error, inside E's head-room. */
TRT_HD float trt_filter_fixed_dir_kk(float cx, float cy, float cz, float kk, float dx, float dy, float dz)
{
    const float cd = __builtin_fmaf(cz, dz, __builtin_fmaf(cy, dy, cx * dx));
    const float cd2 = cd * cd;
    return kk - cd2;
}

TRT_HD unsigned trt_filter_sign_fixed_dir(const trt_ray_filter *f, float cx, float cy, float cz, float kk_fixed)
{
    const float cw = __builtin_fmaf(cz, f->wz, __builtin_fmaf(cy, f->wy, __builtin_fmaf(cx, f->wx, f->neg_thr)));
    const float m = cw - kk_fixed;
    unsigned mb;
    __builtin_memcpy(&mb, &m, 4);
    return mb;
}

/* The same tests in the operation order of the MFMA-fed sweep (csrc/trt_rounds.hpp): v_mfma_f32_32x32x2_f32 is an
 * FMA chain over k = 0..3 (verified bit for bit by tools/mfma_probe), the 4th element of a sphere row is kk, the
 * 4th element of a ray column is 0 for the d-product and -1 for the W-product, and the W accumulator starts at
 * -thr:   cd = fma(kk,0, fma(Cz,dz, fma(Cy,dy, fma(Cx,dx, 0))));  cw = fma(kk,-1, fma(Cz,wz, fma(Cy,wy, fma(Cx,wx, -thr))));
 *         m = fma(cd,cd,cw).     Same error bound as trt_filter_sign: the same roundings in another order. */
TRT_HD unsigned trt_filter_sign_mfma(const trt_ray_filter *f, float cx, float cy, float cz, float kk)
{
    const float cd = __builtin_fmaf(kk, 0.0f, __builtin_fmaf(cz, f->dz, __builtin_fmaf(cy, f->dy, __builtin_fmaf(cx, f->dx, 0.0f))));
    const float cw = __builtin_fmaf(kk, -1.0f, __builtin_fmaf(cz, f->wz, __builtin_fmaf(cy, f->wy, __builtin_fmaf(cx, f->wx, f->neg_thr))));
    const float m = __builtin_fmaf(cd, cd, cw);
    unsigned mb;
    __builtin_memcpy(&mb, &m, 4);
    return mb;
}

TRT_HD unsigned trt_filter_sign_fixed_dir_mfma(const trt_ray_filter *f, float cx, float cy, float cz, float kk_fixed)
{
    const float m = __builtin_fmaf(kk_fixed, -1.0f, __builtin_fmaf(cz, f->wz, __builtin_fmaf(cy, f->wy, __builtin_fmaf(cx, f->wx, f->neg_thr))));
    unsigned mb;
    __builtin_memcpy(&mb, &m, 4);
    return mb;
}

TRT_HD int trt_filter_pass(const trt_ray_filter *f, float cx, float cy, float cz, float kk)
{
    return !f->ok || !(trt_filter_sign(f, cx, cy, cz, kk) >> 31);
}

#endif
