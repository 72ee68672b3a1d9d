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
 * Formulation.  With C = c - c0 (c0 = a per-scene shift, e.g. the centre of the spheres'
 * bounding box), O = o - c0, a = d.d, dh = d/sqrt(a), P = O - ((O.d)/a) d (the part of O
 * perpendicular to the ray), W = 2P, Kc = -|P|^2, kk = |C|^2 - r^2:
 *
 *      disc / (4a)  =  (C.dh)^2 + C.W - kk + Kc            (exact algebra)
 *      (o-c).d      =  O.d - sqrt(a) (C.dh)
 *
 * Per ray (FP64, then rounded to FP32): dh, W, Kc and two thresholds.  Per sphere (FP32, table
 * {Cx,Cy,Cz,kk}): two 3-term dot products by FMA and one more FMA, two compares.
 *
 *      pass  <=>  !( fma(cd,cd, C.W - kk) < thr )  &&  !( cd < cd_min )       cd = C.dh
 *
 * thr = -Kc - E and cd_min = ((O.d) - Eb)/sqrt(a) - slack, where E and Eb bound the total
 * rounding error of the FP32 evaluation PLUS the reference's own FP64 rounding
 * (eps = 2^-24, u = 2^-53; Cn >= max|C|, Rm >= max r, Wn = |W|_1 >= |W|, On = |O|_1 >= |O|):
 *
 *      |error of the FP32 left-hand side|  <=  eps (20 Cn^2 + 8 Cn Wn + 6 Rm^2 + Wn^2/2)   ("filter bound")
 *                                          <=  20 eps (Cn + Wn + Rm)^2
 *      E  = 40 eps (Cn + Wn + Rm)^2 + 256 u (On + Cn + Rm)^2          (2x head-room; FP64 terms)
 *      Eb = 32 eps (Cn + |O.d|/sqrt(a)) sqrt(a) ...                    (see code)
 *
 * Every comparison is written so that NaN/inf (a == 0, overflow) PASS the sphere on to the
 * exact test.  The second condition uses: b = 2 (o-c).d >= 0  =>  t0 <= 0  => miss.
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
    float dx, dy, dz; /* dh */
    float wx, wy, wz; /* W */
    float thr;        /* reject if fma(cd,cd,cwk) < thr */
    float cd_min;     /* reject if cd < cd_min */
} trt_ray_filter;

/* per-scene constants of the culling table */
typedef struct
{
    double c0[3]; /* shift */
    float cn;     /* >= max_i |c_i - c0| (Euclidean), rounded up */
    float rm;     /* >= max_i r_i, rounded up */
} trt_cull_scene;

#define TRT_EPS32 5.9604644775390625e-08f /* 2^-24 */
#define TRT_U64F 1.1102230246251565e-16f  /* 2^-53 as float */

TRT_HD void trt_filter_setup(trt_ray_filter *f, double ox, double oy, double oz, double dx, double dy, double dz, double a,
                             double c0x, double c0y, double c0z, float cn, float rm)
{
    const double Ox = ox - c0x, Oy = oy - c0y, Oz = oz - c0z;
    const double od = Ox * dx + Oy * dy + Oz * dz;
    const double s = 1.0 / a;
    const double t = od * s;
    const double Px = Ox - t * dx, Py = Oy - t * dy, Pz = Oz - t * dz;
    const double rs = __builtin_sqrt(s);
    f->dx = (float)(dx * rs);
    f->dy = (float)(dy * rs);
    f->dz = (float)(dz * rs);
    f->wx = (float)(2.0 * Px);
    f->wy = (float)(2.0 * Py);
    f->wz = (float)(2.0 * Pz);
    const float kc = (float)(Px * Px + Py * Py + Pz * Pz); /* -Kc */
    const float wn = __builtin_fabsf(f->wx) + __builtin_fabsf(f->wy) + __builtin_fabsf(f->wz);
    const float on = (float)(__builtin_fabs(Ox) + __builtin_fabs(Oy) + __builtin_fabs(Oz));
    const float m32 = cn + wn + rm;
    const float m64 = on + cn + rm;
    const float E = 40.0f * TRT_EPS32 * m32 * m32 + 256.0f * TRT_U64F * m64 * m64;
    /* thr = -Kc - E = kc - E, pushed down by the rounding of kc itself and of this subtraction */
    f->thr = kc - E - 4.0f * TRT_EPS32 * kc;
    /* cd_min: (o-c).d = od - cd/rs  <= Eb  <=>  cd >= (od - Eb) rs */
    const float odr = (float)(od * rs);
    const float Eb = 32.0f * TRT_EPS32 * (cn + __builtin_fabsf(odr)) + 256.0f * TRT_U64F * m64;
    f->cd_min = odr - Eb;
}

TRT_HD int trt_filter_pass(const trt_ray_filter *f, float cx, float cy, float cz, float kk)
{
    const float cd = __builtin_fmaf(cz, f->dz, __builtin_fmaf(cy, f->dy, cx * f->dx));
    const float cwk = __builtin_fmaf(cz, f->wz, __builtin_fmaf(cy, f->wy, __builtin_fmaf(cx, f->wx, -kk)));
    const float lhs = __builtin_fmaf(cd, cd, cwk);
    return !(lhs < f->thr) && !(cd < f->cd_min);
}

#endif
