/*
 * trt_lightgrid.h -- light-space candidate masks for shadow rays (host + device, plain C).
 *
 * Every shadow ray of ONE light belongs to a two-parameter family: the rays towards a directional light
 * (TRT.c:903-908) all share one direction, the rays towards a point light (TRT.c:930-936) all pass through the
 * light's position.  For such a family "which spheres can this ray touch?" depends on two numbers only --
 * where the ray's line pierces the plane across the light direction, resp. in which direction it leaves the
 * light -- so it is tabulated once per scene:
 *
 *   directional light   a g x g grid over the plane through g0 spanned by e1, e2 (both across the light direction);
 *                       cell (i, j) holds one bit per sphere: set if the sphere's disc in that plane, grown by the
 *                       error bounds below, reaches the cell
 *   point light         a cube map of 6 x g x g cells of directions about the light's position; a cell holds the
 *                       spheres whose cone of directions (seen from the light, grown likewise) reaches the cell,
 *                       and every sphere closer to the light than `near`
 *
 * A shadow ray looks up ONE cell (about 20 FP32 operations and one load) instead of sweeping the whole culling
 * table, and hands the cell's spheres to the EXACT FP64 test in ascending index order, exactly like the sweep of
 * trt_filter.h does.  Like that filter the grid NEVER decides a hit: it may only leave out spheres whose exact
 * test cannot succeed -- or, for a point light, whose hit cannot change "is the light visible" (see (3)).
 * tests/test_lightgrid.py checks "exact hit => in the cell" on millions of rays with this very code on the host.
 *
 * Bounds (u = 2^-53, eps = 2^-24).  Only origins with |o - centre| <= rg are looked up (centre = g0 resp. the
 * light); any other origin, and any ray that is not normalised (|d.d - 1| > 2^-40), reports `far` and the caller
 * falls back to the sweep.
 *
 * (1) The reference's discriminant (TRT.c:644-652) equals 4a(r^2 - dist^2) up to its own rounding, dist = distance
 *     from the centre to the ray's line; a dozen FP64 operations on magnitudes <= 4.1 (|o-c| + r)^2 bound the
 *     error of dist^2 by 2^-48 (|o-c| + r)^2.  The tables use  rho^2 = r^2 + 2^-37 M^2,  M >= |o-c| + r  for every
 *     admissible origin: the exact test can only succeed if dist <= rho.
 * (2) FP32 look-up.  Directional: the two plane coordinates carry <= 10 eps rg of rounding (conversion of o - g0,
 *     two 3-term FMA chains, FP32 basis vs the exact orthonormal one) and the cell coordinate <= 2 eps (rg + g);
 *     the builder grows every disc by delta = 3e-6 rg and every cell by 0.01 cell.  Point: the face coordinates
 *     u = a / |major| come out of v_rcp_f32 with <= 4 eps relative error, and the ray's direction
 *     unit(light - o) misses the light by <= 2^-50 |light - o|; the builder grows every cone by 1e-5 rad, every
 *     cell by 0.01 cell and the face by 1e-4 rad past its edges (the look-up clamps into the edge cells, and near
 *     a cube edge either face may be chosen).
 * (3) Point light, spheres BEYOND the light.  A cell holds the spheres on the origin's side of the light and all
 *     spheres within `near` of it.  A sphere left out because it lies on the other side has every point >= near
 *     from the light, so a hit on it is >= near farther away than the light: "lit" whether or not it is the
 *     closest hit (TRT.c:939-946 compares the nudged hit distance with the light's), and any blocker in front of
 *     the light is closer than it.  near = 0.02 + 4e-6 rg keeps that decision out of reach of the 1e-6 nudge.
 */
#ifndef TRT_LIGHTGRID_H
#define TRT_LIGHTGRID_H

#include "trt_filter.h"

/* admissible origins: within TRT_LIGHTGRID_RANGE times the scene's reach (max |c - c0| + max r; for a point light plus
 * the light's distance from c0) of the look-up's centre.  Larger = fewer fall-backs to the sweep for far-away ground
 * points, but the FP32 look-up error delta, which every disc is grown by, scales with it. */
#ifndef TRT_LIGHTGRID_RANGE
#define TRT_LIGHTGRID_RANGE 256.0
#endif

typedef struct
{
    double g0[3];       /* centre of the look-up (the culling table's shift c0) */
    float e1[3], e2[3]; /* FP32 basis of the plane across the light direction */
    float u0, v0;       /* plane coordinates of the grid's lower corner */
    float inv_cell;     /* cells per unit length */
    float g_max;        /* g - 1 */
    float rg2;          /* admissible |o - g0|^2 */
    int g;              /* cells per side */
    int words;          /* 64-bit words per cell = ceil(n / 64) */
} trt_dirgrid;

typedef struct
{
    double l[3];  /* the light's position */
    float half_g; /* g / 2 */
    float g_max;  /* g - 1 */
    float rg2;    /* admissible |o - l|^2 */
    int g;        /* cells per face side */
    int words;
} trt_pointgrid;

/* cell index of the shadow ray that starts at o; *far != 0: do not use the grid for this ray */
TRT_HD int trt_dirgrid_cell(const trt_dirgrid *G, double ox, double oy, double oz, int *far)
{
    const float x = (float)(ox - G->g0[0]), y = (float)(oy - G->g0[1]), z = (float)(oz - G->g0[2]);
    const float r2 = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
    *far = !(r2 <= G->rg2); /* also true for NaN */
    const float u = __builtin_fmaf(z, G->e1[2], __builtin_fmaf(y, G->e1[1], x * G->e1[0]));
    const float v = __builtin_fmaf(z, G->e2[2], __builtin_fmaf(y, G->e2[1], x * G->e2[0]));
    float cu = (u - G->u0) * G->inv_cell, cv = (v - G->v0) * G->inv_cell;
    cu = __builtin_fminf(__builtin_fmaxf(cu, 0.0f), G->g_max);
    cv = __builtin_fminf(__builtin_fmaxf(cv, 0.0f), G->g_max);
    return (int)cv * G->g + (int)cu;
}

TRT_HD int trt_pointgrid_cell(const trt_pointgrid *G, double ox, double oy, double oz, int *far)
{
    const float w[3] = {(float)(ox - G->l[0]), (float)(oy - G->l[1]), (float)(oz - G->l[2])};
    const float r2 = __builtin_fmaf(w[2], w[2], __builtin_fmaf(w[1], w[1], w[0] * w[0]));
    *far = !(r2 <= G->rg2) || !(r2 > 0.0f);
    const float ax = __builtin_fabsf(w[0]), ay = __builtin_fabsf(w[1]), az = __builtin_fabsf(w[2]);
    /* major axis k; u along axis k+1, v along axis k+2 (cyclic) */
    float major, pu, pv;
    int face;
    if (ax >= ay && ax >= az)
        major = w[0], pu = w[1], pv = w[2], face = 0;
    else if (ay >= az)
        major = w[1], pu = w[2], pv = w[0], face = 2;
    else
        major = w[2], pu = w[0], pv = w[1], face = 4;
    face += major < 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    const float inv = __builtin_amdgcn_rcpf(__builtin_fabsf(major));
#else
    const float inv = 1.0f / __builtin_fabsf(major);
#endif
    float cu = __builtin_fmaf(pu * inv, G->half_g, G->half_g), cv = __builtin_fmaf(pv * inv, G->half_g, G->half_g);
    cu = __builtin_fminf(__builtin_fmaxf(cu, 0.0f), G->g_max);
    cv = __builtin_fminf(__builtin_fmaxf(cv, 0.0f), G->g_max);
    return (face * G->g + (int)cv) * G->g + (int)cu;
}

/* ------------------------------------------- builders (host) ------------------------------------------- */
/* sphere j of a chunk of 64 sits at bit 63 - j, the order the exact stage walks with count-leading-zeros */
static inline void trt_lightgrid_set(unsigned long long *cell, int sphere) { cell[sphere >> 6] |= 0x8000000000000000ull >> (sphere & 63); }

/* Directional light with unit to-light direction `to_light` (TRT.c:903-904).  `masks` must hold g*g*words words.
 * cs: the culling table's scene constants (shift and bounds).  Returns the number of (cell, sphere) bits set. */
static inline long trt_dirgrid_build(const double *spheres, int n, const trt_cull_scene *cs, const double to_light[3], int g,
                                     trt_dirgrid *G, unsigned long long *masks)
{
    const int words = (n + 63) / 64 > 0 ? (n + 63) / 64 : 1;
    const double reach = (double)cs->cn + (double)cs->rm;
    const double rg = TRT_LIGHTGRID_RANGE * reach + 1.0;
    const double M = rg + reach;
    const double E = 0x1p-37 * M * M, delta = 3e-6 * rg;
    /* orthonormal basis across the light direction, in double; the look-up uses its FP32 rounding */
    double d[3] = {to_light[0], to_light[1], to_light[2]};
    const double dl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int k = 0; k < 3; k++)
        d[k] /= dl;
    int thin = 0;
    for (int k = 1; k < 3; k++)
        if (fabs(d[k]) < fabs(d[thin]))
            thin = k;
    double t[3] = {0, 0, 0};
    t[thin] = 1.0;
    double e1[3] = {d[1] * t[2] - d[2] * t[1], d[2] * t[0] - d[0] * t[2], d[0] * t[1] - d[1] * t[0]};
    const double e1l = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    for (int k = 0; k < 3; k++)
        e1[k] /= e1l;
    const double e2[3] = {d[1] * e1[2] - d[2] * e1[1], d[2] * e1[0] - d[0] * e1[2], d[0] * e1[1] - d[1] * e1[0]};
    /* discs: centre in plane coordinates, radius rho + delta */
    double lo[2] = {0, 0}, hi[2] = {0, 0};
    for (int i = 0; i < n; i++)
    {
        const double *s = spheres + 9 * i;
        const double C[3] = {s[0] - cs->c0[0], s[1] - cs->c0[1], s[2] - cs->c0[2]};
        const double pu = C[0] * e1[0] + C[1] * e1[1] + C[2] * e1[2], pv = C[0] * e2[0] + C[1] * e2[1] + C[2] * e2[2];
        const double rad = sqrt(s[3] * s[3] + E) + delta;
        lo[0] = (i == 0 || pu - rad < lo[0]) ? pu - rad : lo[0];
        hi[0] = (i == 0 || pu + rad > hi[0]) ? pu + rad : hi[0];
        lo[1] = (i == 0 || pv - rad < lo[1]) ? pv - rad : lo[1];
        hi[1] = (i == 0 || pv + rad > hi[1]) ? pv + rad : hi[1];
    }
    /* square cells; two empty cells all round, so origins outside the discs' box clamp into empty cells */
    double extent = hi[0] - lo[0] > hi[1] - lo[1] ? hi[0] - lo[0] : hi[1] - lo[1];
    if (!(extent > 1e-9))
        extent = 1e-9;
    const double cell = extent / (g - 4);
    for (int k = 0; k < 3; k++)
    {
        G->g0[k] = cs->c0[k];
        G->e1[k] = (float)e1[k];
        G->e2[k] = (float)e2[k];
    }
    G->u0 = (float)(0.5 * (lo[0] + hi[0]) - 0.5 * g * cell);
    G->v0 = (float)(0.5 * (lo[1] + hi[1]) - 0.5 * g * cell);
    G->inv_cell = (float)(1.0 / cell);
    G->g_max = (float)(g - 1);
    G->rg2 = (float)(rg * rg * (1.0 - 1e-6));
    G->g = g;
    G->words = words;
    for (long i = 0; i < (long)g * g * words; i++)
        masks[i] = 0;
    long bits = 0;
    const double inv = (double)G->inv_cell, u0 = (double)G->u0, v0 = (double)G->v0; /* the look-up's own constants */
    for (int i = 0; i < n; i++)
    {
        const double *s = spheres + 9 * i;
        const double C[3] = {s[0] - cs->c0[0], s[1] - cs->c0[1], s[2] - cs->c0[2]};
        const double pu = ((C[0] * e1[0] + C[1] * e1[1] + C[2] * e1[2]) - u0) * inv;
        const double pv = ((C[0] * e2[0] + C[1] * e2[1] + C[2] * e2[2]) - v0) * inv;
        const double rad = (sqrt(s[3] * s[3] + E) + delta) * inv; /* in cells */
        int i0 = (int)floor(pu - rad - 0.01), i1 = (int)floor(pu + rad + 0.01);
        int j0 = (int)floor(pv - rad - 0.01), j1 = (int)floor(pv + rad + 0.01);
        i0 = i0 < 0 ? 0 : i0, j0 = j0 < 0 ? 0 : j0, i1 = i1 > g - 1 ? g - 1 : i1, j1 = j1 > g - 1 ? g - 1 : j1;
        for (int j = j0; j <= j1; j++)
            for (int c = i0; c <= i1; c++)
            {
                /* distance from the disc's centre to the cell grown by 0.01 */
                const double nx = pu < c - 0.01 ? c - 0.01 - pu : (pu > c + 1.01 ? pu - (c + 1.01) : 0.0);
                const double ny = pv < j - 0.01 ? j - 0.01 - pv : (pv > j + 1.01 ? pv - (j + 1.01) : 0.0);
                if (nx * nx + ny * ny <= rad * rad)
                {
                    trt_lightgrid_set(masks + ((long)j * g + c) * words, i);
                    bits++;
                }
            }
    }
    return bits;
}

/* [lo, hi] = the part of the arc [centre - half, centre + half] (angles, mod 2 pi) inside [-limit, limit]; returns 0 if none.
 * Two separate pieces are merged into their hull (conservative). */
static inline int trt_arc_clip(double centre, double half, double limit, double *lo, double *hi)
{
    int any = 0;
    for (int k = -1; k <= 1; k++)
    {
        const double a = centre - half + 6.283185307179586 * k, b = centre + half + 6.283185307179586 * k;
        const double l = a > -limit ? a : -limit, h = b < limit ? b : limit;
        if (l <= h)
        {
            *lo = (!any || l < *lo) ? l : *lo;
            *hi = (!any || h > *hi) ? h : *hi;
            any = 1;
        }
    }
    return any;
}

/* smallest angle between the unit vector a and the arc of the great circle from unit p to unit q (less than pi apart) */
static inline double trt_angle_to_arc(const double a[3], const double p[3], const double q[3])
{
    double nrm[3] = {p[1] * q[2] - p[2] * q[1], p[2] * q[0] - p[0] * q[2], p[0] * q[1] - p[1] * q[0]};
    const double nl = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    const double dp = a[0] * p[0] + a[1] * p[1] + a[2] * p[2], dq = a[0] * q[0] + a[1] * q[1] + a[2] * q[2];
    double best = acos(dp > 1.0 ? 1.0 : (dp < -1.0 ? -1.0 : dp));
    const double aq = acos(dq > 1.0 ? 1.0 : (dq < -1.0 ? -1.0 : dq));
    best = aq < best ? aq : best;
    if (nl > 1e-300)
    {
        for (int k = 0; k < 3; k++)
            nrm[k] /= nl;
        const double h = a[0] * nrm[0] + a[1] * nrm[1] + a[2] * nrm[2];
        const double f[3] = {a[0] - h * nrm[0], a[1] - h * nrm[1], a[2] - h * nrm[2]}; /* foot of a in the arc's plane */
        /* the foot lies on the arc iff it is on q's side of p and on p's side of q (within the plane) */
        const double cp[3] = {p[1] * f[2] - p[2] * f[1], p[2] * f[0] - p[0] * f[2], p[0] * f[1] - p[1] * f[0]};
        const double cq[3] = {f[1] * q[2] - f[2] * q[1], f[2] * q[0] - f[0] * q[2], f[0] * q[1] - f[1] * q[0]};
        if (cp[0] * nrm[0] + cp[1] * nrm[1] + cp[2] * nrm[2] >= 0.0 && cq[0] * nrm[0] + cq[1] * nrm[1] + cq[2] * nrm[2] >= 0.0)
        {
            const double ah = fabs(h) > 1.0 ? 1.0 : fabs(h);
            best = asin(ah) < best ? asin(ah) : best;
        }
    }
    return best;
}

/* smallest angle between the unit vector (au, av, am) (face frame, am along the face's axis) and the directions
 * (u, v, 1), u in [u0, u1], v in [v0, v1] */
static inline double trt_angle_to_cell(double au, double av, double am, double u0, double u1, double v0, double v1)
{
    if (am > 0.0 && au >= u0 * am && au <= u1 * am && av >= v0 * am && av <= v1 * am)
        return 0.0;
    const double a[3] = {au, av, am};
    const double cu[4] = {u0, u1, u1, u0}, cv[4] = {v0, v0, v1, v1};
    double c[4][3];
    for (int k = 0; k < 4; k++)
    {
        const double l = sqrt(cu[k] * cu[k] + cv[k] * cv[k] + 1.0);
        c[k][0] = cu[k] / l, c[k][1] = cv[k] / l, c[k][2] = 1.0 / l;
    }
    double best = 4.0;
    for (int k = 0; k < 4; k++)
    {
        const double t = trt_angle_to_arc(a, c[k], c[(k + 1) & 3]);
        best = t < best ? t : best;
    }
    return best;
}

/* Point light at `light` (TRT.c:930).  `masks` must hold 6*g*g*words words. */
static inline long trt_pointgrid_build(const double *spheres, int n, const trt_cull_scene *cs, const double light[3], int g, trt_pointgrid *G,
                                       unsigned long long *masks)
{
    const int words = (n + 63) / 64 > 0 ? (n + 63) / 64 : 1;
    const double reach = (double)cs->cn + (double)cs->rm;
    const double lc[3] = {light[0] - cs->c0[0], light[1] - cs->c0[1], light[2] - cs->c0[2]};
    const double away = sqrt(lc[0] * lc[0] + lc[1] * lc[1] + lc[2] * lc[2]); /* light to the scene's centre */
    const double rg = TRT_LIGHTGRID_RANGE * (reach + away) + 1.0;
    const double M = rg + away + reach;
    const double E = 0x1p-37 * M * M, near = 0.02 + 4e-6 * rg;
    const double grow = 1e-5, edge = 0.7853981633974483 + 1e-4; /* cone growth; face half-angle past its edges */
    for (int k = 0; k < 3; k++)
        G->l[k] = light[k];
    G->half_g = (float)(0.5 * g);
    G->g_max = (float)(g - 1);
    G->rg2 = (float)(rg * rg * (1.0 - 1e-6));
    G->g = g;
    G->words = words;
    const long cells = 6L * g * g;
    for (long i = 0; i < cells * words; i++)
        masks[i] = 0;
    long bits = 0;
    for (int i = 0; i < n; i++)
    {
        const double *s = spheres + 9 * i;
        const double a[3] = {s[0] - light[0], s[1] - light[1], s[2] - light[2]};
        const double D = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        const double rho = sqrt(s[3] * s[3] + E) + 0x1p-45 * M;
        if (!(D > rho + near) || !(rho / D < 0.999999))
        { /* the light is inside or next to the sphere: every ray may meet it */
            for (long c = 0; c < cells; c++)
                trt_lightgrid_set(masks + c * words, i);
            bits += cells;
            continue;
        }
        const double alpha = asin(rho / D) + grow, sa = sin(alpha);
        for (int face = 0; face < 6; face++)
        {
            const int k = face >> 1;
            const double sg = (face & 1) ? -1.0 : 1.0;
            /* the cone's axis in the face's frame: (u, v, major), major > 0 on the face */
            const double au = a[(k + 1) % 3] / D, av = a[(k + 2) % 3] / D, am = sg * a[k] / D;
            double range[2][2];
            int hit = 1;
            for (int c = 0; c < 2 && hit; c++)
            {
                const double ac = c == 0 ? au : av;          /* project along the other face axis: a wedge in the (ac, am) plane */
                const double len = sqrt(ac * ac + am * am);  /* sin(beta) = sin(alpha) / len */
                double lo = -edge, hi = edge;
                if (sa < len)
                    hit = trt_arc_clip(atan2(ac, am), asin(sa / len) + 1e-9, edge, &lo, &hi);
                range[c][0] = (tan(lo) + 1.0) * 0.5 * g;
                range[c][1] = (tan(hi) + 1.0) * 0.5 * g;
            }
            if (!hit)
                continue;
            int i0 = (int)floor(range[0][0] - 0.01), i1 = (int)floor(range[0][1] + 0.01);
            int j0 = (int)floor(range[1][0] - 0.01), j1 = (int)floor(range[1][1] + 0.01);
            i0 = i0 < 0 ? 0 : i0, j0 = j0 < 0 ? 0 : j0, i1 = i1 > g - 1 ? g - 1 : i1, j1 = j1 > g - 1 ? g - 1 : j1;
            /* inside the box of the two wedges: keep the cells the cone really reaches (cell grown by 0.01, and the
             * edge cells stretched past the face's edge, as the look-up clamps into them) */
            const double te = tan(edge), step = 2.0 / g;
            for (int j = j0; j <= j1; j++)
                for (int c = i0; c <= i1; c++)
                {
                    const double cu0 = c == 0 ? -te : -1.0 + (c - 0.01) * step, cu1 = c == g - 1 ? te : -1.0 + (c + 1.01) * step;
                    const double cv0 = j == 0 ? -te : -1.0 + (j - 0.01) * step, cv1 = j == g - 1 ? te : -1.0 + (j + 1.01) * step;
                    if (trt_angle_to_cell(au, av, am, cu0, cu1, cv0, cv1) <= alpha + 1e-9)
                    {
                        trt_lightgrid_set(masks + (((long)face * g + j) * g + c) * words, i);
                        bits++;
                    }
                }
        }
    }
    return bits;
}

#endif
