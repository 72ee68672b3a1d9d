/* lightgrid_check.c -- host-side validation of the light-space candidate masks (csrc/trt_lightgrid.h).
 * Test helper only: compiled by tests/test_lightgrid.py with gcc -O2 -ffp-contract=off.
 * For every shadow ray it looks up the ray's cell exactly as the kernel does and compares with the EXACT
 * reference test (TRT.c:638-672, FP64, reference operation order) of every sphere:
 *   directional light: a sphere the exact test hits must be in the cell (a violation otherwise);
 *   point light: the same for every hit not farther than the light + near/2, and the lit/dark decision of
 *   TRT.c:936-946 taken from the cell's spheres alone must equal the decision taken from all spheres. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "trt_lightgrid.h"

typedef struct
{
    unsigned long long rays, far, exact_hits, candidates, violations, decision_mismatches, bits_set, cells;
    unsigned long long wave_max_cand, wave_groups;
    unsigned long long cand_hist[17];
    double first_violation[8]; /* ray(6), sphere index, cell */
} grid_stats;

static int exact_hit(const double *o, const double *d, double a, const double *s, double *t_out)
{
    const double ocx = o[0] - s[0], ocy = o[1] - s[1], ocz = o[2] - s[2];
    const double b = 2.0 * (ocx * d[0] + ocy * d[1] + ocz * d[2]);
    const double c = (ocx * ocx + ocy * ocy + ocz * ocz) - s[3] * s[3];
    const double disc = b * b - 4.0 * a * c;
    if (disc < 0.0)
        return 0;
    const double t0 = (-b - sqrt(disc)) / (2.0 * a);
    *t_out = t0;
    return t0 > 0.0;
}

static int in_cell(const unsigned long long *cell, int i) { return (cell[i >> 6] >> (63 - (i & 63))) & 1; }

static void note(grid_stats *st, const double *ray, int sphere, int cell)
{
    if (!st->violations)
    {
        memcpy(st->first_violation, ray, 6 * sizeof(double));
        st->first_violation[6] = sphere;
        st->first_violation[7] = cell;
    }
    st->violations++;
}

static void tally(grid_stats *st, unsigned cand, unsigned *group_max, size_t r, size_t n_rays)
{
    st->candidates += cand;
    st->cand_hist[cand > 16 ? 16 : cand]++;
    *group_max = cand > *group_max ? cand : *group_max;
    if ((r & 63) == 63 || r + 1 == n_rays)
    {
        st->wave_max_cand += *group_max;
        st->wave_groups++;
        *group_max = 0;
    }
}

/* all rays share the direction rays[3..5] (the unit to-light vector) */
void dirgrid_check(const double *spheres, int n, const double *rays, size_t n_rays, int g, int slabs, grid_stats *st)
{
    memset(st, 0, sizeof *st);
    if (!n_rays)
        return;
    const int padded = trt_cull_padded(n, 8);
    float *table = (float *)malloc(sizeof(float) * 4 * (size_t)(padded ? padded : 1));
    trt_cull_scene cs;
    trt_cull_build(spheres, n, 8, table, &cs);
    const int words = (n + 63) / 64 > 0 ? (n + 63) / 64 : 1;
    if (slabs < 1)
        slabs = 1;
    unsigned long long *masks = (unsigned long long *)malloc(sizeof(unsigned long long) * (size_t)slabs * g * g * words);
    trt_dirgrid G;
    trt_dirgrid_disc *discs = (trt_dirgrid_disc *)malloc(sizeof(trt_dirgrid_disc) * (size_t)(n ? n : 1));
    st->bits_set = (unsigned long long)trt_dirgrid_build(spheres, n, &cs, rays + 3, g, slabs, &G, masks, discs);
    free(discs);
    st->cells = (unsigned long long)slabs * g * g;
    /* the clamp relies on an empty border */
    for (int s = 0; s < slabs; s++)
        for (int j = 0; j < g; j++)
            for (int c = 0; c < g; c++)
                if (j == 0 || c == 0 || j == g - 1 || c == g - 1)
                    for (int w = 0; w < words; w++)
                        if (masks[(((size_t)s * g + j) * g + c) * words + w])
                            st->violations += 1000000;
    unsigned group_max = 0;
    for (size_t r = 0; r < n_rays; r++)
    {
        const double *o = rays + 6 * r, *d = o + 3;
        const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        int far;
        const int cell = trt_dirgrid_cell(&G, o[0], o[1], o[2], &far);
        st->rays++;
        if (far || !(fabs(a - 1.0) <= 9.094947017729282e-13))
        {
            st->far++;
            continue;
        }
        if (cell < 0 || cell >= slabs * g * g)
        {
            note(st, o, -1, cell);
            continue;
        }
        const unsigned long long *m = masks + (size_t)cell * words;
        unsigned cand = 0;
        for (int i = 0; i < n; i++)
        {
            double t;
            const int hit = exact_hit(o, d, a, spheres + 9 * i, &t);
            st->exact_hits += hit;
            cand += in_cell(m, i);
            if (hit && !in_cell(m, i))
                note(st, o, i, cell);
        }
        tally(st, cand, &group_max, r, n_rays);
    }
    free(masks);
    free(table);
}

/* squared distance from o to the hit point nudged back along the ray, formed as TRT.c:871-874 and :939-942 do */
static double nudged_d2(const double *o, const double *d, double t)
{
    const double p[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
    double back[3] = {o[0] - p[0], o[1] - p[1], o[2] - p[2]};
    const double l = sqrt(back[0] * back[0] + back[1] * back[1] + back[2] * back[2]);
    double q[3];
    for (int k = 0; k < 3; k++)
        q[k] = (p[k] + (back[k] / l) * 0.000001) - o[k];
    return q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
}

/* lit/dark from the closest hit among the spheres selected by `m` (NULL = all), TRT.c:808-820 and :936-946 */
static int decide_lit(const double *spheres, int n, const unsigned long long *m, const double *o, const double *d, double a, double light_d2)
{
    double best_d2 = INFINITY, best_t = 0.0;
    int best = -1;
    for (int i = 0; i < n; i++)
    {
        if (m && !in_cell(m, i))
            continue;
        double t;
        if (!exact_hit(o, d, a, spheres + 9 * i, &t))
            continue;
        const double p[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
        const double d2 = (o[0] - p[0]) * (o[0] - p[0]) + (o[1] - p[1]) * (o[1] - p[1]) + (o[2] - p[2]) * (o[2] - p[2]);
        if (d2 < best_d2)
            best_d2 = d2, best = i, best_t = t;
    }
    if (best < 0)
        return 1;
    return light_d2 < nudged_d2(o, d, best_t);
}

void pointgrid_check(const double *spheres, int n, const double *light, const double *rays, size_t n_rays, int g, int shells, grid_stats *st)
{
    memset(st, 0, sizeof *st);
    const int padded = trt_cull_padded(n, 8);
    float *table = (float *)malloc(sizeof(float) * 4 * (size_t)(padded ? padded : 1));
    trt_cull_scene cs;
    trt_cull_build(spheres, n, 8, table, &cs);
    const int words = (n + 63) / 64 > 0 ? (n + 63) / 64 : 1;
    if (shells < 1)
        shells = 1;
    unsigned long long *masks = (unsigned long long *)malloc(sizeof(unsigned long long) * 6 * (size_t)shells * g * g * words);
    trt_pointgrid G;
    trt_pointgrid_cone *cones = (trt_pointgrid_cone *)malloc(sizeof(trt_pointgrid_cone) * (size_t)(n ? n : 1));
    st->bits_set = (unsigned long long)trt_pointgrid_build(spheres, n, &cs, light, g, shells, &G, masks, cones);
    free(cones);
    st->cells = 6ull * shells * g * g;
    const double near = 0.02 + 4e-6 * sqrt((double)G.rg2);
    unsigned group_max = 0;
    for (size_t r = 0; r < n_rays; r++)
    {
        const double *o = rays + 6 * r, *d = o + 3;
        const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        int far;
        const int cell = trt_pointgrid_cell(&G, o[0], o[1], o[2], &far);
        st->rays++;
        if (far || !(fabs(a - 1.0) <= 9.094947017729282e-13))
        {
            st->far++;
            continue;
        }
        if (cell < 0 || cell >= 6 * shells * g * g)
        {
            note(st, o, -1, cell);
            continue;
        }
        const unsigned long long *m = masks + (size_t)cell * words;
        const double to_light[3] = {light[0] - o[0], light[1] - o[1], light[2] - o[2]};
        const double light_d2 = to_light[0] * to_light[0] + to_light[1] * to_light[1] + to_light[2] * to_light[2];
        const double light_d = sqrt(light_d2);
        unsigned cand = 0;
        for (int i = 0; i < n; i++)
        {
            double t;
            const int hit = exact_hit(o, d, a, spheres + 9 * i, &t);
            st->exact_hits += hit;
            cand += in_cell(m, i);
            if (hit && !in_cell(m, i) && t <= light_d + 0.5 * near)
                note(st, o, i, cell);
        }
        if (decide_lit(spheres, n, NULL, o, d, a, light_d2) != decide_lit(spheres, n, m, o, d, a, light_d2))
            st->decision_mismatches++;
        tally(st, cand, &group_max, r, n_rays);
    }
    free(masks);
    free(table);
}

/* TRT.c:677-695 */
static int plane_hit(const double *ground, const double *o, const double *d, double *t_out)
{
    const double *gp = ground, *gn = ground + 3;
    const double denom = d[0] * gn[0] + d[1] * gn[1] + d[2] * gn[2];
    if (!(fabs(denom) > 0.00001))
        return 0;
    const double t = ((gp[0] - o[0]) * gn[0] + (gp[1] - o[1]) * gn[1] + (gp[2] - o[2]) * gn[2]) / denom;
    *t_out = t;
    return t > 0.00001;
}

/* the reference's decision (TRT.c:805-853, :936-946) over all spheres and the ground (NULL: none) */
static int reference_lit(const double *spheres, int n, const double *ground, const double *o, const double *d, double a, double light_d2)
{
    double best_d2 = INFINITY, best_t = 0.0;
    int best = -1;
    for (int i = 0; i <= n; i++)
    {
        double t;
        if (i < n ? !exact_hit(o, d, a, spheres + 9 * i, &t) : !(ground && plane_hit(ground, o, d, &t)))
            continue;
        const double p[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
        const double d2 = (o[0] - p[0]) * (o[0] - p[0]) + (o[1] - p[1]) * (o[1] - p[1]) + (o[2] - p[2]) * (o[2] - p[2]);
        if (d2 < best_d2)
            best_d2 = d2, best = i, best_t = t;
    }
    if (best < 0)
        return 1;
    /* TRT.c:871-874: normalize_vector leaves vectors of length <= 1e-4 alone */
    const double p[3] = {o[0] + best_t * d[0], o[1] + best_t * d[1], o[2] + best_t * d[2]};
    double back[3] = {o[0] - p[0], o[1] - p[1], o[2] - p[2]};
    const double l = sqrt(back[0] * back[0] + back[1] * back[1] + back[2] * back[2]);
    if (l > 0.0001)
        back[0] /= l, back[1] /= l, back[2] /= l;
    double q[3];
    for (int k = 0; k < 3; k++)
        q[k] = (p[k] + back[k] * 0.000001) - o[k];
    return light_d2 < q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
}

typedef struct
{
    unsigned long long rays, far, dark, lit, unsure, wrong_dark, wrong_lit, tests, tests_closest;
    double first_wrong[6];
} anyhit_stats;

/* point_light_search of csrc/trt_rounds.hpp on the host: 0 dark, 1 lit, 2 unsure */
static int anyhit_class(const double *spheres, int n, const unsigned long long *m, const double *ground, const double *o, const double *d, double lo, double hi,
                        unsigned long long *tests)
{
    const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    int unsure = 0;
    for (int i = 0; i < n; i++)
    {
        if (!in_cell(m, i))
            continue;
        ++*tests;
        const double *s = spheres + 9 * i;
        const double ocx = o[0] - s[0], ocy = o[1] - s[1], ocz = o[2] - s[2];
        const double b = 2.0 * (ocx * d[0] + ocy * d[1] + ocz * d[2]);
        const double cc = (ocx * ocx + ocy * ocy + ocz * ocz) - s[3] * s[3];
        const double disc = b * b - 4.0 * a * cc;
        if (!(disc < 0.0) && b < 0.0)
        {
            const double q = -b - sqrt(disc), qq = q * q;
            const int blocks = q > TRT_SHADOW_QMIN && qq * TRT_SHADOW_K1 <= lo;
            if (blocks)
                return 0;
            unsure |= q > 0.0 && !(qq >= hi);
        }
    }
    if (ground)
    {
        const double *gp = ground, *gn = ground + 3;
        const double denom = d[0] * gn[0] + d[1] * gn[1] + d[2] * gn[2];
        if (fabs(denom) > 0.00001)
        {
            const double num = (gp[0] - o[0]) * gn[0] + (gp[1] - o[1]) * gn[1] + (gp[2] - o[2]) * gn[2];
            if (!((num < 0.0) != (denom < 0.0)) || num == 0.0) /* the kernel skips the division when the sign bits differ: t <= 0 then */
            {
                const double t = num / denom;
                if (t > 0.00001)
                {
                    const double qq = (t * t) * (4.0 * a) * a;
                    if (qq * TRT_SHADOW_K1 <= lo)
                        return 0;
                    unsure |= !(qq >= hi);
                }
            }
        }
    }
    return unsure ? 2 : 1;
}

void pointgrid_anyhit_check(const double *spheres, int n, const double *ground, const double *light, const double *rays, size_t n_rays, int g,
                            int shells, anyhit_stats *st)
{
    memset(st, 0, sizeof *st);
    const int padded = trt_cull_padded(n, 8);
    float *table = (float *)malloc(sizeof(float) * 4 * (size_t)(padded ? padded : 1));
    trt_cull_scene cs;
    trt_cull_build(spheres, n, 8, table, &cs);
    const int words = (n + 63) / 64 > 0 ? (n + 63) / 64 : 1;
    if (shells < 1)
        shells = 1;
    unsigned long long *masks = (unsigned long long *)malloc(sizeof(unsigned long long) * 6 * (size_t)shells * g * g * words);
    trt_pointgrid G;
    trt_pointgrid_cone *cones = (trt_pointgrid_cone *)malloc(sizeof(trt_pointgrid_cone) * (size_t)(n ? n : 1));
    trt_pointgrid_build(spheres, n, &cs, light, g, shells, &G, masks, cones);
    free(cones);
    for (size_t r = 0; r < n_rays; r++)
    {
        const double *o = rays + 6 * r, *d = o + 3;
        const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        int far;
        const int cell = trt_pointgrid_cell(&G, o[0], o[1], o[2], &far);
        st->rays++;
        if (far || !(fabs(a - 1.0) <= 9.094947017729282e-13) || cell < 0 || cell >= 6 * shells * g * g)
        {
            st->far++;
            continue;
        }
        const unsigned long long *m = masks + (size_t)cell * words;
        const double to_light[3] = {light[0] - o[0], light[1] - o[1], light[2] - o[2]};
        const double light_d2 = to_light[0] * to_light[0] + to_light[1] * to_light[1] + to_light[2] * to_light[2];
        double lo, hi;
        trt_point_shadow_bounds(&G, light_d2, a, &lo, &hi);
        const int cls = anyhit_class(spheres, n, m, ground, o, d, lo, hi, &st->tests);
        for (int i = 0; i < n; i++)
            st->tests_closest += (unsigned)in_cell(m, i);
        st->dark += cls == 0, st->lit += cls == 1, st->unsure += cls == 2;
        if (cls == 2)
            continue;
        const int ref = reference_lit(spheres, n, ground, o, d, a, light_d2);
        if (ref != cls)
        {
            if (!st->wrong_dark && !st->wrong_lit)
                memcpy(st->first_wrong, o, 6 * sizeof(double));
            st->wrong_dark += cls == 0, st->wrong_lit += cls == 1;
        }
    }
    free(masks);
    free(table);
}

/* The tables themselves, as the host reference builders make them (the GPU tests compare the device-built tables).
 * kind 0: directional light with to-light direction v, masks slabs*g*g*words; kind 1: point light at v, masks slabs*6*g*g*words
 * (`slabs` = slabs of depth resp. shells of distance, trt_lightgrid.h (5)). */
long lightgrid_host_table(const double *spheres, int n, int kind, const double *v, int g, int slabs, unsigned long long *masks)
{
    const int padded = trt_cull_padded(n, 8);
    float *table = (float *)malloc(sizeof(float) * 4 * (size_t)(padded ? padded : 1));
    trt_cull_scene cs;
    trt_cull_build(spheres, n, 8, table, &cs);
    free(table);
    long bits;
    if (kind == 0)
    {
        trt_dirgrid G;
        trt_dirgrid_disc *discs = (trt_dirgrid_disc *)malloc(sizeof(trt_dirgrid_disc) * (size_t)(n ? n : 1));
        bits = trt_dirgrid_build(spheres, n, &cs, v, g, slabs, &G, masks, discs);
        free(discs);
    }
    else
    {
        trt_pointgrid G;
        trt_pointgrid_cone *cones = (trt_pointgrid_cone *)malloc(sizeof(trt_pointgrid_cone) * (size_t)(n ? n : 1));
        bits = trt_pointgrid_build(spheres, n, &cs, v, g, slabs, &G, masks, cones);
        free(cones);
    }
    return bits;
}
