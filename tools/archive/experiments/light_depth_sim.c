/* light_depth_sim.c -- EXPERIMENT (round 4): how much shorter would the shadow rays' candidate lists be if the light tables had
 * a third coordinate -- depth along a directional light's direction, distance from a point light -- so that a cell lists only the
 * spheres that have some part between the origin's slab and the light?  And what would an any-hit search of the point light's
 * shadow rays save?  Host simulation on the oracle's shadow rays; lists are formed directly per ray.  Not product code. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "trt_lightgrid.h"

typedef struct
{
    unsigned long long rays, cand_now, cand_slab, tests_now, tests_any, tests_slab_any, wave_now, wave_any, wave_slab_any, wave_cand_slab, groups;
} depth_stats;

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

/* kind 1: directional (all rays share d), kind 2: point light at `light`; closest_now: today's search of kind-2 rays visits every candidate */
void depth_sim(const double *spheres, int n, const double *rays, const unsigned char *kinds, size_t n_rays, int kind, const double *light, int g,
               int slabs, depth_stats *st)
{
    memset(st, 0, sizeof *st);
    const int padded = trt_cull_padded(n, 8);
    float *table = (float *)malloc(sizeof(float) * 4 * (size_t)(padded ? padded : 1));
    trt_cull_scene cs;
    trt_cull_build(spheres, n, 8, table, &cs);
    trt_dirgrid G;
    trt_pointgrid Gp;
    trt_dirgrid_disc *discs = (trt_dirgrid_disc *)malloc(sizeof(trt_dirgrid_disc) * (size_t)n);
    trt_pointgrid_cone *cones = (trt_pointgrid_cone *)malloc(sizeof(trt_pointgrid_cone) * (size_t)n);
    double dir[3] = {0, 0, 0};
    int have = 0;
    double zlo = 1e300, zhi = -1e300;
    unsigned gm_now = 0, gm_any = 0, gm_slab = 0, gm_cs = 0;
    size_t seen = 0;
    for (size_t r = 0; r < n_rays; r++)
    {
        if (kinds[r] != kind)
            continue;
        const double *o = rays + 6 * r, *d = o + 3;
        if (!have)
        {
            have = 1;
            if (kind == 1)
            {
                memcpy(dir, d, sizeof dir);
                trt_dirgrid_prepare(spheres, n, &cs, dir, g, &G, discs);
                for (int i = 0; i < n; i++)
                {
                    const double *s = spheres + 9 * i;
                    const double z = (s[0] - cs.c0[0]) * dir[0] + (s[1] - cs.c0[1]) * dir[1] + (s[2] - cs.c0[2]) * dir[2];
                    zlo = z - fabs(s[3]) < zlo ? z - fabs(s[3]) : zlo;
                    zhi = z + fabs(s[3]) > zhi ? z + fabs(s[3]) : zhi;
                }
            }
            else
            {
                trt_pointgrid_prepare(spheres, n, &cs, light, g, &Gp, cones);
                zlo = 0.0;
                for (int i = 0; i < n; i++)
                {
                    const double *s = spheres + 9 * i;
                    const double D = sqrt((s[0] - light[0]) * (s[0] - light[0]) + (s[1] - light[1]) * (s[1] - light[1]) + (s[2] - light[2]) * (s[2] - light[2]));
                    zhi = D + fabs(s[3]) > zhi ? D + fabs(s[3]) : zhi;
                }
            }
        }
        st->rays++;
        const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        int far = 0, cell, cand = 0, cslab = 0, list[256], in_slab[256];
        double light_d2 = 0.0;
        if (kind == 1)
        {
            cell = trt_dirgrid_cell(&G, o[0], o[1], o[2], &far);
            const double zo = (o[0] - cs.c0[0]) * dir[0] + (o[1] - cs.c0[1]) * dir[1] + (o[2] - cs.c0[2]) * dir[2];
            const double h = (zhi - zlo) / slabs;
            const double slab_lo = zlo + floor((zo - zlo) / h) * h; /* the lowest depth an origin of this slab has */
            for (int i = 0; i < n; i++)
                if (trt_dirgrid_reaches(discs + i, cell % g, cell / g))
                {
                    const double *s = spheres + 9 * i;
                    const double z = (s[0] - cs.c0[0]) * dir[0] + (s[1] - cs.c0[1]) * dir[1] + (s[2] - cs.c0[2]) * dir[2];
                    in_slab[cand] = zo < zlo || z + fabs(s[3]) + 1e-3 >= slab_lo;
                    cslab += in_slab[cand];
                    list[cand++] = i;
                }
        }
        else
        {
            cell = trt_pointgrid_cell(&Gp, o[0], o[1], o[2], &far);
            const double w[3] = {light[0] - o[0], light[1] - o[1], light[2] - o[2]};
            light_d2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
            const double Ro = sqrt(light_d2), h = zhi / slabs;
            const double shell_hi = (floor(Ro / h) + 1.0) * h; /* the largest distance from the light an origin of this shell has */
            const int face = cell / (g * g), j = (cell / g) % g, c = cell % g;
            for (int i = 0; i < n; i++)
                if (trt_pointgrid_reaches(cones + i, face, c, j, g))
                {
                    const double *s = spheres + 9 * i;
                    const double D = sqrt((s[0] - light[0]) * (s[0] - light[0]) + (s[1] - light[1]) * (s[1] - light[1]) + (s[2] - light[2]) * (s[2] - light[2]));
                    in_slab[cand] = D - fabs(s[3]) - 1e-3 <= shell_hi;
                    cslab += in_slab[cand];
                    list[cand++] = i;
                }
        }
        /* tests: today (directional: until the first hit; point: every candidate), any-hit (until the first hit that is
         * nearer than the light), any-hit on the slab's list */
        int t_now = 0, t_any = 0, t_slab = 0, done_now = 0, done_any = 0, done_slab = 0;
        for (int k = 0; k < cand; k++)
        {
            double t;
            const int hit = exact_hit(o, d, a, spheres + 9 * list[k], &t);
            const int blocks = hit && (kind == 1 || t * t * a <= light_d2);
            if (!done_now)
                t_now++;
            if (kind == 1 && hit)
                done_now = 1;
            if (!done_any)
                t_any++;
            if (blocks)
                done_any = 1;
            if (in_slab[k] && !done_slab)
            {
                t_slab++;
                if (blocks)
                    done_slab = 1;
            }
        }
        st->cand_now += (unsigned)cand, st->cand_slab += (unsigned)cslab;
        st->tests_now += (unsigned)t_now, st->tests_any += (unsigned)t_any, st->tests_slab_any += (unsigned)t_slab;
        gm_now = (unsigned)t_now > gm_now ? (unsigned)t_now : gm_now;
        gm_any = (unsigned)t_any > gm_any ? (unsigned)t_any : gm_any;
        gm_slab = (unsigned)t_slab > gm_slab ? (unsigned)t_slab : gm_slab;
        gm_cs = (unsigned)cslab > gm_cs ? (unsigned)cslab : gm_cs;
        if ((++seen & 63) == 0)
        {
            st->wave_now += gm_now, st->wave_any += gm_any, st->wave_slab_any += gm_slab, st->wave_cand_slab += gm_cs;
            st->groups++;
            gm_now = gm_any = gm_slab = gm_cs = 0;
        }
    }
    free(cones);
    free(discs);
    free(table);
}
