/* filter_check.c -- host-side validation of the FP32 culling filter (csrc/trt_filter.h).
 * Test helper only: compiled by tests/test_filter.py with gcc -O2 -ffp-contract=off.
 * For every (ray, sphere) pair it evaluates the EXACT reference test (TRT.c:638-672, FP64,
 * reference operation order) and the filter, and reports any pair that the exact test hits
 * but the filter rejects (a violation: must be zero), plus candidate statistics. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "trt_filter.h"

typedef struct
{
    unsigned long long rays, pairs, exact_hits, line_hits, passed, violations;
    unsigned long long cand_hist[17]; /* candidates per ray, last bin = 16+ */
    unsigned long long hit_hist[17];  /* exact hits per ray */
    unsigned long long wave_max_cand; /* sum over groups of 64 consecutive rays of max candidates */
    unsigned long long wave_groups;
    double first_violation[8];        /* ray(6), sphere index, disc */
} filter_stats;

static int exact_hit(const double *o, const double *d, double a, const double *s, double *disc_out)
{
    const double ocx = o[0] - s[0], ocy = o[1] - s[1], ocz = o[2] - s[2];
    const double b = 2.0 * (ocx * d[0] + ocy * d[1] + ocz * d[2]);
    const double c = (ocx * ocx + ocy * ocy + ocz * ocz) - s[3] * s[3];
    const double disc = b * b - 4.0 * a * c;
    *disc_out = disc;
    if (disc < 0.0)
        return 0;
    const double t0 = (-b - sqrt(disc)) / (2.0 * a);
    return t0 > 0.0 ? 2 : 1; /* 2 = hit, 1 = line intersects but not ahead */
}

/* 0: the VALU sweep's operation order (trt_filter_sign); 1: the MFMA sweep's order (trt_filter_sign_mfma) */
int g_filter_order = 0;

void filter_check(const double *spheres, int n, const double *rays, size_t n_rays, filter_stats *st)
{
    memset(st, 0, sizeof *st);
    const int group = 16, padded = trt_cull_padded(n, group);
    float *table = (float *)malloc(sizeof(float) * 4 * (size_t)(padded ? padded : 1));
    trt_cull_scene cs;
    trt_cull_build(spheres, n, group, table, &cs);
    unsigned group_max = 0;
    for (size_t r = 0; r < n_rays; r++)
    {
        const double *o = rays + 6 * r, *d = o + 3;
        const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        trt_ray_filter f;
        trt_filter_setup(&f, o[0], o[1], o[2], d[0], d[1], d[2], a, cs.c0[0], cs.c0[1], cs.c0[2], cs.cn, cs.rm);
        unsigned cand = 0, hits = 0;
        for (int i = 0; i < padded; i++)
        {
            const int pass = g_filter_order == 0 ? trt_filter_pass(&f, table[4 * i], table[4 * i + 1], table[4 * i + 2], table[4 * i + 3])
                                                 : (!f.ok || !(trt_filter_sign_mfma(&f, table[4 * i], table[4 * i + 1], table[4 * i + 2], table[4 * i + 3]) >> 31));
            if (i >= n)
            {
                if (pass)
                    cand++; /* pad entries only pass on NaN/inf rays; the kernel guards i < n */
                continue;
            }
            double disc;
            const int e = exact_hit(o, d, a, spheres + 9 * i, &disc);
            st->pairs++;
            st->line_hits += e >= 1;
            st->exact_hits += e == 2;
            hits += e == 2;
            cand += pass != 0;
            st->passed += pass != 0;
            if (e == 2 && !pass)
            {
                if (!st->violations)
                {
                    memcpy(st->first_violation, o, 6 * sizeof(double));
                    st->first_violation[6] = i;
                    st->first_violation[7] = disc;
                }
                st->violations++;
            }
        }
        st->rays++;
        st->cand_hist[cand > 16 ? 16 : cand]++;
        st->hit_hist[hits > 16 ? 16 : hits]++;
        group_max = cand > group_max ? cand : group_max;
        if ((r & 63) == 63 || r + 1 == n_rays)
        {
            st->wave_max_cand += group_max;
            st->wave_groups++;
            group_max = 0;
        }
    }
    free(table);
}

/* Fixed-direction variant (directional-light shadow rays): every ray shares the direction of rays[3..5].
 * The table carries kk' = kk - (C.d)^2 exactly as the rounds kernel builds it. */
void filter_check_fixed_dir(const double *spheres, int n, const double *rays, size_t n_rays, filter_stats *st)
{
    memset(st, 0, sizeof *st);
    if (!n_rays)
        return;
    const int group = 16, padded = trt_cull_padded(n, group);
    float *table = (float *)malloc(sizeof(float) * 4 * (size_t)(padded ? padded : 1));
    trt_cull_scene cs;
    trt_cull_build(spheres, n, group, table, &cs);
    const float dx = (float)rays[3], dy = (float)rays[4], dz = (float)rays[5];
    for (int i = 0; i < padded; i++)
        table[4 * i + 3] = trt_filter_fixed_dir_kk(table[4 * i], table[4 * i + 1], table[4 * i + 2], table[4 * i + 3], dx, dy, dz);
    for (size_t r = 0; r < n_rays; r++)
    {
        const double *o = rays + 6 * r, *d = o + 3;
        const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        trt_ray_filter f;
        trt_filter_setup(&f, o[0], o[1], o[2], d[0], d[1], d[2], a, cs.c0[0], cs.c0[1], cs.c0[2], cs.cn, cs.rm);
        unsigned cand = 0, hits = 0;
        for (int i = 0; i < n; i++)
        {
            const unsigned sgn = g_filter_order == 0 ? trt_filter_sign_fixed_dir(&f, table[4 * i], table[4 * i + 1], table[4 * i + 2], table[4 * i + 3])
                                                     : trt_filter_sign_fixed_dir_mfma(&f, table[4 * i], table[4 * i + 1], table[4 * i + 2], table[4 * i + 3]);
            const int pass = !f.ok || !(sgn >> 31);
            double disc;
            const int e = exact_hit(o, d, a, spheres + 9 * i, &disc);
            st->pairs++;
            st->line_hits += e >= 1;
            st->exact_hits += e == 2;
            hits += e == 2;
            cand += pass != 0;
            st->passed += pass != 0;
            if (e == 2 && !pass)
            {
                if (!st->violations)
                {
                    memcpy(st->first_violation, o, 6 * sizeof(double));
                    st->first_violation[6] = i;
                    st->first_violation[7] = disc;
                }
                st->violations++;
            }
        }
        st->rays++;
        st->cand_hist[cand > 16 ? 16 : cand]++;
        st->hit_hist[hits > 16 ? 16 : hits]++;
    }
    free(table);
}
