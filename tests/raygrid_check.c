/* raygrid_check.c -- host-side validation of the path-ray candidate tables (csrc/trt_raygrid.h).
 * Test helper only: compiled by tests/test_raygrid.py with gcc -O2 -ffp-contract=off -fopenmp.
 *
 * Builds the tables of all 2 + 2NP families (P patches per sphere, trt_raygrid.h) with the host reference builder, packs
 * them into list cells exactly as the library does, and walks path rays in trace order the way the kernel does: a ray that
 * starts at the eye belongs to family 0; otherwise the family follows from what the PREVIOUS path ray hit (sphere i -> the
 * patch of sphere i its origin lies on, ground -> the mirror family of the parent's).  For every ray that passes the run-time membership test of its family, every sphere the
 * EXACT reference test hits (TRT.c:638-672, FP64, reference operation order) must be in the list of the ray's cell.
 * `brute` additionally tests the ray against EVERY family whose membership test it passes (the tables must be
 * conservative for any member ray, wherever it came from). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "trt_raygrid.h"

typedef struct
{
    unsigned long long rays, members, non_members, exact_hits, candidates, violations, brute_pairs, brute_violations;
    unsigned long long wave_max_cand, wave_groups, pooled_cells, none_cells, pool_words, cells, bits_set;
    unsigned long long by_family[4]; /* eye, mirror eye, sphere, mirror sphere */
    unsigned long long cand_hist[17];
    double first_violation[8]; /* ray(6), sphere, family */
} ray_stats;

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

/* closest hit of TRT.c:805-853: returns -1 nothing, i sphere, n ground */
static int closest(const double *spheres, int n, const double *ground, const double *o, const double *d)
{
    const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double best = INFINITY;
    int hit = -1;
    for (int i = 0; i < n; i++)
    {
        double t;
        if (!exact_hit(o, d, a, spheres + 9 * i, &t))
            continue;
        const double p[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
        const double d2 = (o[0] - p[0]) * (o[0] - p[0]) + (o[1] - p[1]) * (o[1] - p[1]) + (o[2] - p[2]) * (o[2] - p[2]);
        if (d2 < best)
            best = d2, hit = i;
    }
    const double *gp = ground, *gn = ground + 3;
    const double denom = d[0] * gn[0] + d[1] * gn[1] + d[2] * gn[2];
    if (fabs(denom) > 0.00001)
    {
        const double t = ((gp[0] - o[0]) * gn[0] + (gp[1] - o[1]) * gn[1] + (gp[2] - o[2]) * gn[2]) / denom;
        if (t > 0.00001)
        {
            const double p[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
            const double d2 = (o[0] - p[0]) * (o[0] - p[0]) + (o[1] - p[1]) * (o[1] - p[1]) + (o[2] - p[2]) * (o[2] - p[2]);
            if (d2 < best)
                hit = n;
        }
    }
    return hit;
}

typedef struct
{
    int n, g_eye, g_sph, families;
    trt_patchset patches; /* the sub-families of every sphere (trt_raygrid.h): P = patches.count tables per sphere, and P mirrored */
    trt_rayfamily *fam;
    unsigned long long *cells; /* list cells: 2 * 6 g_eye^2, then 2 n P * 6 g_sph^2 */
    unsigned long long *pool;
    size_t pool_words, pool_cap;
    int bits; /* entry width of the list cells: 8 up to 256 spheres, 16 above (trt_raygrid.h) */
} tables;

static size_t family_base(const tables *T, int f)
{
    const size_t ce = 6 * (size_t)T->g_eye * T->g_eye, cs = 6 * (size_t)T->g_sph * T->g_sph;
    return f < 2 ? (size_t)f * ce : 2 * ce + (size_t)(f - 2) * cs;
}

/* the families of a scene in the library's order: eye, mirror eye, n P patches of the spheres, n P mirror images of them */
void raygrid_families(const double *spheres, int n, const double *ground, const double *eye, int patch_m, trt_rayfamily *fam)
{
    trt_patchset P;
    trt_patchset_init(&P, patch_m);
    trt_family_consts consts;
    const int padded = trt_cull_padded(n, 8);
    float *table = (float *)malloc(sizeof(float) * 4 * (size_t)(padded ? padded : 1));
    trt_cull_scene cs;
    trt_cull_build(spheres, n, 8, table, &cs);
    free(table);
    trt_eye_families(eye, ground, &cs, fam);
    trt_sphere_families(spheres, n, ground, &cs, &P, fam + 2, NULL, &consts);
}

/* table of the family a path ray with the kernel's family code `code` is looked up in (trt_raygrid.h: 0 eye, 1 mirror eye,
 * 2 + i sphere i -- the patch follows from the origin --, 2 + n + (i << TRT_PATCH_SHIFT | k) mirror image of patch k of sphere i) */
static int table_of(const tables *T, const double *spheres, int code, const double *o, int *patch_out)
{
    const int n = T->n, P = T->patches.count;
    *patch_out = 0;
    if (code < 2)
        return code;
    if (code < 2 + n)
    {
        const int i = code - 2;
        const int k = trt_patch_of(T->patches.m, o[0] - spheres[9 * i], o[1] - spheres[9 * i + 1], o[2] - spheres[9 * i + 2]);
        *patch_out = k;
        return 2 + i * P + k;
    }
    const int s = code - 2 - n, i = s >> TRT_PATCH_SHIFT, k = s & ((1 << TRT_PATCH_SHIFT) - 1);
    *patch_out = k;
    return 2 + (n + i) * P + k;
}

static tables *build(const double *spheres, int n, const double *ground, const double *eye, int g_eye, int g_sph, int patch_m, ray_stats *st)
{
    tables *T = (tables *)calloc(1, sizeof *T);
    trt_patchset_init(&T->patches, patch_m);
    T->n = n, T->g_eye = g_eye, T->g_sph = g_sph, T->families = 2 + 2 * n * T->patches.count;
    T->fam = (trt_rayfamily *)malloc(sizeof(trt_rayfamily) * (size_t)T->families);
    raygrid_families(spheres, n, ground, eye, patch_m, T->fam);
    const int words = (n + 63) / 64 > 0 ? (n + 63) / 64 : 1;
    const size_t total = family_base(T, T->families);
    T->cells = (unsigned long long *)malloc(sizeof(unsigned long long) * total);
    T->bits = n > TRT_LIST_MAX_SPHERES ? 16 : 8;
    T->pool_cap = total * (size_t)(T->bits / 8);
    T->pool = (unsigned long long *)malloc(sizeof(unsigned long long) * T->pool_cap);
    unsigned long long bits = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : bits)
    for (int f = 0; f < T->families; f++)
    {
        const int g = f < 2 ? g_eye : g_sph;
        const size_t cells = 6 * (size_t)g * g;
        unsigned long long *masks = (unsigned long long *)malloc(sizeof(unsigned long long) * cells * words);
        trt_pointgrid_cone *cones = (trt_pointgrid_cone *)malloc(sizeof(trt_pointgrid_cone) * (size_t)(n ? n : 1));
        bits += (unsigned long long)trt_rayfamily_build(spheres, n, &T->fam[f], g, masks, cones);
        unsigned long long *out = T->cells + family_base(T, f);
        for (size_t c = 0; c < cells; c++)
        {
            const int count = trt_list_count(masks + c * words, words);
            size_t at = 0;
            int room = 1;
            if (trt_list_pool_words(count, T->bits))
            {
#pragma omp critical
                {
                    at = T->pool_words;
                    room = at + (size_t)trt_list_pool_words(count, T->bits) <= T->pool_cap;
                    if (room)
                        T->pool_words += (size_t)trt_list_pool_words(count, T->bits);
                }
            }
            out[c] = trt_list_pack(masks + c * words, words, count, room ? T->pool : NULL, (unsigned)at, T->bits);
        }
        free(cones);
        free(masks);
    }
    st->bits_set = bits;
    st->cells = total;
    st->pool_words = T->pool_words;
    for (size_t c = 0; c < total; c++)
    {
        const unsigned ctl = (unsigned)(T->cells[c] >> 56);
        st->pooled_cells += ctl == TRT_LIST_POOLED;
        st->none_cells += ctl == TRT_LIST_NONE;
    }
    return T;
}

static void destroy(tables *T)
{
    free(T->fam);
    free(T->cells);
    free(T->pool);
    free(T);
}

static int in_list(const tables *T, unsigned long long cell, int sphere)
{
    const int e = trt_list_entries(cell);
    int prev = -1;
    for (int k = 0; k < e; k++)
    {
        const int i = trt_list_entry(cell, T->pool, k, T->bits);
        if (i <= prev)
            return -2; /* not ascending: a malformed list */
        prev = i;
        if (i == sphere)
            return 1;
    }
    return 0;
}

/* candidates of ray (o, d) in family f; -1: not a member (or no list) */
static int lookup(const tables *T, int f, const double *o, const double *d, unsigned long long *cell_out)
{
    const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (!(fabs(a - 1.0) <= 9.094947017729282e-13) || !trt_rayfamily_member(&T->fam[f], o[0], o[1], o[2], d[0], d[1], d[2]))
        return -1;
    const int g = f < 2 ? T->g_eye : T->g_sph;
    const int c = trt_cubemap_cell((float)d[0], (float)d[1], (float)d[2], 0.5f * (float)g, (float)(g - 1), g);
    if (c < 0 || c >= 6 * g * g)
        return -3;
    *cell_out = T->cells[family_base(T, f) + (size_t)c];
    return trt_list_entries(*cell_out);
}

static void note(ray_stats *st, const double *ray, int sphere, int family)
{
    if (!st->violations)
    {
        memcpy(st->first_violation, ray, 6 * sizeof(double));
        st->first_violation[6] = sphere;
        st->first_violation[7] = family;
    }
    st->violations++;
}

/* rays: n_rays x 6 doubles in trace order, kinds[r] == 0 marks path rays (other rays are skipped) */
void raygrid_check(const double *spheres, int n, const double *ground, const double *eye, const double *rays, const unsigned char *kinds,
                   size_t n_rays, int g_eye, int g_sph, int patch_m, int brute, ray_stats *st)
{
    memset(st, 0, sizeof *st);
    tables *T = build(spheres, n, ground, eye, g_eye, g_sph, patch_m, st);
    int src = -1;
    unsigned group_max = 0;
    size_t seen = 0;
    for (size_t r = 0; r < n_rays; r++)
    {
        if (kinds[r] != 0)
            continue;
        const double *o = rays + 6 * r, *d = o + 3;
        const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (o[0] == eye[0] && o[1] == eye[1] && o[2] == eye[2])
            src = 0; /* a new sample */
        st->rays++;
        unsigned long long cell = 0;
        int patch = 0;
        const int table = src >= 0 ? table_of(T, spheres, src, o, &patch) : -1;
        const int cand = src >= 0 ? lookup(T, table, o, d, &cell) : -1;
        if (cand < 0)
        {
            st->non_members++;
            if (cand == -3)
                note(st, o, -3, src);
        }
        else
        {
            st->members++;
            st->by_family[src < 2 ? src : (src < 2 + n ? 2 : 3)]++;
            st->candidates += (unsigned)cand;
            st->cand_hist[cand > 16 ? 16 : cand]++;
            group_max = (unsigned)cand > group_max ? (unsigned)cand : group_max;
            for (int i = 0; i < n; i++)
            {
                double t;
                const int hit = exact_hit(o, d, a, spheres + 9 * i, &t);
                st->exact_hits += hit;
                const int in = in_list(T, cell, i);
                if (in < 0 || (hit && !in))
                    note(st, o, i, src);
            }
        }
        if ((++seen & 63) == 0)
        {
            st->wave_max_cand += group_max;
            st->wave_groups++;
            group_max = 0;
        }
        if (brute)
            for (int f = 0; f < T->families; f++)
            {
                unsigned long long c2 = 0;
                if (lookup(T, f, o, d, &c2) < 0)
                    continue;
                st->brute_pairs++;
                for (int i = 0; i < n; i++)
                {
                    double t;
                    if (exact_hit(o, d, a, spheres + 9 * i, &t) && in_list(T, c2, i) != 1)
                    {
                        st->brute_violations++;
                        note(st, o, i, f);
                    }
                }
            }
        /* the family of the NEXT path ray of this sample */
        const int what = closest(spheres, n, ground, o, d);
        if (what < 0)
            src = -1;
        else if (what < n)
            src = 2 + what;
        else
            src = src == 0 ? 1 : (src >= 2 && src < 2 + n ? 2 + n + (((src - 2) << TRT_PATCH_SHIFT) | patch) : -1);
    }
    destroy(T);
}

/* the host reference tables as list cells + pool, for the GPU test that compares the device-built tables: returns the pool
 * words used; cells must hold 2*6*g_eye^2 + 2nP*6*g_sph^2 words (P patches per sphere).  Cells are compared through their entries (the pool
 * offsets depend on the order in which cells reserve their words). */
long raygrid_host_cells(const double *spheres, int n, const double *ground, const double *eye, int g_eye, int g_sph, int patch_m,
                        unsigned long long *cells, unsigned long long *pool, long pool_cap)
{
    ray_stats st;
    memset(&st, 0, sizeof st);
    tables *T = build(spheres, n, ground, eye, g_eye, g_sph, patch_m, &st);
    const size_t total = family_base(T, T->families);
    memcpy(cells, T->cells, total * sizeof(unsigned long long));
    const long used = (long)T->pool_words;
    if (used <= pool_cap)
        memcpy(pool, T->pool, (size_t)used * sizeof(unsigned long long));
    destroy(T);
    return used;
}

/* The patches of trt_patchset_init: `samples` pseudo-random points of the unit sphere (and points pushed onto the patches' edges,
 * the cube map's edges and corners), each looked up exactly as the kernel does (trt_patch_of, FP32); returns the largest
 * |u - t_k| / rho_k met (must be <= 1: the patch's ball holds every point that is looked up in it) and, in *worst_rho, the largest
 * rho_k.  Also checks that every patch index in [0, count) occurs. */
double raygrid_patch_cover(int m, long samples, double *worst_rho, int *patches_seen)
{
    trt_patchset P;
    trt_patchset_init(&P, m);
    unsigned char seen[TRT_PATCH_MAX];
    memset(seen, 0, sizeof seen);
    double worst = 0.0;
    *worst_rho = 0.0;
    for (int k = 0; k < P.count; k++)
        *worst_rho = P.rec[k][3] > *worst_rho ? P.rec[k][3] : *worst_rho;
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    for (long n = 0; n < samples; n++)
    {
        double u[3];
        for (int a = 0; a < 3; a++)
        {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            u[a] = (double)(long long)(s >> 11) * 0x1p-52 - 1.0; /* [-1, 1) */
        }
        if (n % 5 == 1 && m) /* onto an edge between two patches of a face */
            u[(n / 5) % 3] = u[((n / 5) + 1) % 3] * (double)((n / 15) % (2 * m + 1) - m) / (double)m;
        if (n % 5 == 2) /* onto a cube edge: two coordinates of equal magnitude */
            u[(n / 5) % 3] = (n & 64 ? -1.0 : 1.0) * u[((n / 5) + 1) % 3];
        if (n % 5 == 3) /* near a cube corner */
            u[0] = (n & 64 ? -1.0 : 1.0) * u[2] * (1.0 + 1e-9 * (double)(n % 7)), u[1] = (n & 128 ? -1.0 : 1.0) * u[2];
        const double len = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (!(len > 1e-3))
            continue;
        for (int a = 0; a < 3; a++)
            u[a] /= len;
        /* the kernel looks the patch up from o - c = |r| u with any |r|: the look-up only sees ratios */
        const double radius = 0.1 + (double)(n % 97);
        const int k = trt_patch_of(m, radius * u[0], radius * u[1], radius * u[2]);
        if (k < 0 || k >= P.count)
            return 1e9;
        seen[k] = 1;
        const double e[3] = {u[0] - P.rec[k][0], u[1] - P.rec[k][1], u[2] - P.rec[k][2]};
        const double ratio = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) / P.rec[k][3];
        worst = ratio > worst ? ratio : worst;
    }
    *patches_seen = 0;
    for (int k = 0; k < P.count; k++)
        *patches_seen += seen[k];
    return worst;
}
