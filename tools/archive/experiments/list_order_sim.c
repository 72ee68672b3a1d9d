/* list_order_sim.c -- EXPERIMENT (round 4): what would candidate lists ordered by distance from the family's apex buy?
 * Host-side simulation on the oracle's path rays with the host reference tables (tests/raygrid_check.c is included for its
 * builder).  For every member ray: the number of exact tests today (the whole list, ascending index) against the number when
 * the list is ordered by L_j = |c_j - A| - r_j and a lane stops as soon as its best hit is nearer than anything the rest of
 * the list could offer (t_best < L_j - |o - A|), with the ground tested first.  Also the per-64-ray maxima (a wave runs as
 * long as its busiest lane).  Not product code. */
#include "../../../tests/raygrid_check.c"

typedef struct
{
    unsigned long long rays, members, tests_now, tests_sorted, tests_sorted_q, wave_now, wave_sorted, wave_sorted_q, groups;
    unsigned long long hist_now[33], hist_sorted[33];
} order_stats;

static double plane_t(const double *ground, const double *o, const double *d)
{
    const double *gp = ground, *gn = ground + 3;
    const double denom = d[0] * gn[0] + d[1] * gn[1] + d[2] * gn[2];
    if (fabs(denom) > 0.00001)
    {
        const double t = ((gp[0] - o[0]) * gn[0] + (gp[1] - o[1]) * gn[1] + (gp[2] - o[2]) * gn[2]) / denom;
        if (t > 0.00001)
            return t;
    }
    return INFINITY;
}

void order_sim(const double *spheres, int n, const double *ground, const double *eye, const double *rays, const unsigned char *kinds,
               size_t n_rays, int g_eye, int g_sph, int patch_m, double quantum, int plane_first, order_stats *os)
{
    ray_stats st;
    memset(&st, 0, sizeof st);
    memset(os, 0, sizeof *os);
    /* no tables: the list of a ray's cell is formed directly (cone of every sphere against the cell), so that fine resolutions
     * whose host build would take hours can be simulated */
    tables *T = (tables *)calloc(1, sizeof *T);
    trt_patchset_init(&T->patches, patch_m);
    T->n = n, T->g_eye = g_eye, T->g_sph = g_sph, T->families = 2 + 2 * n * T->patches.count;
    T->fam = (trt_rayfamily *)malloc(sizeof(trt_rayfamily) * (size_t)T->families);
    raygrid_families(spheres, n, ground, eye, patch_m, T->fam);
    (void)st;
    int src = -1;
    unsigned gmax_now = 0, gmax_s = 0, gmax_q = 0;
    size_t seen = 0;
    for (size_t r = 0; r < n_rays; r++)
    {
        if (kinds[r] != 0)
            continue;
        const double *o = rays + 6 * r, *d = o + 3;
        const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (o[0] == eye[0] && o[1] == eye[1] && o[2] == eye[2])
            src = 0;
        os->rays++;
        unsigned long long cell = 0;
        int patch = 0;
        const int table = src >= 0 ? table_of(T, spheres, src, o, &patch) : -1;
        int cand = -1, direct[256];
        if (src >= 0 && fabs(a - 1.0) <= 9.094947017729282e-13 && trt_rayfamily_member(&T->fam[table], o[0], o[1], o[2], d[0], d[1], d[2]))
        {
            const int g = table < 2 ? g_eye : g_sph;
            const int c = trt_cubemap_cell((float)d[0], (float)d[1], (float)d[2], 0.5f * (float)g, (float)(g - 1), g);
            const int face = c / (g * g), j = (c / g) % g, cc = c % g;
            cand = 0;
            for (int i = 0; i < n; i++)
            {
                trt_pointgrid_cone cone;
                trt_rayfamily_cone(&T->fam[table], spheres + 9 * i, &cone);
                if (trt_pointgrid_reaches(&cone, face, cc, j, g))
                    direct[cand++] = i;
            }
        }
        (void)cell;
        if (cand >= 0)
        {
            os->members++;
            const trt_rayfamily *F = &T->fam[table];
            const double w[3] = {o[0] - F->a[0], o[1] - F->a[1], o[2] - F->a[2]};
            const double wl = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            int idx[256];
            double key[256];
            for (int k = 0; k < cand; k++)
            {
                const int i = direct[k];
                const double *s = spheres + 9 * i;
                const double e[3] = {s[0] - F->a[0], s[1] - F->a[1], s[2] - F->a[2]};
                idx[k] = i;
                key[k] = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) - fabs(s[3]);
            }
            for (int x = 1; x < cand; x++) /* insertion sort by key */
                for (int y = x; y > 0 && key[y] < key[y - 1]; y--)
                {
                    const double tk = key[y];
                    key[y] = key[y - 1], key[y - 1] = tk;
                    const int ti = idx[y];
                    idx[y] = idx[y - 1], idx[y - 1] = ti;
                }
            const double tp = plane_first ? plane_t(ground, o, d) : INFINITY;
            for (int variant = 0; variant < 2; variant++)
            {
                double best = tp;
                int tests = 0;
                for (int k = 0; k < cand; k++)
                {
                    double bound = key[k];
                    if (variant)
                        bound = floor(bound / quantum) * quantum; /* what an 8-bit key would keep */
                    if (best < bound - wl - F->r_chk)
                        break;
                    tests++;
                    double t;
                    if (exact_hit(o, d, a, spheres + 9 * idx[k], &t) && t < best)
                        best = t;
                }
                if (variant)
                {
                    os->tests_sorted_q += (unsigned)tests;
                    gmax_q = (unsigned)tests > gmax_q ? (unsigned)tests : gmax_q;
                }
                else
                {
                    os->tests_sorted += (unsigned)tests;
                    os->hist_sorted[tests > 32 ? 32 : tests]++;
                    gmax_s = (unsigned)tests > gmax_s ? (unsigned)tests : gmax_s;
                }
            }
            os->tests_now += (unsigned)cand;
            os->hist_now[cand > 32 ? 32 : cand]++;
            gmax_now = (unsigned)cand > gmax_now ? (unsigned)cand : gmax_now;
        }
        if ((++seen & 63) == 0)
        {
            os->wave_now += gmax_now, os->wave_sorted += gmax_s, os->wave_sorted_q += gmax_q;
            os->groups++;
            gmax_now = gmax_s = gmax_q = 0;
        }
        const int what = closest(spheres, n, ground, o, d);
        if (what < 0)
            src = -1;
        else if (what < n)
            src = 2 + what;
        else
            src = src == 0 ? 1 : (src >= 2 && src < 2 + n ? 2 + n + (((src - 2) << TRT_PATCH_SHIFT) | patch) : -1);
    }
    free(T->fam);
    free(T);
}
