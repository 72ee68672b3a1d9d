/*
 * trt_raygrid.h -- candidate tables for PATH rays, and the list form of every candidate table (host + device, plain C).
 *
 * trace_ray (TRT.c:793-856) tests every sphere against every ray.  The production kernel used to find the few spheres
 * a path ray can touch with a wave-uniform FP32 sweep over all N spheres (trt_filter.h: 9 VALU per sphere and ray).
 * Path rays, however, come in FAMILIES whose members all pass (almost) through one point:
 *
 *   family 0          primary rays: they start at the eye (TRT.c:1010)
 *   family 1          rays reflected by the ground plane whose parent ray started at the eye: mirror reflection about a
 *                     plane maps the parent's line onto a line through the MIRROR IMAGE of the eye (TRT.c:1054-1056)
 *   family 2 + i      rays that start on sphere i (reflected there): their line passes within r_i of its centre c_i
 *   family 2 + N + i  rays reflected by the ground whose parent started on sphere i: within r_i of the mirror image of c_i
 *
 * A path ray's parent is the eye, a sphere or the ground, and a ray never goes from the ground to the ground, so these
 * 2 + 2N families cover every path ray.  For a family F = (apex A, radius R): every ray (o, d), |d| = 1, whose LINE passes
 * within R of A and whose origin is not more than R "behind" A, lambda = (o - A).d >= -R.  Claim: if such a ray hits a
 * sphere (centre c, radius rho) at some t > 0, then with D = |c - A| and s = (rho + R)/D either s >= 0.7 or the angle
 * between d and c - A is at most asin(s).  [With a = the line's point closest to A: |a - A| <= R, a - A is perpendicular
 * to d, the hit point is a + mu d with mu = lambda + t > -R.  (c - A).d = (c - a).d >= mu - rho > -(R + rho), i.e.
 * cos(angle) > -s; the distance from c to the parallel line through A is <= rho + R, i.e. sin(angle) <= s.  For s < 0.7
 * the branch angle >= pi - asin(s) has cos <= -sqrt(1 - s^2) < -s: excluded.]  "Which spheres can the ray touch" thus
 * depends on d alone, and is tabulated per family over a cube map of 6 x g x g direction cells exactly like the point
 * lights' tables of trt_lightgrid.h (same cone records, same conservative cone / cell predicate, same FP32 look-up and
 * growth constants); spheres with s >= 0.7 -- sphere i itself in its own families -- sit in every cell.
 *
 * Whether a ray BELONGS to the family it is looked up in is not taken on trust: trt_rayfamily_member() checks the two
 * conditions above in FP64 for every ray (a cross product and two dot products), so the tables are conservative for any
 * ray that passes, wherever it came from; a ray that fails (hit points far from the origin lose digits, degenerate
 * normals, rays beyond the tables' range) falls back to the sweep.  Like the FP32 filter and the light tables, a table
 * NEVER decides a hit: it only proposes the spheres that go to the EXACT FP64 test, in ascending index order.
 * tests/test_raygrid.py checks "exact hit => in the list" on every path ray of real frames with this very code.
 *
 * Bounds.  rho^2 = r^2 + 2^-37 M^2 + ..., M >= |o - c| + r over the admissible origins (|o - A| <= rg), covers the
 * reference's own discriminant rounding as in trt_lightgrid.h (1); the cones are built with R_build = r_chk (1 + 1e-9) +
 * 1e-13 rg, which covers the rounding of the membership test itself (<= 8u |o - A| on the cross product) and
 * |d.d - 1| <= 2^-40; the FP32 look-up of the cell from d is the point lights' (cones grown by 1e-5 rad, cells by 0.01
 * cell, edge cells stretched past the face).
 *
 * LIST CELLS.  Every candidate table the kernel reads -- these and the light tables -- is stored as one 64-bit word per
 * cell: byte 7 = count 0..7 and bytes 0..6 the sphere indices in ascending order; or byte 7 = 0x80, bits 32..47 = count,
 * bits 0..31 = the offset of ceil(count/8) words of indices in a pool; or byte 7 = 0xFF: no list (pool exhausted), the
 * ray falls back to the sweep.  One 8-byte load whatever N is, and the exact stage pops bytes.  Scenes of more than 256
 * spheres use 16-bit entries (3 inline, 4 per pool word) in every table; beyond 1024 spheres their path rays sweep.
 */
#ifndef TRT_RAYGRID_H
#define TRT_RAYGRID_H

#include "trt_lightgrid.h"

typedef struct
{
    double a[3];   /* apex */
    double r_chk;  /* membership: line within r_chk of the apex, origin not more than r_chk behind it */
    double r_chk2; /* r_chk^2 */
    double rg2;    /* admissible |o - apex|^2 */
} trt_rayfamily;

#define TRT_RAYFAMILY_DOUBLES 6

/* list cells index spheres with 8 bits (scenes of up to 256 spheres: every table) or 16 bits (larger scenes: the light tables up to
 * TRT_LIST_MAX_SPHERES_WIDE spheres, the path rays' family tables up to TRT_PATH_MAX_SPHERES -- a workgroup of the device builder
 * holds a family's cones, 64 bytes a sphere, and its tile's mask words in LDS; beyond that the path rays sweep) */
#define TRT_LIST_MAX_SPHERES 256
#define TRT_PATH_MAX_SPHERES 1024
#define TRT_LIST_MAX_SPHERES_WIDE 65535
#define TRT_LIST_POOLED 0x80u
#define TRT_LIST_NONE 0xFFu

/* P mirrored about the plane through p0 with normal nrm (any length > 0): P - 2 ((P - p0).nrm / nrm.nrm) nrm */
TRT_HD void trt_mirror_point(const double p[3], const double p0[3], const double nrm[3], double out[3])
{
    const double nn = nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2];
    const double h = ((p[0] - p0[0]) * nrm[0] + (p[1] - p0[1]) * nrm[1] + (p[2] - p0[2]) * nrm[2]) / nn;
    out[0] = p[0] - 2.0 * h * nrm[0];
    out[1] = p[1] - 2.0 * h * nrm[1];
    out[2] = p[2] - 2.0 * h * nrm[2];
}

/* Slacks of the membership radius: the 1e-6 nudge of TRT.c:871-874 moves a ray's origin off the surface it starts on (and
 * off the ideal line through a mirror apex), rounded hit points add a little; mirror families inherit their parent's. */
#define TRT_FAMILY_SLACK 4e-6

TRT_HD void trt_rayfamily_set(trt_rayfamily *F, const double apex[3], double r_chk, double rg)
{
    F->a[0] = apex[0], F->a[1] = apex[1], F->a[2] = apex[2];
    F->r_chk = r_chk;
    F->r_chk2 = r_chk * r_chk;
    F->rg2 = rg * rg;
}

/* the two families of the eye: 0 = rays from the eye, 1 = their reflections by the ground (apex = the eye's mirror image) */
static inline void trt_eye_families(const double eye[3], const double *ground /* Plane: point, normal */, const trt_cull_scene *cs,
                                    trt_rayfamily fam[2])
{
    const double reach = (double)cs->cn + (double)cs->rm;
    double m[3];
    trt_mirror_point(eye, ground, ground + 3, m);
    const double *apex[2] = {eye, m};
    for (int k = 0; k < 2; k++)
    {
        const double lc[3] = {apex[k][0] - cs->c0[0], apex[k][1] - cs->c0[1], apex[k][2] - cs->c0[2]};
        const double away = __builtin_sqrt(lc[0] * lc[0] + lc[1] * lc[1] + lc[2] * lc[2]);
        const double mag = __builtin_fabs(apex[k][0]) + __builtin_fabs(apex[k][1]) + __builtin_fabs(apex[k][2]) + reach;
        trt_rayfamily_set(&fam[k], apex[k], (k ? TRT_FAMILY_SLACK : 0.0) + 1e-9 * mag, TRT_LIGHTGRID_RANGE * (reach + away) + 1.0);
    }
}

/* cell of direction (x, y, z) (any length > 0) in a cube map of 6 x g x g cells: the look-up of trt_pointgrid_cell (trt_cube_lookup) */
TRT_HD int trt_cubemap_cell(float x, float y, float z, float half_g, float g_max, int g)
{
    int face;
    float pu, pv, ma2;
    trt_cube_lookup(x, y, z, &face, &pu, &pv, &ma2);
#if defined(__HIP_DEVICE_COMPILE__)
    const float inv = __builtin_amdgcn_rcpf(0.5f * __builtin_fabsf(ma2));
#else
    const float inv = 1.0f / (0.5f * __builtin_fabsf(ma2));
#endif
    float cu = __builtin_fmaf(pu * inv, half_g, half_g), cv = __builtin_fmaf(pv * inv, half_g, half_g);
    cu = __builtin_fminf(__builtin_fmaxf(cu, 0.0f), g_max);
    cv = __builtin_fminf(__builtin_fmaxf(cv, 0.0f), g_max);
    return (face * g + (int)cv) * g + (int)cu;
}

/* ---- SUB-FAMILIES of a sphere ("patches") --------------------------------------------------------------------------
 * "Every ray that starts anywhere on sphere i" is a fat family: its members pass within r_i of the centre, so a direction
 * cell holds every sphere within r_i + r_j of the line through the centre -- 6.5 candidates per path ray for 1.1 exact
 * hits in a dense scene (256 spheres).  The ORIGIN of such a ray is known, though: the surface of sphere i is cut into
 * P = 6 m^2 patches by a cube map of the direction centre -> origin (m cells per face side), and every patch k gets its own
 * pair of families: apex  c_i + |r_i| t_k  (t_k inside the unit ball, under the middle of the patch), membership radius
 * |r_i| rho_k + slack with rho_k = the largest distance from t_k to a point of the patch on the unit sphere (0.82 for m = 1,
 * 0.52 for m = 2, 0.44 for m = 3, 0.33 for m = 4 instead of 1).  A ray that starts on patch k has its origin, hence its line,
 * within that radius of the apex: it is a member -- and trt_rayfamily_member() still checks that for every ray, so a ray
 * looked up in the wrong patch (the FP32 choice of the patch near a patch's edge, a degenerate normal) falls back to the sweep
 * like any other non-member; nothing is taken on trust.  The mirror family of a patch (reflections by the ground of rays that
 * started on it) has the mirror image of that apex.  m = 0: one family per sphere (t = 0, rho = 1), the tables of round 2.
 * A path ray's family code with patches: 0 eye, 1 mirror eye, 2 + i: it starts on sphere i (the patch follows from its origin),
 * 2 + n + (i << TRT_PATCH_SHIFT | k): reflected by the ground, parent started on patch k of sphere i; < 0: none.  Without
 * (m = 0): 0, 1, 2 + i, 2 + n + i.
 * Tables: 2 of the eye, then n P of the patches (sphere-major), then n P of their mirror images. */
#define TRT_PATCH_MAX_M 4
#define TRT_PATCH_MAX (6 * TRT_PATCH_MAX_M * TRT_PATCH_MAX_M)
#define TRT_PATCH_SHIFT 7
#define TRT_PATCH_RECORD 8 /* doubles per patch in the kernel's copy: t (3), rho | mirrored t (3), rho */

typedef struct
{
    int m;     /* cells per face side of the origin's cube map; 0: one family per sphere */
    int count; /* P: 1, or 6 m^2 */
    double rec[TRT_PATCH_MAX][4]; /* per patch: apex offset t in units of the radius (3), membership radius rho in units of the radius */
} trt_patchset;

static inline void trt_patchset_init(trt_patchset *P, int m)
{
    if (m < 0)
        m = 0;
    if (m > TRT_PATCH_MAX_M)
        m = TRT_PATCH_MAX_M;
    P->m = m;
    P->count = m ? 6 * m * m : 1;
    for (int k = 0; k < TRT_PATCH_MAX; k++)
        P->rec[k][0] = P->rec[k][1] = P->rec[k][2] = 0.0, P->rec[k][3] = 1.0 + 1e-6;
    if (!m)
        return; /* the whole sphere: apex = the centre, radius r (1 + 1e-6) */
    const double step = 2.0 / (double)m;
    for (int face = 0; face < 6; face++)
        for (int j = 0; j < m; j++)
            for (int c = 0; c < m; c++)
            {
                /* the patch grown by 0.02 cell, and past the face's edge where the look-up clamps: its FP32 choice of the
                 * cell is off by <= 1e-6, and near a cube edge either face may be chosen */
                const double u0 = c == 0 ? -1.001 : -1.0 + ((double)c - 0.02) * step, u1 = c == m - 1 ? 1.001 : -1.0 + ((double)c + 1.02) * step;
                const double v0 = j == 0 ? -1.001 : -1.0 + ((double)j - 0.02) * step, v1 = j == m - 1 ? 1.001 : -1.0 + ((double)j + 1.02) * step;
                const double cu[4] = {u0, u1, u1, u0}, cv[4] = {v0, v0, v1, v1};
                double corner[4][3], mid[3] = {0.0, 0.0, 0.0};
                for (int q = 0; q < 4; q++)
                {
                    trt_face_direction(face, cu[q], cv[q], corner[q]);
                    const double len = __builtin_sqrt(corner[q][0] * corner[q][0] + corner[q][1] * corner[q][1] + corner[q][2] * corner[q][2]);
                    for (int a = 0; a < 3; a++)
                        corner[q][a] /= len, mid[a] += corner[q][a];
                }
                /* apex offset: centre of the smallest ball round the four corners (Badoiu-Clarkson: step towards the farthest
                 * corner, 1/(i+1) of the way).  A ball round a point t inside the unit ball cuts a cap out of the unit sphere, and
                 * a cap (smaller than a hemisphere) that holds the corners of a convex spherical quadrilateral holds all of it.
                 * Whatever the iteration arrives at is valid: rho below is measured from it. */
                double *rec = P->rec[(face * m + j) * m + c];
                for (int a = 0; a < 3; a++)
                    rec[a] = 0.25 * mid[a];
                for (int it = 1; it <= 4096; it++)
                {
                    int far_q = 0;
                    double far_d2 = -1.0;
                    for (int q = 0; q < 4; q++)
                    {
                        const double e[3] = {corner[q][0] - rec[0], corner[q][1] - rec[1], corner[q][2] - rec[2]};
                        const double d2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
                        if (d2 > far_d2)
                            far_d2 = d2, far_q = q;
                    }
                    for (int a = 0; a < 3; a++)
                        rec[a] += (corner[far_q][a] - rec[a]) / (double)(it + 1);
                }
                double far2 = 0.0;
                for (int q = 0; q < 4; q++)
                {
                    const double e[3] = {corner[q][0] - rec[0], corner[q][1] - rec[1], corner[q][2] - rec[2]};
                    const double d2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
                    far2 = d2 > far2 ? d2 : far2;
                }
                rec[3] = __builtin_sqrt(far2) * (1.0 + 1e-6) + 1e-9;
            }
}

/* patch of the origin o of a ray that starts on the sphere with centre c: the look-up of trt_cubemap_cell on o - c */
TRT_HD int trt_patch_of(int m, double wx, double wy, double wz)
{
    return m ? trt_cubemap_cell((float)wx, (float)wy, (float)wz, 0.5f * (float)m, (float)(m - 1), m) : 0;
}

/* what every family of the spheres shares: the admissible range and the slack of the membership radius */
typedef struct
{
    double rg;    /* admissible |o - apex| */
    double slack; /* membership radius of a patch's family = |r| rho + slack; TRT_FAMILY_SLACK more for its mirror image */
} trt_family_consts;

/* apex and membership radius of patch k of a sphere (centre c or, mirrored, its mirror image; t the patch's offset or its
 * mirror image): the SAME two expressions in the kernel's look-up (path_cell) and in the builders */
#define TRT_PATCH_APEX(c, r_abs, t) ((c) + (r_abs) * (t))
#define TRT_PATCH_RCHK(r_abs, rho, slack) ((r_abs) * (rho) + (slack))

/* mirror image of a DIRECTION (or offset) in the plane with normal nrm: t - 2 (t.nrm / nrm.nrm) nrm */
static inline void trt_mirror_offset(const double t[3], const double nrm[3], double out[3])
{
    const double zero[3] = {0.0, 0.0, 0.0};
    trt_mirror_point(t, zero, nrm, out);
}

/* The kernel's copy of the patches: per patch {t (3), rho, mirrored t (3), rho}. */
static inline void trt_patch_records(const trt_patchset *P, const double *ground, double *out /* [P->count * TRT_PATCH_RECORD] */)
{
    for (int k = 0; k < P->count; k++)
    {
        double tm[3];
        trt_mirror_offset(P->rec[k], ground + 3, tm);
        double *o = out + (size_t)k * TRT_PATCH_RECORD;
        o[0] = P->rec[k][0], o[1] = P->rec[k][1], o[2] = P->rec[k][2], o[3] = P->rec[k][3];
        o[4] = tm[0], o[5] = tm[1], o[6] = tm[2], o[7] = P->rec[k][3];
    }
}

/* The 2 n P families of the spheres: fam[i P + k] = rays starting on patch k of sphere i, fam[n P + i P + k] = their
 * reflections by the ground (apex = the mirror image of the centre + |r| x the mirrored offset, membership radius
 * TRT_FAMILY_SLACK larger).  All share one admissible range.  `centres` receives what the kernel keeps per sphere beside the
 * sphere itself: {mirror image of the centre (3), w} with w = the membership radius of the sphere's family when there is one
 * family per sphere (P->m == 0; the kernel adds TRT_FAMILY_SLACK for the mirror family) and |r| when there are patches. */
static inline void trt_sphere_families(const double *spheres, int n, const double *ground, const trt_cull_scene *cs, const trt_patchset *P,
                                       trt_rayfamily *fam, double *centres, trt_family_consts *consts)
{
    const double reach = (double)cs->cn + (double)cs->rm;
    const double mag = __builtin_fabs(cs->c0[0]) + __builtin_fabs(cs->c0[1]) + __builtin_fabs(cs->c0[2]) + reach + 1.0;
    double away = reach;
    for (int i = 0; i < n; i++)
    {
        double m[3];
        trt_mirror_point(spheres + 9 * i, ground, ground + 3, m);
        const double lc[3] = {m[0] - cs->c0[0], m[1] - cs->c0[1], m[2] - cs->c0[2]};
        const double far = __builtin_sqrt(lc[0] * lc[0] + lc[1] * lc[1] + lc[2] * lc[2]);
        away = far > away ? far : away; /* NaN (a ground without a normal) leaves it alone; such tables never pass the membership test */
    }
    const double rg = TRT_LIGHTGRID_RANGE * (reach + away) + 1.0;
    consts->rg = rg;
    consts->slack = TRT_FAMILY_SLACK + 1e-9 * mag;
    const int count = P->count;
    for (int i = 0; i < n; i++)
    {
        const double *c = spheres + 9 * i;
        const double r_abs = __builtin_fabs(c[3]);
        double m[3];
        trt_mirror_point(c, ground, ground + 3, m);
        if (centres)
            centres[4 * i + 0] = m[0], centres[4 * i + 1] = m[1], centres[4 * i + 2] = m[2],
                            centres[4 * i + 3] = P->m ? r_abs : TRT_PATCH_RCHK(r_abs, P->rec[0][3], consts->slack);
        for (int k = 0; k < count; k++)
        {
            const double *t = P->rec[k];
            double tm[3];
            trt_mirror_offset(t, ground + 3, tm);
            const double a[3] = {TRT_PATCH_APEX(c[0], r_abs, t[0]), TRT_PATCH_APEX(c[1], r_abs, t[1]), TRT_PATCH_APEX(c[2], r_abs, t[2])};
            const double am[3] = {TRT_PATCH_APEX(m[0], r_abs, tm[0]), TRT_PATCH_APEX(m[1], r_abs, tm[1]), TRT_PATCH_APEX(m[2], r_abs, tm[2])};
            const double r_chk = TRT_PATCH_RCHK(r_abs, t[3], consts->slack);
            trt_rayfamily_set(&fam[(size_t)i * count + k], a, r_chk, rg);
            trt_rayfamily_set(&fam[((size_t)n + i) * count + k], am, r_chk + TRT_FAMILY_SLACK, rg);
        }
    }
}

/* does the ray (o, d), d a unit vector up to 2^-40, belong to the family?  false for NaN */
TRT_HD int trt_rayfamily_member(const trt_rayfamily *F, double ox, double oy, double oz, double dx, double dy, double dz)
{
    const double wx = ox - F->a[0], wy = oy - F->a[1], wz = oz - F->a[2];
    const double cx = wy * dz - wz * dy, cy = wz * dx - wx * dz, cz = wx * dy - wy * dx;
    const double c2 = cx * cx + cy * cy + cz * cz, lam = wx * dx + wy * dy + wz * dz, w2 = wx * wx + wy * wy + wz * wz;
    return c2 <= F->r_chk2 && lam >= -F->r_chk && w2 <= F->rg2;
}

/* The cone of directions, seen from the family's apex, in which a member ray can hit sphere `s` (9-double record). */
TRT_HD void trt_rayfamily_cone(const trt_rayfamily *F, const double *s, trt_pointgrid_cone *c)
{
    const double rg = __builtin_sqrt(F->rg2);
    const double a[3] = {s[0] - F->a[0], s[1] - F->a[1], s[2] - F->a[2]};
    const double D = __builtin_sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const double M = rg + D + __builtin_fabs(s[3]);
    const double E = 0x1p-37 * M * M;
    const double r_build = F->r_chk * (1.0 + 1e-9) + 1e-13 * rg;
    const double rho = __builtin_sqrt(s[3] * s[3] + E) + 0x1p-45 * M + r_build;
    const double sin_grow = 1.0000000000e-5, cos_grow = 0.99999999995; /* sin and cos of the 1e-5 rad the cones grow by */
    c->everywhere = !(rho / D < 0.7) ? 1.0 : 0.0; /* also for D = 0 and NaN */
    if (c->everywhere != 0.0)
    {
        c->a[0] = c->a[1] = c->a[2] = 0.0;
        c->sin_a = 1.0, c->cos_a = 0.0;
        return;
    }
    for (int k = 0; k < 3; k++)
        c->a[k] = a[k] / D;
    const double sn = rho / D, cn = __builtin_sqrt(1.0 - sn * sn);
    const double sg = sn * cos_grow + cn * sin_grow + 1e-12, cg = cn * cos_grow - sn * sin_grow - 1e-12; /* half-angle + 1e-5 rad */
    c->sin_a = sg < 1.0 ? sg : 1.0;
    c->cos_a = cg > 0.0 ? cg : 0.0;
    if (!(cg > 0.0))
        c->everywhere = 1.0;
}

/* ---- list cells ---- */

/* Pack the spheres marked in a cell's mask words (sphere j of word w at bit 63 - (j & 63)) into a list cell.  count <= 7:
 * inline.  Otherwise ceil(count/8) pool words starting at *pool_next (the caller reserved them): returns the pooled
 * cell and writes the indices; pool == NULL (no room): TRT_LIST_NONE. */
TRT_HD int trt_list_count(const unsigned long long *mask, int words)
{
    int count = 0;
    for (int w = 0; w < words; w++)
        count += __builtin_popcountll(mask[w]);
    return count;
}

/* `bits`: 8 (scenes of up to 256 spheres) or 16 bits per entry: 7 or 3 entries inline, 8 or 4 per pool word */
TRT_HD unsigned long long trt_list_pack(const unsigned long long *mask, int words, int count, unsigned long long *pool, unsigned pool_offset,
                                        int bits)
{
    const int per = 64 / bits, inline_max = 56 / bits;
    if (count <= inline_max)
    {
        unsigned long long cell = (unsigned long long)count << 56;
        int k = 0;
        for (int w = 0; w < words; w++)
        {
            unsigned long long m = mask[w];
            while (m)
            {
                const int lead = __builtin_clzll(m);
                m &= ~(0x8000000000000000ull >> lead);
                cell |= (unsigned long long)(unsigned)(w * 64 + lead) << (bits * k++);
            }
        }
        return cell;
    }
    if (!pool)
        return (unsigned long long)TRT_LIST_NONE << 56;
    unsigned long long cur = 0;
    int k = 0;
    for (int w = 0; w < words; w++)
    {
        unsigned long long m = mask[w];
        while (m)
        {
            const int lead = __builtin_clzll(m);
            m &= ~(0x8000000000000000ull >> lead);
            cur |= (unsigned long long)(unsigned)(w * 64 + lead) << (bits * (k % per));
            if (++k % per == 0)
            {
                pool[pool_offset + (unsigned)(k / per) - 1u] = cur;
                cur = 0;
            }
        }
    }
    if (k % per)
        pool[pool_offset + (unsigned)(k / per)] = cur;
    return ((unsigned long long)TRT_LIST_POOLED << 56) | ((unsigned long long)(unsigned)count << 32) | pool_offset;
}

/* pool words a list of `count` entries takes (0: it is inline) */
TRT_HD unsigned trt_list_pool_words(int count, int bits)
{
    const int per = 64 / bits, inline_max = 56 / bits;
    return count <= inline_max ? 0u : (unsigned)((count + per - 1) / per);
}

/* number of entries of a list cell, -1 for TRT_LIST_NONE */
TRT_HD int trt_list_entries(unsigned long long cell)
{
    const unsigned ctl = (unsigned)(cell >> 56);
    if (ctl == TRT_LIST_NONE)
        return -1;
    return ctl & TRT_LIST_POOLED ? (int)((cell >> 32) & 0xffffu) : (int)ctl;
}

/* entry k of a list cell */
TRT_HD int trt_list_entry(unsigned long long cell, const unsigned long long *pool, int k, int bits)
{
    const int per = 64 / bits;
    const unsigned long long emask = (1ull << bits) - 1ull;
    if ((unsigned)(cell >> 56) & TRT_LIST_POOLED)
        return (int)((pool[(unsigned)cell + (unsigned)(k / per)] >> (bits * (k % per))) & emask);
    return (int)((cell >> (bits * k)) & emask);
}

/* ---- marking the cells ----
 * bit (cell, sphere) of a family's table = the cone / cell predicate of trt_lightgrid.h for the cell AND, when the table's side
 * is a multiple of TRT_FAMILY_TILE, the same predicate for the TILE the cell lies in -- cell (c / TILE, j / TILE) of the grid of
 * side g / TILE on the same face.  Both are conservative for their cell (trt_lightgrid.h (2)): a member ray whose direction
 * the look-up assigns to cell (c, j) lies in that cell grown by 0.01 of its side, which lies inside the tile grown by 0.01 of
 * ITS side, so a sphere such a ray can hit passes both.  The tile's test is what makes building a table cheap: a builder
 * evaluates it once per (tile, sphere) and then the cell's predicate only for the few spheres whose cone reaches the tile
 * (csrc/trt_tables.hip: build_family_lists_kernel, a tile per workgroup; the eye's tables, which are rebuilt whenever the
 * camera moves, took 0.95 ms at 256 spheres with every cell asking every sphere). */
#define TRT_FAMILY_TILE 8
#define TRT_LIST_MAX_SPHERES_HOST 256 /* the host builder caches a tile row's answers for up to this many spheres */
TRT_HD int trt_family_tiled(int g) { return g >= TRT_FAMILY_TILE && g % TRT_FAMILY_TILE == 0; }

/* ---- host reference builder (tests; the library marks the cells on the device with the same predicates) ---- */
static inline long trt_rayfamily_build(const double *spheres, int n, const trt_rayfamily *F, int g, unsigned long long *masks,
                                       trt_pointgrid_cone *cones)
{
    const int words = (n + 63) / 64 > 0 ? (n + 63) / 64 : 1;
    for (int i = 0; i < n; i++)
        trt_rayfamily_cone(F, spheres + 9 * i, cones + i);
    const int tiled = trt_family_tiled(g), gt = g / TRT_FAMILY_TILE;
    long bits = 0;
    long tile_of_row = -1;                 /* the tile row whose answers `reach` holds: (face, j / TILE) */
    unsigned char reach[64 * TRT_LIST_MAX_SPHERES_HOST]; /* [tile column][sphere]: does the cone reach the tile?  (gt <= 64 columns) */
    for (long cell = 0; cell < 6L * g * g; cell++)
    {
        unsigned long long *m = masks + cell * words;
        for (int w = 0; w < words; w++)
            m[w] = 0;
        const int face = (int)(cell / ((long)g * g)), j = (int)((cell / g) % g), c = (int)(cell % g);
        const int cached = tiled && gt <= 64 && n <= TRT_LIST_MAX_SPHERES_HOST;
        if (cached && tile_of_row != (long)face * gt + j / TRT_FAMILY_TILE)
        {
            tile_of_row = (long)face * gt + j / TRT_FAMILY_TILE;
            for (int tc = 0; tc < gt; tc++)
                for (int i = 0; i < n; i++)
                    reach[tc * TRT_LIST_MAX_SPHERES_HOST + i] = (unsigned char)trt_pointgrid_reaches(cones + i, face, tc, j / TRT_FAMILY_TILE, gt);
        }
        for (int i = 0; i < n; i++)
        {
            const int in_tile = !tiled ? 1 : (cached ? reach[(c / TRT_FAMILY_TILE) * TRT_LIST_MAX_SPHERES_HOST + i]
                                                    : trt_pointgrid_reaches(cones + i, face, c / TRT_FAMILY_TILE, j / TRT_FAMILY_TILE, gt));
            if (in_tile && trt_pointgrid_reaches(cones + i, face, c, j, g))
            {
                trt_lightgrid_set(m, i);
                bits++;
            }
        }
    }
    return bits;
}

#endif
