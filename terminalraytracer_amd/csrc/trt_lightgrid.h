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
 * (4) Point light, ANY-HIT search (round 4).  apply_lighting asks trace_ray for the CLOSEST blocker and calls the point lit iff
 *     there is none or  light_d2 < id2,  id2 = the squared distance from the origin to the blocker's hit point nudged 1e-6
 *     back along the ray (TRT.c:871-874, :937-942).  Claim: if ANY hit j of the shadow ray (t_j > 0, TRT.c:657-659, or the
 *     ground's t_j > 1e-5, TRT.c:685) has  d2_j <= light_d2  (d2_j as TRT.c:810-815 forms it) and  light_d2 > dark_floor,
 *     the reference's answer is DARK.  [The closest hit c has d2_c <= d2_j <= light_d2, so it is enough to show id2_c <=
 *     light_d2.  With D = |o - p_c| (d2_c = D^2 (1 +- 4u)) the nudged point is p_c + unit(o - p_c) 1e-6, or p_c + (o - p_c) 1e-6
 *     when D <= 1e-4 (normalize_vector leaves short vectors alone, TRT.c:444): its distance from o is at most
 *     max(D - 1e-6 (1 - 1e-15), D (1 - 1e-6)) + e,  e <= 2 sqrt(3) u Mg the rounding of the nudged point and of its difference
 *     from o, Mg >= every coordinate involved.  Mg <= 2^28 gives e <= 1.1e-7.  If 1e-6 D >= 2e the distance is <= D (1 - 5e-7)
 *     (D <= 1e-4) or <= D - 8.9e-7 (D > 1e-4; D <= rg << 1.7e9), so id2 < d2_c <= light_d2.  Otherwise D < 2e6 e and
 *     id2 <= (D + e)^2 (1 + 4u) < 6.2e-19 Mg^2 <= dark_floor = 2^-60 Mg^2 < light_d2.]
 *     Likewise a hit j with  id2_j > light_d2  cannot make the answer dark and may be IGNORED: if it is not the closest it plays no
 *     part, and if it is, every other hit is at least as far and the answer is "lit" -- what the search returns when nothing
 *     (else) is hit.  [id2_j >= ((D - delta)^2)(1 - 4u), delta = 1e-6 (1 + 1e-15) + e <= 1.2e-6, and (D - delta)^2 >= D^2 (1 -
 *     delta) - delta.]
 *     The kernel does not form d2_j: with q = -b - sqrt(disc) (TRT.c:657: t0 = q / (2a)) it takes
 *         q > 2^-500  and  q^2 (1 + 2^-30) <= lo,   lo = (light_d2 - e2) 4a (1 - 2^-30)                       as proof of "dark",
 *         q^2 >= hi,                                hi = ((light_d2 + 1.3e-6)(1 + 1.3e-6) + e2) 4a (1 + 2^-30)  as proof of "beyond",
 *     and calls the lane UNSURE for any other q > 0 (its wave then runs the closest-hit search instead: a blocker about as far as
 *     the light, about one wave in 1e4).  q > 2^-500 rules out an underflow of t0 (a is 1 up to 2^-40), so t0 > 0 as TRT.c:659
 *     demands.  d2_j against q: p_j = o + t0 d and o - p_j carry <= u (|o|_inf + 3.1 |t0 d_k|) of rounding per component, so
 *     | d2_j - t0^2 a | <= 12u t0^2 a + 2 sqrt(3) u Mg |t0| sqrt(a) <= 14u t0^2 a + 2^-52 Mg (rg + Mg)  (AM-GM; any t0), and
 *     t0^2 a = q^2/(4a) (1 +- 2.1u):  | d2_j - q^2/(4a) | <= 2^-40 q^2/(4a) + e2/16  with  e2 = 2^-48 Mg (rg + Mg).  The ground
 *     likewise with q := 2 a t (TRT.c:685: t > 1e-5).  Mg = |l|_inf + 2 rg bounds every coordinate of an origin the table admits
 *     and of a hit point nearer than the light (farther hit points only enter "beyond", where their own size drowns the nudge);
 *     if Mg > 2^28, dark_floor = +inf: lo = -1, hi = +inf, every hit is unsure (the closest-hit search, always).
 * (5) DEPTH (round 4).  The tables have a third coordinate -- `slabs` slabs of depth along a directional light's direction,
 *     `shells` shells of distance from a point light -- and a cell lists only the spheres that can matter to an origin of ITS slab.
 *     Directional: the exact test reports a hit only with t0 > 0, i.e. -b > sqrt(disc) >= 0, i.e. (c - o).d > 0 up to the
 *     rounding of that dot product (<= 2^-50 M): the sphere's CENTRE is ahead of the origin.  Slab s (s > 0) therefore holds the
 *     spheres whose centre's depth, plus delta for the FP32 look-up of the origin's depth (the same 10 eps rg as the plane
 *     coordinates), reaches the slab's lower edge, the edge lowered by 0.01 slab for the rounding of the slab coordinate; slab 0
 *     holds every sphere of the column (origins below the grid clamp into it), origins above the grid clamp into the top slab.
 *     Point: a hit nearer than the light lies at distance < R_o = |o - l| from it, so it can only be on a sphere with
 *     |c - l| - rho < R_o; a sphere with |c - l| - rho >= R_o + near has every point >= near farther from the light than the
 *     origin is: the ray meets it, if at all, beyond the light by >= near -- the case (3) shows to be "lit" whether listed or not.
 *     Shell s (not the outermost) holds the spheres with |c - l| - rho - near - delta below the shell's outer radius, the
 *     radius raised by 0.01 shell; the outermost shell holds every sphere of the cone (origins farther away clamp into it).
 *     tests/test_lightgrid.py drives this very classification with the oracle's shadow rays and adversarial ones: every
 *     "dark" and every "lit" it returns must be the reference's answer.
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
    float e3[3];        /* FP32 unit vector towards the light: depth = (o - g0).e3 */
    float w0;           /* depth of the lowest slab's lower edge */
    float inv_slab;     /* slabs per unit depth */
    float s_max;        /* slabs - 1 */
    int g;              /* cells per side */
    int slabs;          /* slabs of depth (5): the table has slabs * g * g cells, cell = (slab * g + row) * g + column */
    int words;          /* 64-bit words per cell = ceil(n / 64) */
    int pad_;
} trt_dirgrid;

typedef struct
{
    double l[3];  /* the light's position */
    double dark_floor; /* (4): a blocker nearer than the light proves "dark" only if light_d2 > dark_floor; +inf = never */
    double e2;         /* (4): absolute slack of d2_j against q^2 / (4a) */
    float half_g; /* g / 2 */
    float g_max;  /* g - 1 */
    float rg2;    /* admissible |o - l|^2 */
    float inv_shell; /* shells per unit distance from the light */
    float s_max;     /* shells - 1 */
    int g;        /* cells per face side */
    int shells;   /* shells of distance from the light (5): shells * 6 * g * g cells, cell = ((shell * 6 + face) * g + row) * g + column */
    int words;
} trt_pointgrid;

/* (4): the bounds a candidate's q^2 is compared with: q > 2^-500 and q^2 (1 + 2^-30) <= *lo proves "dark", q^2 >= *hi proves
 * "beyond the light" (ignore the hit), any other q > 0 is unsure.  a = d.d of the shadow ray (1 up to 2^-40 for every ray that is
 * looked up).  Without the guarantees of (4) (dark_floor = +inf, or NaN): *lo = -1, *hi = +inf, every hit is unsure. */
#define TRT_SHADOW_K1 (1.0 + 0x1p-30)
#define TRT_SHADOW_QMIN 0x1p-500
TRT_HD void trt_point_shadow_bounds(const trt_pointgrid *G, double light_d2, double a, double *lo, double *hi)
{
    const double four_a = 4.0 * a;
    const double l = (light_d2 - G->e2) * four_a * (1.0 - 0x1p-30);
    const double h = ((light_d2 + 1.3e-6) * (1.0 + 1.3e-6) + G->e2) * four_a * (1.0 + 0x1p-30);
    const int ok = light_d2 > G->dark_floor; /* false for NaN and for dark_floor = +inf */
    *lo = ok ? l : -1.0;
    *hi = ok ? h : __builtin_inf();
}

/* Cube-map face and face coordinates of a direction (x, y, z), AS gfx9's V_CUBEID / V_CUBESC / V_CUBETC / V_CUBEMA DEFINE THEM
 * (four instructions on the device; the C below restates the ISA manual's pseudo-code, ties included, and a GPU test compares the
 * two): the major axis is z if |z| >= |x| and |z| >= |y|, else y if |y| >= |x|, else x;
 *     face   +X 0   -X 1   +Y 2   -Y 3   +Z 4   -Z 5
 *     sc     -z     +z     +x     +x     +x     -x
 *     tc     -y     -y     +z     -z     -y     -y          *ma2 = 2 x the major component (signed)
 * so that the direction is proportional to trt_face_direction(face, sc / |major|, tc / |major|).  Every table over a cube map
 * (point lights, ray families, the patches of a sphere's surface) uses these face frames, builders and look-ups alike. */
TRT_HD void trt_cube_lookup(float x, float y, float z, int *face, float *sc, float *tc, float *ma2)
{
#if defined(__HIP_DEVICE_COMPILE__)
    *face = (int)__builtin_amdgcn_cubeid(x, y, z);
    *sc = __builtin_amdgcn_cubesc(x, y, z);
    *tc = __builtin_amdgcn_cubetc(x, y, z);
    *ma2 = __builtin_amdgcn_cubema(x, y, z);
#else
    /* the instructions COMPARE and DOUBLE with FP32 denormals flushed to zero, and pass sc and tc through as they are (observed:
     * tests/test_gpu_parity.py::test_the_cube_instructions_equal_their_c_restatement) */
    const float fx = __builtin_fabsf(x) < 1.17549435e-38f ? 0.0f * x : x, fy = __builtin_fabsf(y) < 1.17549435e-38f ? 0.0f * y : y,
                fz = __builtin_fabsf(z) < 1.17549435e-38f ? 0.0f * z : z;
    const float ax = __builtin_fabsf(fx), ay = __builtin_fabsf(fy), az = __builtin_fabsf(fz);
    if (az >= ax && az >= ay)
        *face = fz < 0.0f ? 5 : 4, *sc = fz < 0.0f ? -x : x, *tc = -y, *ma2 = fz * 2.0f;
    else if (ay >= ax)
        *face = fy < 0.0f ? 3 : 2, *sc = x, *tc = fy < 0.0f ? -z : z, *ma2 = fy * 2.0f;
    else
        *face = fx < 0.0f ? 1 : 0, *sc = fx < 0.0f ? z : -z, *tc = -y, *ma2 = fx * 2.0f;
#endif
}

/* direction of the point (u, v) of a cube-map face, major component +-1, in world axes: the frame of trt_cube_lookup */
TRT_HD void trt_face_direction(int face, double u, double v, double out[3])
{
    switch (face)
    {
    case 0: out[0] = 1.0, out[1] = -v, out[2] = -u; break;
    case 1: out[0] = -1.0, out[1] = -v, out[2] = u; break;
    case 2: out[0] = u, out[1] = 1.0, out[2] = v; break;
    case 3: out[0] = u, out[1] = -1.0, out[2] = -v; break;
    case 4: out[0] = u, out[1] = -v, out[2] = 1.0; break;
    default: out[0] = -u, out[1] = -v, out[2] = -1.0; break;
    }
}

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
    const float w = __builtin_fmaf(z, G->e3[2], __builtin_fmaf(y, G->e3[1], x * G->e3[0]));
    const float cs = __builtin_fminf(__builtin_fmaxf((w - G->w0) * G->inv_slab, 0.0f), G->s_max); /* NaN -> 0: the slab that holds everything */
    return ((int)cs * G->g + (int)cv) * G->g + (int)cu;
}

TRT_HD int trt_pointgrid_cell(const trt_pointgrid *G, double ox, double oy, double oz, int *far)
{
    const float w[3] = {(float)(ox - G->l[0]), (float)(oy - G->l[1]), (float)(oz - G->l[2])};
    const float r2 = __builtin_fmaf(w[2], w[2], __builtin_fmaf(w[1], w[1], w[0] * w[0]));
    *far = !(r2 <= G->rg2) || !(r2 > 0.0f);
    int face;
    float pu, pv, ma2;
    trt_cube_lookup(w[0], w[1], w[2], &face, &pu, &pv, &ma2);
#if defined(__HIP_DEVICE_COMPILE__)
    const float inv = __builtin_amdgcn_rcpf(0.5f * __builtin_fabsf(ma2));
#else
    const float inv = 1.0f / (0.5f * __builtin_fabsf(ma2));
#endif
    float cu = __builtin_fmaf(pu * inv, G->half_g, G->half_g), cv = __builtin_fmaf(pv * inv, G->half_g, G->half_g);
    cu = __builtin_fminf(__builtin_fmaxf(cu, 0.0f), G->g_max);
    cv = __builtin_fminf(__builtin_fmaxf(cv, 0.0f), G->g_max);
    /* shell of the origin's distance from the light; min(x, s_max) with x first: NaN -> the outermost shell, which holds everything.
     * On the device the square root is the bare v_sqrt_f32 (1 ulp; no denormal scaling: r2 > 0 is far from denormal for any origin
     * that is not `far`): the builder's growth of every sphere by delta and of every shell by 0.01 shell covers an ulp many times */
#if defined(__HIP_DEVICE_COMPILE__)
    const float dist = __builtin_amdgcn_sqrtf(r2);
#else
    const float dist = __builtin_sqrtf(r2);
#endif
    const float cs = __builtin_fminf(dist * G->inv_shell, G->s_max);
    return (((int)cs * 6 + face) * G->g + (int)cv) * G->g + (int)cu;
}

/* ------------------------------------------------ builders ------------------------------------------------
 * A table is built in two steps: a per-light PREPARE on the host (O(N): grid placement, one small record per
 * sphere) and the marking of every (cell, sphere) pair by a predicate that uses only + - * / and sqrt -- IEEE
 * operations that round identically on the host and on the device -- so the device kernels (trt_tables.hip, one
 * thread per cell) and the host reference builders below (tests) produce the same tables bit for bit. */

/* sphere j of a chunk of 64 sits at bit 63 - j, the order the exact stage walks with count-leading-zeros */
TRT_HD void trt_lightgrid_set(unsigned long long *cell, int sphere) { cell[sphere >> 6] |= 0x8000000000000000ull >> (sphere & 63); }

/* a sphere as a directional light's grid sees it: disc centre and radius in CELL units, growth included */
typedef struct
{
    double pu, pv, rad;
    double zs; /* (5): depth of the centre, grown by delta, in SLAB units above the lowest slab's lower edge */
} trt_dirgrid_disc;

/* a sphere as a point light's cube map sees it: unit axis light -> centre, sine and cosine of the grown half-angle */
typedef struct
{
    double a[3];
    double sin_a, cos_a;
    double everywhere; /* != 0: the light is inside or next to the sphere, every cell holds it */
    double shell_lo;   /* (5), point lights only: |c - l| - rho - near - delta in SHELL units: the sphere matters from this shell outwards */
} trt_pointgrid_cone;

/* does the disc reach cell (c, j), the cell grown by 0.01 on every side? */
TRT_HD int trt_dirgrid_reaches(const trt_dirgrid_disc *s, int c, int j)
{
    const double x0 = (double)c - 0.01, x1 = (double)c + 1.01, y0 = (double)j - 0.01, y1 = (double)j + 1.01;
    const double nx = s->pu < x0 ? x0 - s->pu : (s->pu > x1 ? s->pu - x1 : 0.0);
    const double ny = s->pv < y0 ? y0 - s->pv : (s->pv > y1 ? s->pv - y1 : 0.0);
    return nx * nx + ny * ny <= s->rad * s->rad;
}

/* (5): does the sphere matter to origins of slab s of `slabs`?  Slab 0 holds every sphere of its column. */
TRT_HD int trt_dirgrid_in_slab(const trt_dirgrid_disc *s, int slab) { return slab == 0 || s->zs >= (double)slab - 0.01; }
/* (5): ... to origins of shell s of `shells`?  The outermost shell holds every sphere of its cone. */
TRT_HD int trt_pointgrid_in_shell(const trt_pointgrid_cone *s, int shell, int shells) { return shell == shells - 1 || s->shell_lo <= (double)shell + 1.01; }

/* tangent of (45 degrees + 1e-4 rad) and a little more: how far an edge cell stretches past the face's edge */
#define TRT_POINTGRID_EDGE 1.00021

/* One edge of a cell in the face frame (p, q, m): the directions (e, t, 1), t in [t0, t1] (p is the coordinate held
 * fixed at e).  Does the cone come within its half-angle of that arc's INTERIOR?  (End points are the cell's corners,
 * tested separately.)  The arc lies on the great circle with unit normal (1, 0, -e)/sqrt(1 + e^2); the angle between
 * the axis and that circle has sine |h|, and the nearest point of the circle is the foot f of the axis in its plane. */
TRT_HD int trt_cone_near_edge(double ap, double aq, double am, double sin_a, double e, double t0, double t1)
{
    const double nl = __builtin_sqrt(1.0 + e * e);
    const double h = (ap - e * am) / nl;
    if (!(__builtin_fabs(h) <= sin_a))
        return 0;
    const double fq = aq, fm = am + h * e / nl; /* foot = a - h n, n = (1, 0, -e)/nl; its p-coordinate is not needed */
    return fm > 0.0 && fq >= t0 * fm && fq <= t1 * fm;
}

/* does the cone reach cell (c, j) of `face` (g cells per side)?  Cell grown by 0.01 cell; edge cells stretch to
 * TRT_POINTGRID_EDGE since the look-up clamps into them. */
TRT_HD int trt_pointgrid_reaches(const trt_pointgrid_cone *s, int face, int c, int j, int g)
{
    if (s->everywhere != 0.0)
        return 1;
    /* the cone's axis in the face's frame (trt_cube_lookup / trt_face_direction): (u, v, major), major > 0 on the face */
    double au, av, am;
    switch (face)
    {
    case 0: au = -s->a[2], av = -s->a[1], am = s->a[0]; break;
    case 1: au = s->a[2], av = -s->a[1], am = -s->a[0]; break;
    case 2: au = s->a[0], av = s->a[2], am = s->a[1]; break;
    case 3: au = s->a[0], av = -s->a[2], am = -s->a[1]; break;
    case 4: au = s->a[0], av = -s->a[1], am = s->a[2]; break;
    default: au = -s->a[0], av = -s->a[1], am = -s->a[2]; break;
    }
    const double step = 2.0 / (double)g;
    const double u0 = c == 0 ? -TRT_POINTGRID_EDGE : -1.0 + ((double)c - 0.01) * step;
    const double u1 = c == g - 1 ? TRT_POINTGRID_EDGE : -1.0 + ((double)c + 1.01) * step;
    const double v0 = j == 0 ? -TRT_POINTGRID_EDGE : -1.0 + ((double)j - 0.01) * step;
    const double v1 = j == g - 1 ? TRT_POINTGRID_EDGE : -1.0 + ((double)j + 1.01) * step;
    if (am > 0.0 && au >= u0 * am && au <= u1 * am && av >= v0 * am && av <= v1 * am)
        return 1; /* the axis itself points into the cell */
    const double cu[4] = {u0, u1, u1, u0}, cv[4] = {v0, v0, v1, v1};
    for (int q = 0; q < 4; q++) /* a corner inside the cone: cos(angle) >= cos(half-angle) */
        if ((au * cu[q] + av * cv[q] + am) / __builtin_sqrt(cu[q] * cu[q] + cv[q] * cv[q] + 1.0) >= s->cos_a)
            return 1;
    return trt_cone_near_edge(au, av, am, s->sin_a, u0, v0, v1) || trt_cone_near_edge(au, av, am, s->sin_a, u1, v0, v1) ||
           trt_cone_near_edge(av, au, am, s->sin_a, v0, u0, u1) || trt_cone_near_edge(av, au, am, s->sin_a, v1, u0, u1);
}

/* Directional light with to-light direction `to_light` (TRT.c:903-904; normalised here).  Fills G and one disc per
 * sphere (`discs` must hold n).  cs: the culling table's scene constants (shift and bounds). */
static inline void trt_dirgrid_prepare(const double *spheres, int n, const trt_cull_scene *cs, const double to_light[3], int g, int slabs,
                                       trt_dirgrid *G, trt_dirgrid_disc *discs)
{
    if (slabs < 1)
        slabs = 1;
    const int words = (n + 63) / 64 > 0 ? (n + 63) / 64 : 1;
    const double reach = (double)cs->cn + (double)cs->rm;
    const double rg = TRT_LIGHTGRID_RANGE * reach + 1.0;
    const double M = rg + reach;
    const double E = 0x1p-37 * M * M, delta = 3e-6 * rg;
    /* orthonormal basis across the light direction, in double; the look-up uses its FP32 rounding */
    double d[3] = {to_light[0], to_light[1], to_light[2]};
    const double dl = __builtin_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int k = 0; k < 3; k++)
        d[k] /= dl;
    int thin = 0;
    for (int k = 1; k < 3; k++)
        if (__builtin_fabs(d[k]) < __builtin_fabs(d[thin]))
            thin = k;
    double t[3] = {0, 0, 0};
    t[thin] = 1.0;
    double e1[3] = {d[1] * t[2] - d[2] * t[1], d[2] * t[0] - d[0] * t[2], d[0] * t[1] - d[1] * t[0]};
    const double e1l = __builtin_sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    for (int k = 0; k < 3; k++)
        e1[k] /= e1l;
    const double e2[3] = {d[1] * e1[2] - d[2] * e1[1], d[2] * e1[0] - d[0] * e1[2], d[0] * e1[1] - d[1] * e1[0]};
    /* discs in plane coordinates: centre, radius rho + delta; depth of the centre along the light direction */
    double lo[2] = {0, 0}, hi[2] = {0, 0}, zlo = 0, zhi = 0;
    for (int i = 0; i < n; i++)
    {
        const double *s = spheres + 9 * i;
        const double C[3] = {s[0] - cs->c0[0], s[1] - cs->c0[1], s[2] - cs->c0[2]};
        const double pu = C[0] * e1[0] + C[1] * e1[1] + C[2] * e1[2], pv = C[0] * e2[0] + C[1] * e2[1] + C[2] * e2[2];
        const double rad = __builtin_sqrt(s[3] * s[3] + E) + delta;
        const double z = C[0] * d[0] + C[1] * d[1] + C[2] * d[2];
        discs[i].pu = pu, discs[i].pv = pv, discs[i].rad = rad, discs[i].zs = z;
        zlo = (i == 0 || z < zlo) ? z : zlo;
        zhi = (i == 0 || z > zhi) ? z : zhi;
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
    G->pad_ = 0;
    /* (5) slabs of depth over the centres' range (origins ahead of every centre clamp into the top slab, whose columns are short) */
    double zext = zhi - zlo;
    if (!(zext > 1e-9))
        zext = 1e-9;
    for (int k = 0; k < 3; k++)
        G->e3[k] = (float)d[k];
    G->w0 = (float)zlo;
    G->inv_slab = (float)((double)slabs / zext);
    G->s_max = (float)(slabs - 1);
    G->slabs = slabs;
    const double inv = (double)G->inv_cell, u0 = (double)G->u0, v0 = (double)G->v0; /* the look-up's own constants */
    const double w0 = (double)G->w0, inv_slab = (double)G->inv_slab;
    for (int i = 0; i < n; i++)
    { /* to cell units / slab units */
        discs[i].pu = (discs[i].pu - u0) * inv;
        discs[i].pv = (discs[i].pv - v0) * inv;
        discs[i].rad = discs[i].rad * inv;
        discs[i].zs = (discs[i].zs + delta - w0) * inv_slab;
    }
}

/* Point light at `light` (TRT.c:930).  Fills G and one cone per sphere (`cones` must hold n). */
static inline void trt_pointgrid_prepare(const double *spheres, int n, const trt_cull_scene *cs, const double light[3], int g, int shells,
                                         trt_pointgrid *G, trt_pointgrid_cone *cones)
{
    if (shells < 1)
        shells = 1;
    const int words = (n + 63) / 64 > 0 ? (n + 63) / 64 : 1;
    const double reach = (double)cs->cn + (double)cs->rm;
    const double lc[3] = {light[0] - cs->c0[0], light[1] - cs->c0[1], light[2] - cs->c0[2]};
    const double away = __builtin_sqrt(lc[0] * lc[0] + lc[1] * lc[1] + lc[2] * lc[2]); /* light to the scene's centre */
    const double rg = TRT_LIGHTGRID_RANGE * (reach + away) + 1.0;
    const double M = rg + away + reach;
    const double E = 0x1p-37 * M * M, near = 0.02 + 4e-6 * rg, delta = 3e-6 * rg;
    const double sin_grow = 1.0000000000e-5, cos_grow = 0.99999999995; /* sin and cos of the 1e-5 rad the cones grow by */
    for (int k = 0; k < 3; k++)
        G->l[k] = light[k];
    G->half_g = (float)(0.5 * g);
    G->g_max = (float)(g - 1);
    G->rg2 = (float)(rg * rg * (1.0 - 1e-6));
    G->g = g;
    G->words = words;
    {
        double li = __builtin_fabs(light[0]);
        li = __builtin_fabs(light[1]) > li ? __builtin_fabs(light[1]) : li;
        li = __builtin_fabs(light[2]) > li ? __builtin_fabs(light[2]) : li;
        const double Mg = li + 2.0 * rg;
        G->dark_floor = Mg <= 0x1p28 ? 0x1p-60 * Mg * Mg : __builtin_inf(); /* also +inf for NaN */
        G->e2 = 0x1p-48 * Mg * (rg + Mg);
    }
    /* (5) shells of distance from the light, out to the farthest sphere (origins farther away clamp into the outermost shell) */
    double far_d = 0.0;
    for (int i = 0; i < n; i++)
    {
        const double *s = spheres + 9 * i;
        const double a[3] = {s[0] - light[0], s[1] - light[1], s[2] - light[2]};
        const double D = __builtin_sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) + __builtin_fabs(s[3]);
        far_d = D > far_d ? D : far_d; /* NaN: stays */
    }
    if (!(far_d > 1e-9))
        far_d = 1e-9;
    G->inv_shell = (float)((double)shells / far_d);
    G->s_max = (float)(shells - 1);
    G->shells = shells;
    const double inv_shell = (double)G->inv_shell;
    for (int i = 0; i < n; i++)
    {
        const double *s = spheres + 9 * i;
        const double a[3] = {s[0] - light[0], s[1] - light[1], s[2] - light[2]};
        const double D = __builtin_sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        const double rho = __builtin_sqrt(s[3] * s[3] + E) + 0x1p-45 * M;
        trt_pointgrid_cone *c = cones + i;
        c->shell_lo = (D - rho - near - delta) * inv_shell;
        c->everywhere = (!(D > rho + near) || !(rho / D < 0.999999)) ? 1.0 : 0.0; /* the light is inside or next to the sphere */
        if (c->everywhere != 0.0)
        {
            c->a[0] = c->a[1] = c->a[2] = 0.0;
            c->sin_a = 1.0, c->cos_a = 0.0;
            continue;
        }
        for (int k = 0; k < 3; k++)
            c->a[k] = a[k] / D;
        const double sn = rho / D, cn = __builtin_sqrt(1.0 - sn * sn);
        const double sg = sn * cos_grow + cn * sin_grow + 1e-12, cg = cn * cos_grow - sn * sin_grow - 1e-12; /* half-angle + 1e-5 rad */
        c->sin_a = sg < 1.0 ? sg : 1.0;
        c->cos_a = cg > 0.0 ? cg : 0.0;
        if (!(cg > 0.0)) /* grown past 90 degrees: treat like a sphere next to the light */
            c->everywhere = 1.0;
    }
}

/* Host reference builders (tests; the library builds the same tables on the device).  Return the bits set. */
static inline long trt_dirgrid_build(const double *spheres, int n, const trt_cull_scene *cs, const double to_light[3], int g, int slabs, trt_dirgrid *G,
                                     unsigned long long *masks, trt_dirgrid_disc *discs)
{
    trt_dirgrid_prepare(spheres, n, cs, to_light, g, slabs, G, discs);
    const int words = G->words;
    long bits = 0;
    for (long cell = 0; cell < (long)G->slabs * g * g; cell++)
    {
        unsigned long long *m = masks + cell * words;
        for (int w = 0; w < words; w++)
            m[w] = 0;
        for (int i = 0; i < n; i++)
            if (trt_dirgrid_in_slab(discs + i, (int)(cell / ((long)g * g))) && trt_dirgrid_reaches(discs + i, (int)(cell % g), (int)((cell / g) % g)))
            {
                trt_lightgrid_set(m, i);
                bits++;
            }
    }
    return bits;
}

static inline long trt_pointgrid_build(const double *spheres, int n, const trt_cull_scene *cs, const double light[3], int g, int shells, trt_pointgrid *G,
                                       unsigned long long *masks, trt_pointgrid_cone *cones)
{
    trt_pointgrid_prepare(spheres, n, cs, light, g, shells, G, cones);
    const int words = G->words;
    long bits = 0;
    for (long cell = 0; cell < 6L * G->shells * g * g; cell++)
    {
        unsigned long long *m = masks + cell * words;
        for (int w = 0; w < words; w++)
            m[w] = 0;
        const int shell = (int)(cell / (6L * g * g)), face = (int)((cell / ((long)g * g)) % 6), j = (int)((cell / g) % g), c = (int)(cell % g);
        for (int i = 0; i < n; i++)
            if (trt_pointgrid_in_shell(cones + i, shell, G->shells) && trt_pointgrid_reaches(cones + i, face, c, j, g))
            {
                trt_lightgrid_set(m, i);
                bits++;
            }
    }
    return bits;
}

#endif
