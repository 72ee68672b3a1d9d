// trt_rounds.hpp -- production frame-producer kernel for gfx950: persistent waves, MODE-SYNCHRONOUS ROUNDS.
//
// Shape of the work: per pixel `spp` samples, per sample a bounce loop, per bounce one
// closest-hit trace plus one shadow trace per light; a trace = N sphere tests + 1 plane test.
//
// Work units are single SAMPLES (unit = pixel*spp + k), pulled from a global queue a chunk at a time into a
// wave-level pool.  A lane always owns exactly one pending PATH ray.  One round of the main loop advances
// every lane's path by one ray:
//
//   P      closest hit of the path ray, all lanes (TRT.c:1024)
//          sky  -> skybox texel, the sample ends (TRT.c:858-867, 1044-1048)
//          hit  -> nudged point, unit normal, material (TRT.c:868-886)
//   S(i)   for every light i (wave-uniform loop): shadow ray from the hit point for the lanes that hit something
//          (TRT.c:900-957); directional lights only ask "anything in the way?", so their search stops at the
//          first hit; lit colour accumulates in the reference's light order
//   END    bounce bookkeeping (TRT.c:1034-1056); a finished sample is normalised by its weight (TRT.c:1061),
//          written to f.samples[unit] and the lane pulls its next unit; one shared unit() gives every lane its
//          next direction (mirror reflection or fresh primary ray)
//
// Because all lanes of a wave are in the same stage, there is no per-lane state machine and no merging of
// modes; the code of each stage is straight-line with exec masking only for "hit" vs "sky".  The price is
// that lanes whose path ray reached the sky idle through the shadow stages of that round.
//
// The trace itself is a two-phase search.  Phase 1 proposes candidate spheres with ONE table look-up per ray: path rays
// come in families that (nearly) pass through one point -- the eye, its mirror image in the ground, a sphere, a sphere's
// mirror image -- and each family has a cube map of direction cells (trt_raygrid.h); the shadow rays of one light are a
// two-parameter family too (trt_lightgrid.h).  A cell is one 64-bit word holding up to seven sphere indices (longer lists
// live in a pool).  A ray that does not pass its family's membership test, or whose origin is outside a light table's
// range, makes its wave fall back to the wave-uniform FP32 sweep over a culling table in LDS (trt_filter.h: broadcast
// ds_read_b128, 9 VALU per sphere, verdict in a sign bit).  Phase 2 is per lane: the EXACT FP64 test of the few candidates
// in ascending index order.  No table or filter ever decides a hit.  FP64, contraction off: results are bit-identical to the
// reference.  The mean over a pixel's samples is formed by reduce_samples_kernel in the reference's order (TRT.c:1063-1065).
#pragma once

#include "trt_device.hpp"
#include "trt_filter.h"
#include "trt_lightgrid.h"
#include "trt_raygrid.h"
#include "trt_common.hpp"

// A/B switches of round 5's instruction-level changes (profiles/r05/f_ab_log.txt); 1 = as shipped unless noted
#ifndef TRT_OPT_DIRCONST
#define TRT_OPT_DIRCONST 1
#endif
#ifndef TRT_OPT_POOLCHECK
#define TRT_OPT_POOLCHECK 1 // the exact loops ask the lanes for a pool word only in the iterations that can need one
#endif
#ifndef TRT_OPT_EXPECT
#define TRT_OPT_EXPECT 1 // the whole-wave sweep and the closest-hit search of a point light are the rare ways: laid out as such
#endif
#if TRT_OPT_EXPECT
#define TRT_EXPECT_LIST(c) __builtin_expect(!!(c), 1)
#define TRT_EXPECT_RARE(c) __builtin_expect(!!(c), 0)
#else
#define TRT_EXPECT_LIST(c) (c)
#define TRT_EXPECT_RARE(c) (c)
#endif
#ifndef TRT_OPT_BZSZ
#define TRT_OPT_BZSZ 1
#endif

namespace trt
{


#define CONSTANT_AS __attribute__((address_space(4)))
// whole 64-bit words through the constant address space: scalar loads (the kernel's arguments; wave-uniform records)
template <class T>
TRT_DEV T load_kernel_argument(const char CONSTANT_AS *at)
{
    static_assert(sizeof(T) % 8 == 0 && alignof(T) == 8, "argument structs are whole 64-bit words");
    T out;
    const unsigned long long CONSTANT_AS *src = (const unsigned long long CONSTANT_AS *)at;
    unsigned long long *dst = (unsigned long long *)&out;
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 8; i++)
        dst[i] = src[i];
    return out;
}

// Candidate tables.  Every table cell is one 64-bit LIST CELL (trt_raygrid.h): up to seven sphere indices inline, longer
// lists in `pool`.
struct GridView
{
    // light-space tables of the shadow rays (trt_lightgrid.h): headers and one list cell per table cell
    const trt_dirgrid *dir;     // [num_dir]
    const trt_pointgrid *point; // [num_point]
    const unsigned long long *dir_lists, *point_lists;
    unsigned dir_stride, point_stride; // cells per light
    int enabled;
    int list_bits; // bits per entry of every list cell: 8, or 16 for scenes of more than 256 spheres (light tables only)
    // direction tables of the path rays' families (trt_raygrid.h): families 0 (eye) and 1 (mirror eye) have 6*g_eye^2 cells
    // each, from cell eye_at of path_lists (they change with the camera), then 2NP families (patch k of sphere i at i P + k, then
    // their mirror images) of 6*g_sph^2 cells, from cell sph_at (they change with the scene only)
    int path_enabled;
    int g_eye, g_sph;
    int patch_m, patch_count; // cells per face side of the origin's cube map (0: one family per sphere), P
    unsigned eye_at, sph_at;
    const unsigned long long *path_lists;
    const unsigned long long *pool;
    const double *sphere_fam; // per sphere: mirror image of the centre (3), then r_chk of the sphere's family (patch_m == 0) or |r|
    const double *patch_rec;  // per patch: t (3), rho | mirrored t (3), rho (TRT_PATCH_RECORD doubles)
    double rg2_sph;           // admissible |o - apex|^2 of the families of the spheres
    double slack0;            // membership radius of a patch's family = |r| rho + slack0 (TRT_FAMILY_SLACK more for its mirror image)
    trt_rayfamily eye[2];     // families 0 and 1, by value: they change with the camera
};

static_assert(sizeof(trt_rayfamily) == 8 * TRT_RAYFAMILY_DOUBLES, "families of the eye in the LDS image");
static_assert(sizeof(trt_dirgrid) == 8 * kDirGridDoubles && sizeof(trt_pointgrid) == 8 * kPointGridDoubles, "light-table headers in the LDS image");

constexpr int kDirRecord = 8; // doubles per directional light in the LDS image

struct LdsImage
{
    const float4 *cull;  // culling table {Cx,Cy,Cz,kk}
    const float4 *cull_dir; // per directional light: {Cx,Cy,Cz,kk - (C.d)^2}, `padded` entries each
    const double *sph;   // per sphere {cx, cy, cz, r^2}: one 32-byte record, 16-byte aligned, read with two ds_read_b128 by the exact
                         // test (measured 1.9 % faster than four ds_read_b64 from a structure of arrays)
    const double *mat;   // (n+2) x {colour, reflectivity, specularity}: spheres, ground even, ground odd
    const double *dir;   // per directional light (kDirRecord doubles): unit to-light (3), colour (3), then what every shadow ray towards it shares:
                         // a = d.d (TRT.c:643) and d.n of the ground test (TRT.c:679), formed once per workgroup from the same operands
    const double *pt;    // per point light: position (3), colour (3), intensity
    const double *b255;  // byte / 255.0
    const double *cam;   // basis x,y,z (9), eye (3), -screen_distance
    const double *jit;   // jitter x[spp], y[spp]
    const trt_dirgrid *dirgrid;     // headers of the light-space tables (trt_lightgrid.h), per directional light
    const trt_pointgrid *pointgrid; // per point light
    const double *fam;              // per sphere: mirror image of the centre (3), r_chk of its family or, with patches, |r| (trt_raygrid.h)
    const double *eye;              // the two families of the eye: 2 x {apex (3), r_chk, r_chk^2, rg^2}
    const double *patch;            // per patch of a sphere's surface: t (3), rho | mirrored t (3), rho
};

// LDS image of a workgroup: culling table {Cx,Cy,Cz,kk} (4 floats per sphere, 16-B aligned, first) | per sphere {cx,cy,cz,r^2} |
// mat[(n+2)*5] (spheres, ground even, ground odd) | dir lights: unit to-light(3) colour(3) | point lights: pos(3) colour(3)
// intensity | byte/255.0 [256] | camera | jitter x[spp] y[spp] | one fixed-direction culling table per directional
// light | headers of the light-space tables | per sphere {mirror centre, |r|} of the path-ray families | the two families
// of the eye | the patches of a sphere's surface (`patches` records).  rounds_lds_bytes and stage_lds_image must agree.
inline size_t rounds_lds_bytes(const SceneView &s, int spp, int patches)
{
    const size_t padded = ((size_t)s.num_spheres + kCullGroup - 1) / kCullGroup * kCullGroup;
    return sizeof(double) * (padded * 2 + (size_t)s.num_spheres * 4 + ((size_t)s.num_spheres + 2) * 5 + (size_t)s.num_dir * kDirRecord +
                             (size_t)s.num_point * 7 + 256 + kLdsCameraDoubles + 2 * (size_t)spp + 1 /* 16-B alignment */ +
                             (size_t)s.num_dir * padded * 2 + (size_t)s.num_dir * kDirGridDoubles + (size_t)s.num_point * kPointGridDoubles +
                             (size_t)s.num_spheres * 4 + 2 * TRT_RAYFAMILY_DOUBLES + (size_t)patches * TRT_PATCH_RECORD);
}

TRT_DEV LdsImage stage_lds_image(double *lds, const SceneView &s, const CullView &cull, const FrameView &f, const GridView &grids)
{
    const int n = s.num_spheres, nd = s.num_dir, np = s.num_point;
    float4 *l_cull = (float4 *)lds;
    double *l_sph = lds + cull.padded * 2; // 16-byte aligned: the culling table before it is whole float4s
    double *l_mat = l_sph + 4 * n, *l_dir = l_mat + (n + 2) * 5, *l_pt = l_dir + nd * kDirRecord, *l_255 = l_pt + np * 7;
    double *l_cam = l_255 + 256, *l_jit = l_cam + kLdsCameraDoubles;
    for (int i = threadIdx.x; i < cull.padded; i += blockDim.x)
        l_cull[i] = ((const float4 *)cull.table)[i];
    for (int i = threadIdx.x; i < n; i += blockDim.x)
    {
        const double *sp = s.spheres + (long)i * kSphereDoubles;
        l_sph[4 * i + 0] = sp[0];
        l_sph[4 * i + 1] = sp[1];
        l_sph[4 * i + 2] = sp[2];
        l_sph[4 * i + 3] = sp[3] * sp[3]; // radius*radius exactly as TRT.c:648 forms it
        for (int j = 0; j < 5; j++)
            l_mat[i * 5 + j] = sp[4 + j];
    }
    for (int i = threadIdx.x; i < 10; i += blockDim.x)
        l_mat[n * 5 + i] = s.ground[6 + i];
    for (int i = threadIdx.x; i < nd; i += blockDim.x)
    {
        const double *li = s.dir_lights + i * kDirLightDoubles;
        const d3 tl = unit(scale(load3(li), -1.0)); // TRT.c:903-904, the same value for every hit point
        l_dir[i * kDirRecord + 0] = tl.x, l_dir[i * kDirRecord + 1] = tl.y, l_dir[i * kDirRecord + 2] = tl.z;
        l_dir[i * kDirRecord + 3] = li[3], l_dir[i * kDirRecord + 4] = li[4], l_dir[i * kDirRecord + 5] = li[5];
        l_dir[i * kDirRecord + 6] = dot(tl, tl);                     // a of ray_intersects_sphere (TRT.c:643) for every shadow ray towards this light
        l_dir[i * kDirRecord + 7] = dot(tl, load3(s.ground + 3)); // d.n of ray_intersects_plane (TRT.c:679)
    }
    for (int i = threadIdx.x; i < np * 7; i += blockDim.x)
        l_pt[i] = s.point_lights[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x)
        l_255[i] = (double)i / 255.0; // TRT.c:866
    for (int i = threadIdx.x; i < 12; i += blockDim.x)
        l_cam[i] = f.cam[i];
    if (threadIdx.x < 3) // TRT.c:989, :1000-1002: basis z scaled by sz = -screen_distance, the same product for every primary ray
        l_cam[13 + threadIdx.x] = f.cam[6 + threadIdx.x] * -f.cam[12];
    if (threadIdx.x == 0)
        l_cam[12] = -f.cam[12];
    for (int i = threadIdx.x; i < 2 * f.spp; i += blockDim.x)
        l_jit[i] = f.jitter[i];
    // fixed-direction tables behind everything else, on a 16-byte boundary (all offsets above are whole doubles)
    const long dir_tables_at = ((l_jit + 2 * f.spp) - lds + 1) & ~1L;
    float4 *l_cull_dir = (float4 *)(lds + dir_tables_at);
    __syncthreads();
    for (int i = threadIdx.x; i < nd * cull.padded; i += blockDim.x)
    {
        const int li = i / cull.padded, j = i - li * cull.padded;
        const float4 e = l_cull[j];
        const float dx = (float)l_dir[li * kDirRecord + 0], dy = (float)l_dir[li * kDirRecord + 1], dz = (float)l_dir[li * kDirRecord + 2]; // = trt_filter_setup's d
        l_cull_dir[i] = float4{e.x, e.y, e.z, trt_filter_fixed_dir_kk(e.x, e.y, e.z, e.w, dx, dy, dz)};
    }
    // headers of the light-space tables behind the fixed-direction tables (whole doubles again: 4 floats per entry)
    double *l_dirgrid = (double *)(l_cull_dir + nd * cull.padded), *l_pointgrid = l_dirgrid + nd * kDirGridDoubles;
    if (grids.enabled)
    {
        for (int i = threadIdx.x; i < nd * kDirGridDoubles; i += blockDim.x)
            l_dirgrid[i] = ((const double *)grids.dir)[i];
        for (int i = threadIdx.x; i < np * kPointGridDoubles; i += blockDim.x)
            l_pointgrid[i] = ((const double *)grids.point)[i];
    }
    double *l_fam = l_pointgrid + np * kPointGridDoubles, *l_eye = l_fam + 4 * n, *l_patch = l_eye + 2 * TRT_RAYFAMILY_DOUBLES;
    if (grids.path_enabled)
    {
        for (int i = threadIdx.x; i < n * 4; i += blockDim.x)
            l_fam[i] = grids.sphere_fam[i];
        for (int i = threadIdx.x; i < 2 * TRT_RAYFAMILY_DOUBLES; i += blockDim.x)
            l_eye[i] = ((const double *)grids.eye)[i];
        for (int i = threadIdx.x; i < grids.patch_count * TRT_PATCH_RECORD; i += blockDim.x)
            l_patch[i] = grids.patch_rec[i];
    }
    __syncthreads();
    // A directional light's table is for rays along its UNIT direction (trt_lightgrid.h): a light whose direction could not be
    // normalised (TRT.c:444 leaves vectors shorter than 1e-4 alone) gets a range no origin is in -- "far" for every ray, its waves
    // sweep -- here, once per workgroup, instead of the shading stage testing |sd.sd - 1| for every light in every pass.
    if (grids.enabled)
    {
        for (int i = threadIdx.x; i < nd; i += blockDim.x)
        {
            const d3 sd = load3(l_dir + i * kDirRecord);
            if (!(__builtin_fabs(dot(sd, sd) - 1.0) <= 9.094947017729282e-13))
                ((trt_dirgrid *)l_dirgrid)[i].rg2 = -1.0f;
        }
        __syncthreads();
    }
    return LdsImage{l_cull, l_cull_dir, l_sph, l_mat, l_dir, l_pt, l_255, l_cam, l_jit,
                    (const trt_dirgrid *)l_dirgrid, (const trt_pointgrid *)l_pointgrid, l_fam, l_eye, l_patch};
}

struct Hit
{
    double d2; // squared distance origin -> (un-nudged) hit point, TRT.c:815
    d3 p;      // hit point as the intersection routine produced it
    int i;     // -1: nothing; [0,n): sphere; n: ground
    double t;  // while the spheres are searched: the ray parameter of the best hit.  Its hit point p = o + t d (TRT.c:664-666) is
               // formed once, after the loop -- the same expression on the same operands: the same bits -- so that an iteration
               // replaces five registers with selects, not nine (round 3 measured this form no faster; on round 4's kernel, whose
               // VALU is the bound, it is: C5 +2.6 %)
};


// One EXACT sphere test of TRT.c:638-672 in the reference's operation order, folded into the running closest hit of
// TRT.c:808-826 (strict '<': the first index wins ties).  Predicated rather than branched: a lane without a candidate
// (`valid` false) tests sphere 0 and discards the result.  Returns true when an ANY_HIT search is answered.
// `inside` (refraction extension only, -1 otherwise): the sphere the ray travels inside of is met at the FAR root.
// UNIT_A (the candidates come from a table: every ray that is looked up has |a - 1| <= 2^-40): an ANY_HIT search does not form
// t0 = q / (2a), q = -b - sqrt(disc), at all -- t0 > 0 (TRT.c:659) holds iff q > 0 unless the quotient underflows, which a
// q > 2^-500 cannot; a q in (0, 2^-500] is divided as the reference divides it.
template <bool ANY_HIT, bool REFRACT = false, bool UNIT_A = false>
TRT_DEV bool exact_step(const LdsImage &L, d3 o, d3 d, double a, int i, bool valid, Hit &best, int inside = -1)
{
    static_assert(!(ANY_HIT && REFRACT), "shadow rays are the reference's");
    const double2 c01 = ((const double2 *)L.sph)[2 * i], c23 = ((const double2 *)L.sph)[2 * i + 1];
    const d3 c = d3{c01.x, c01.y, c23.x};
    const d3 oc = sub(o, c);
    const double b = 2.0 * dot(oc, d);
    const double cc = dot(oc, oc) - c23.y;
    const double disc = b * b - 4.0 * a * cc;
    const bool far_root = REFRACT && i == inside;
    bool done = false;
    if (valid && !(disc < 0.0) && (b < 0.0 || far_root)) // b >= 0  =>  -b - sqrt(disc) <= 0  =>  t0 <= 0 or NaN: a miss (TRT.c:657-659)
    { // no further branches: selects keep `best` in place (nested ifs cost five register copies per level)
        const double root = sqrt_exact(disc);
        if (ANY_HIT)
        { // "anything in the way?" (TRT.c:908): the first hit answers it; its position is never used
            const double q = -b - root;
            bool hit = UNIT_A ? q > TRT_SHADOW_QMIN : q / (2.0 * a) > 0.0;
            if (UNIT_A && q > 0.0 && !hit)
                hit = q / (2.0 * a) > 0.0;
            best.i = hit ? i : best.i;
            done = hit;
        }
        else
        {
            const double t0 = (far_root ? -b + root : -b - root) / (2.0 * a);
            const bool hit = t0 > 0.0;
            const d3 p = d3{o.x + t0 * d.x, o.y + t0 * d.y, o.z + t0 * d.z};
            const double d2 = dist2(o, p);
            const bool closer = hit && d2 < best.d2;
            best.d2 = closer ? d2 : best.d2;
            best.t = closer ? t0 : best.t;
            best.i = closer ? i : best.i;
        }
    }
    return done;
}

// Closest hit of TRT.c:793-856 for the lanes with `active`.  ANY_HIT: the caller only asks whether anything
// is hit (directional-light shadow ray, TRT.c:908), so a lane stops at its first hit and the ground is skipped
// once a sphere was found.
// `use_list` (wave-uniform): every active lane's candidates are the entries of its LIST CELL `cell` (trt_raygrid.h;
// indices ascending, up to seven inline, longer lists in `pool`).  Otherwise the wave sweeps the FP32 culling table
// (trt_filter.h) -- `fixed` != nullptr: all rays share the direction that culling table was built for -- and each lane
// pops its candidate bits.
// TRT_LIST_PREFILTER: lists longer than this many entries in some lane of the wave are first thinned by the FP32 filter
// (0 = never).
#ifndef TRT_LIST_PREFILTER
#define TRT_LIST_PREFILTER 12
#endif


template <bool ANY_HIT, bool REFRACT = false, int MARK_BASE = 0>
TRT_DEV Hit trace(const LdsImage &L, const CullView &cull, int n, d3 o, d3 d, bool active, d3 gp, d3 gn, unsigned &phase2_rounds, unsigned &lane_tests,
                  const float4 *fixed, bool use_list, unsigned long long cell, const unsigned long long *pool, int list_bits, int inside = -1,
                  const double *shared_ad = nullptr // every lane's ray has the SAME direction (a directional light's shadow rays): {d.d, d.gn} from the LDS image
#if TRT_STAMP
                  ,
                  unsigned long long *stamp_sum = nullptr, unsigned long long *stamp_prev_p = nullptr, int stamp_base = 0
#endif
)
{
#if TRT_STAMP
    unsigned long long &stamp_prev = *stamp_prev_p;
#define TRT_TRACE_STAMP(k) TRT_STAMP_AT(stamp_base + (k))
#elif defined(TRT_MARKS) && TRT_MARKS == 2
#define TRT_TRACE_STAMP(k) TRT_STAMP_AT(MARK_BASE + (k)) // MARK_BASE: the call site's slots (a template parameter: an immediate)
#elif defined(TRT_MARKS)
#define TRT_TRACE_STAMP(k) TRT_STAMP_AT(trace_##k)
#else
#define TRT_TRACE_STAMP(k) \
    do                     \
    {                      \
    } while (0)
#endif
    (void)fixed;
    Hit best;
    best.d2 = __builtin_inf();
    best.p = o;
    best.i = -1;
    best.t = 0.0;
    if (!TRT_OPT_DIRCONST)
        shared_ad = nullptr;
    const double a = shared_ad ? shared_ad[0] : dot(d, d);
    if (TRT_EXPECT_LIST(use_list))
    {
        const unsigned ctl = (unsigned)(cell >> 56);
        bool pooled = (ctl & TRT_LIST_POOLED) != 0;
        int count = active ? (pooled ? (int)((cell >> 32) & 0xffffu) : (int)ctl) : 0;
        const unsigned at = (unsigned)cell; // pooled: offset of the list's words
        const unsigned entry_mask = (1u << list_bits) - 1u;
        const int per_shift = list_bits == 8 ? 3 : 2, per_mask = (1 << per_shift) - 1; // 8 or 4 entries per pool word
        unsigned long long cur = cell;
        int k = 0;
#if TRT_STAMP == 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        TRT_TRACE_STAMP(0); // table load
#if TRT_LIST_PREFILTER
        // Long lists (dense scenes): the FP32 filter of trt_filter.h first goes over the list -- 9 FP32 operations per entry
        // instead of ~19 FP64 -- and only the entries it cannot reject go to the exact test.  Like the sweep it never decides
        // a hit.  Up to eight survivors (four of 16 bits) fit one 64-bit word; if some lane has more, the wave tests its lists directly.
        if (__any(count > TRT_LIST_PREFILTER))
        {
            trt_ray_filter flt;
            trt_filter_setup(&flt, o.x, o.y, o.z, d.x, d.y, d.z, a, cull.c0x, cull.c0y, cull.c0z, cull.cn, cull.rm);
            unsigned long long kept = 0, word = cell;
            int nk = 0;
            bool over = false;
            for (int j = 0; __any(j < count); j++)
            {
                const bool valid = j < count;
#if TRT_OPT_POOLCHECK
                if ((j & per_mask) == 0)
#endif
                if (__any(valid && pooled && (j & per_mask) == 0))
                    if (valid && pooled && (j & per_mask) == 0)
                        word = pool[at + ((unsigned)j >> per_shift)];
                const unsigned i = valid ? (unsigned)word & entry_mask : 0u;
                word >>= list_bits;
                unsigned sign;
                if (ANY_HIT && fixed)
                {
                    const float4 e = fixed[i];
                    sign = trt_filter_sign_fixed_dir(&flt, e.x, e.y, e.z, e.w);
                }
                else
                {
                    const float4 e = L.cull[i];
                    sign = trt_filter_sign(&flt, e.x, e.y, e.z, e.w);
                }
                if (valid && (!flt.ok || !(sign >> 31)))
                {
                    over = over || nk > per_mask;
                    kept |= nk <= per_mask ? (unsigned long long)i << (list_bits * nk) : 0ull;
                    nk++;
                }
            }
            if (!__any(over))
            { // the survivors are the list now: ascending as before
                cur = kept;
                count = nk;
                pooled = false;
            }
        }
#endif
        TRT_TRACE_STAMP(1);
        while (__any(k < count))
        {
            phase2_rounds++;
            const bool valid = k < count;
            lane_tests += valid;
#if TRT_OPT_POOLCHECK
            if ((k & per_mask) == 0) // k is the wave's: seven iterations of eight ask nothing of the lanes
#endif
            if (__any(valid && pooled && (k & per_mask) == 0))
                if (valid && pooled && (k & per_mask) == 0)
                    cur = pool[at + ((unsigned)k >> per_shift)];
            const int i = valid ? (int)((unsigned)cur & entry_mask) : 0;
            cur >>= list_bits;
            k++;
            if (exact_step<ANY_HIT, REFRACT, ANY_HIT>(L, o, d, a, i, valid, best, inside))
                count = 0;
        }
    }
    else
    {
        trt_ray_filter flt;
        trt_filter_setup(&flt, o.x, o.y, o.z, d.x, d.y, d.z, a, cull.c0x, cull.c0y, cull.c0z, cull.cn, cull.rm);
        TRT_TRACE_STAMP(0);
        for (int base = 0; base < cull.padded; base += 64)
        {
            // phase 1: wave-uniform sweep; each verdict is a sign bit shifted into a per-lane word by v_alignbit,
            // sphere base+j ends up at bit 63-j of `cand`
            const int chunk = (cull.padded - base) < 64 ? (cull.padded - base) : 64;
            unsigned word[2];
#pragma unroll
            for (int h = 0; h < 2; h++)
            {
                const int first = base + 32 * h, count = (chunk - 32 * h) < 32 ? (chunk - 32 * h) : 32;
                unsigned bits = ~0u;
                for (int g = 0; g < count; g += kCullGroup)
                {
#pragma unroll
                    for (int j = 0; j < kCullGroup; j++)
                    {
                        if (ANY_HIT && fixed)
                        {
                            const float4 e = fixed[first + g + j];
                            bits = __builtin_amdgcn_alignbit(bits, trt_filter_sign_fixed_dir(&flt, e.x, e.y, e.z, e.w), 31);
                        }
                        else
                        {
                            const float4 e = L.cull[first + g + j]; // same address in every lane: LDS broadcast
                            bits = __builtin_amdgcn_alignbit(bits, trt_filter_sign(&flt, e.x, e.y, e.z, e.w), 31);
                        }
                    }
                }
                word[h] = count > 0 ? ~(bits << (32 - count)) & (count == 32 ? ~0u : ~((1u << (32 - count)) - 1u)) : 0u;
            }
            unsigned long long cand = ((unsigned long long)word[0] << 32) | word[1];
            if (!flt.ok)
                cand = chunk == 64 ? ~0ull : ~((1ull << (64 - chunk)) - 1ull); // degenerate ray: every sphere of the chunk
            if (!active)
                cand = 0;
            TRT_TRACE_STAMP(1); // sweep
            // phase 2: exact FP64 tests of this lane's candidates, ascending index (first index wins ties, TRT.c:816)
            while (__any(cand != 0))
            {
                phase2_rounds++;
                const int lead = __builtin_clzll(cand | 1ull);
                const bool valid = cand != 0 && base + lead < n;
                lane_tests += valid;
                const int i = valid ? base + lead : 0;
                cand &= ~(0x8000000000000000ull >> lead);
                if (exact_step<ANY_HIT, REFRACT>(L, o, d, a, i, valid, best, inside))
                    cand = 0ull;
            }
        }
    }
    if (!ANY_HIT && best.i >= 0)
        best.p = d3{o.x + best.t * d.x, o.y + best.t * d.y, o.z + best.t * d.z}; // the winner's hit point: the expression of TRT.c:664-666 again
    TRT_TRACE_STAMP(2); // exact tests
    // ground plane (TRT.c:831-853; ray_intersects_plane TRT.c:677-695).  One wave-level decision, then straight-line code with
    // selects: a ray can only hit if |d.n| > 1e-5 and numerator and denominator of t have the same sign (opposite signs: t <= 0,
    // whatever the quotient's digits are), so a wave whose rays all head away from the plane skips the division.
    {
        const double denom = shared_ad ? shared_ad[1] : dot(d, gn), num = dot(sub(gp, o), gn);
        const bool maybe = active && !(ANY_HIT && best.i >= 0) && __builtin_fabs(denom) > 0.00001 &&
                           (long long)(__builtin_bit_cast(unsigned long long, num) ^ __builtin_bit_cast(unsigned long long, denom)) >= 0;
        if (__any(maybe))
        {
            const double t = num / denom;
            const bool hit = maybe && t > 0.00001;
            if (ANY_HIT)
                best.i = hit ? n : best.i; // nothing closer can matter: any hit blocks the light
            else
            {
                const d3 p = d3{o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};
                const double d2 = dist2(o, p);
                const bool closer = hit && d2 < best.d2;
                best.d2 = closer ? d2 : best.d2;
                best.p.x = closer ? p.x : best.p.x;
                best.p.y = closer ? p.y : best.p.y;
                best.p.z = closer ? p.z : best.p.z;
                best.i = closer ? n : best.i;
            }
        }
    }
    TRT_TRACE_STAMP(3); // plane
#undef TRT_TRACE_STAMP
    return best;
}

// ANY-HIT search for the shadow ray of a POINT light (trt_lightgrid.h (4); TRT.c:937-942 asks for the closest blocker and whether
// it is nearer than the light): the candidates of the ray's list cell, then the ground.  `dark`: some hit is provably nearer than
// the light -- q > 2^-500 and q^2 (1 + 2^-30) <= lo, q = -b - sqrt(disc) the numerator of the near root (for the ground
// q = 2 a t) -- which decides the reference's answer whichever hit is the closest; the lane stops there.  A hit provably BEYOND
// the light (q^2 >= hi) cannot matter and is passed over.  `unsure`: the lane met a hit (q > 0) that is neither, and found no
// proof of "dark".  Neither flag: nothing nearer than the light is hit, the point is lit.  No division, no hit point, no running
// minimum: 3 FP64 operations and two compares per hit on top of the discriminant and its root.
TRT_DEV void point_light_search(const LdsImage &L, int n, d3 o, d3 d, bool active, d3 gp, d3 gn, unsigned &rounds, unsigned &lane_tests,
                                unsigned long long cell, const unsigned long long *pool, int list_bits, double lo, double hi, bool &dark, bool &unsure)
{
    (void)n;
    const double a = dot(d, d);
    const unsigned ctl = (unsigned)(cell >> 56);
    const bool pooled = (ctl & TRT_LIST_POOLED) != 0;
    int count = active ? (pooled ? (int)((cell >> 32) & 0xffffu) : (int)ctl) : 0;
    const unsigned at = (unsigned)cell;
    const unsigned entry_mask = (1u << list_bits) - 1u;
    const int per_shift = list_bits == 8 ? 3 : 2, per_mask = (1 << per_shift) - 1;
    unsigned long long cur = cell;
    int k = 0;
    dark = false, unsure = false;
    while (__any(k < count))
    {
        rounds++;
        const bool valid = k < count;
        lane_tests += valid;
#if TRT_OPT_POOLCHECK
        if ((k & per_mask) == 0)
#endif
        if (__any(valid && pooled && (k & per_mask) == 0))
            if (valid && pooled && (k & per_mask) == 0)
                cur = pool[at + ((unsigned)k >> per_shift)];
        const int i = valid ? (int)((unsigned)cur & entry_mask) : 0;
        cur >>= list_bits;
        k++;
        const double2 c01 = ((const double2 *)L.sph)[2 * i], c23 = ((const double2 *)L.sph)[2 * i + 1];
        const d3 oc = sub(o, d3{c01.x, c01.y, c23.x});
        const double b = 2.0 * dot(oc, d);
        const double cc = dot(oc, oc) - c23.y;
        const double disc = b * b - 4.0 * a * cc;
        if (valid && !(disc < 0.0) && b < 0.0) // b >= 0: t0 <= 0 or NaN, a miss (TRT.c:657-659)
        {
            const double q = -b - sqrt_exact(disc), qq = q * q;
            const bool blocks = q > TRT_SHADOW_QMIN && qq * TRT_SHADOW_K1 <= lo;
            unsure = unsure || (q > 0.0 && !blocks && !(qq >= hi));
            dark = dark || blocks;
            count = blocks ? 0 : count;
        }
    }
    // the ground (TRT.c:677-695) for the lanes that are not dark yet: one wave-level decision, as in trace() -- a ray can only
    // hit if |d.n| > 1e-5 and numerator and denominator of t have the same sign (opposite signs: t <= 0)
    {
        const double denom = dot(d, gn), num = dot(sub(gp, o), gn);
        const bool maybe = active && !dark && __builtin_fabs(denom) > 0.00001 &&
                           (long long)(__builtin_bit_cast(unsigned long long, num) ^ __builtin_bit_cast(unsigned long long, denom)) >= 0;
        if (__any(maybe))
        {
            const double t = num / denom;
            const bool hit = maybe && t > 0.00001;
            const double qq = (t * t) * (4.0 * a) * a; // q = 2 a t
            const bool blocks = hit && qq * TRT_SHADOW_K1 <= lo;
            dark = dark || blocks;
            unsure = unsure || (hit && !blocks && !(qq >= hi));
        }
    }
    unsure = unsure && !dark;
}

// The list cell of a PATH ray (trt_raygrid.h), ONE FAMILY PER SPHERE (GridView::patch_m == 0).  `fam`: the family the ray is
// expected in (0 eye, 1 mirror eye, 2 + i sphere i, 2 + n + i mirror sphere i, < 0 none).  `fallback` is set for an active lane
// whose ray fails the family's membership test (its line must pass within r_chk of the apex, its origin not more than r_chk
// behind it, within the table's range; a unit direction) or whose cell has no list: the caller then sweeps.
TRT_DEV unsigned long long path_cell(const LdsImage &L, const GridView &G, int n, int fam, d3 o, d3 d, bool active, bool &fallback)
{
    const bool has = active && fam >= 0;
    const int f = has ? fam : 0;
    const int s = f >= 2 ? f - 2 : 0, i = s >= n ? s - n : s; // sphere of the family
    const bool of_eye = f < 2, mirrored = s >= n;
    const double *rec = L.fam + 4 * i; // mirror image of the centre, r_chk of the sphere's family
    d3 apex = mirrored ? load3(rec) : load3(L.sph + 4 * i);
    double r_chk = mirrored ? rec[3] + TRT_FAMILY_SLACK : rec[3];
    const double *E = L.eye + TRT_RAYFAMILY_DOUBLES * (f & 1);
    const double rg2_eye = E[5];
    if (of_eye)
    {
        apex = load3(E);
        r_chk = E[3];
    }
    const double rg2 = of_eye ? rg2_eye : G.rg2_sph; // by value: a pointer chosen between LDS and the kernel arguments is a flat load
    // trt_rayfamily_member, operation for operation (all four conditions evaluated: no branches between them)
    const d3 w = sub(o, apex);
    const d3 c = d3{w.y * d.z - w.z * d.y, w.z * d.x - w.x * d.z, w.x * d.y - w.y * d.x};
    const bool near_line = dot(c, c) <= r_chk * r_chk, ahead = dot(w, d) >= -r_chk, in_range = dot(w, w) <= rg2;
    const bool unit_dir = __builtin_fabs(dot(d, d) - 1.0) <= 9.094947017729282e-13;
    const bool member = near_line & ahead & in_range & unit_dir;
    const int g = of_eye ? G.g_eye : G.g_sph;
    const int at = trt_cubemap_cell((float)d.x, (float)d.y, (float)d.z, 0.5f * (float)g, (float)(g - 1), g);
    const unsigned eye_cells = 6u * (unsigned)G.g_eye * (unsigned)G.g_eye, sph_cells = 6u * (unsigned)G.g_sph * (unsigned)G.g_sph;
    const unsigned base = of_eye ? G.eye_at + (unsigned)f * eye_cells : G.sph_at + (unsigned)s * sph_cells;
    unsigned long long cell = 0;
    if (has && member)
        cell = G.path_lists[base + (unsigned)at];
    fallback = active && (!has || !member || (unsigned)(cell >> 56) == TRT_LIST_NONE);
    return cell;
}

// The same with the surface of every sphere cut into PATCHES that have a family each (GridView::patch_m > 0, trt_raygrid.h).
// `fam`: the family code of the ray (0 eye, 1 mirror eye, 2 + i: it starts on
// sphere i -- the patch of the sphere's surface follows from the origin --, 2 + n + (i << TRT_PATCH_SHIFT | k): reflected by the
// ground, its parent started on patch k of sphere i; < 0 none).  `fam` becomes the family code of the ray's REFLECTION BY THE
// GROUND, should it go on to hit the ground: the mirror family of this one's -- of the patch the ray started on -- or none (a ray
// from the ground cannot hit the ground again).  `fallback` is set for an active lane whose ray fails the family's
// membership test (its line must pass within r_chk of the apex, its origin not more than r_chk behind it, within the table's
// range; a unit direction) or whose cell has no list: the caller then sweeps.
TRT_DEV unsigned long long path_cell_patches(const LdsImage &L, const GridView &G, int n, int &fam, d3 o, d3 d, bool active, bool &fallback)
{
    const bool has = active && fam >= 0;
    const int f = has ? fam : 0;
    const bool of_eye = f < 2, mirrored = f >= 2 + n;
    const int code = f - 2 - n; // mirrored: sphere and patch
    const int i = mirrored ? code >> TRT_PATCH_SHIFT : (of_eye ? 0 : f - 2); // sphere of the family
    // ONE record read per lane, from wherever the family's base point lives (all of it is in LDS: the choice is a 32-bit
    // select of the address): the sphere's centre, its mirror image, or the eye / mirror eye, whose families are "patches" of
    // a sphere of radius 0 -- base + 0 t is the base, 0 rho + slack the slack
    const double *at_sphere = L.sph + 4 * i, *at_mirror = L.fam + 4 * i;
    const double *src = of_eye ? L.eye + TRT_RAYFAMILY_DOUBLES * (f & 1) : (mirrored ? at_mirror : at_sphere);
    const d3 base_at = load3(src);
    // which patch of its sphere the ray starts on (FP32: the membership test below is what counts)
    const int here = trt_cubemap_cell((float)(o.x - base_at.x), (float)(o.y - base_at.y), (float)(o.z - base_at.z), 0.5f * (float)G.patch_m,
                                      (float)(G.patch_m - 1), G.patch_m);
    const int k = mirrored ? code & ((1 << TRT_PATCH_SHIFT) - 1) : (of_eye ? 0 : here);
    fam = !has ? -1 : (f == 0 ? 1 : (!of_eye && !mirrored ? 2 + n + (((f - 2) << TRT_PATCH_SHIFT) | k) : -1)); // a ray without a family has no successor family either
    const double r_abs = of_eye ? 0.0 : at_mirror[3];
    const double *pr = L.patch + TRT_PATCH_RECORD * k + (mirrored ? 4 : 0);
    const d3 apex = d3{TRT_PATCH_APEX(base_at.x, r_abs, pr[0]), TRT_PATCH_APEX(base_at.y, r_abs, pr[1]), TRT_PATCH_APEX(base_at.z, r_abs, pr[2])};
    const double rho = pr[3];
    // the eye's own radius and range come from LDS, the spheres' from the kernel's arguments.  Values first, selects after: a
    // select between a POINTER into LDS and one to the (re-read) arguments is a flat load from a private copy
    double rchk_eye = src[3], rg2_eye = src[5]; // only meaningful where src is a family of the eye
    asm("" : "+v"(rchk_eye), "+v"(rg2_eye));
    double r_chk = TRT_PATCH_RCHK(r_abs, rho, G.slack0), rg2 = G.rg2_sph;
    r_chk = mirrored ? r_chk + TRT_FAMILY_SLACK : r_chk;
    r_chk = of_eye ? rchk_eye : r_chk;
    rg2 = of_eye ? rg2_eye : rg2;
    // trt_rayfamily_member, operation for operation (all four conditions evaluated: no branches between them)
    const d3 w = sub(o, apex);
    const d3 c = d3{w.y * d.z - w.z * d.y, w.z * d.x - w.x * d.z, w.x * d.y - w.y * d.x};
    const bool near_line = dot(c, c) <= r_chk * r_chk, ahead = dot(w, d) >= -r_chk, in_range = dot(w, w) <= rg2;
    const bool unit_dir = __builtin_fabs(dot(d, d) - 1.0) <= 9.094947017729282e-13;
    const bool member = near_line & ahead & in_range & unit_dir;
    const int g = of_eye ? G.g_eye : G.g_sph;
    const int at = trt_cubemap_cell((float)d.x, (float)d.y, (float)d.z, 0.5f * (float)g, (float)(g - 1), g);
    const unsigned eye_cells = 6u * (unsigned)G.g_eye * (unsigned)G.g_eye, sph_cells = 6u * (unsigned)G.g_sph * (unsigned)G.g_sph;
    const unsigned table = ((mirrored ? (unsigned)n : 0u) + (unsigned)i) * (unsigned)G.patch_count + (unsigned)k;
    const unsigned base = of_eye ? G.eye_at + (unsigned)f * eye_cells : G.sph_at + table * sph_cells;
    unsigned long long cell = 0;
    if (has && member)
        cell = G.path_lists[base + (unsigned)at];
    fallback = active && (!has || !member || (unsigned)(cell >> 56) == TRT_LIST_NONE);
    return cell;
}

// how many lanes below this one are set in `mask` (v_mbcnt: no lane mask to keep in registers)
TRT_DEV unsigned lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Per-lane counters of the counting kernel variant and the stamps of the diagnostic build, handed through the stages.
struct Tally
{
    unsigned path = 0, shadow = 0, rounds = 0, swept = 0, passes = 0;
    unsigned iters[3] = {0, 0, 0}; // wave-level iterations of the exact-test loops: path rays, directional-light shadow rays, point-light ones
    unsigned tests[3] = {0, 0, 0}; // exact tests of THIS lane in those loops (tests / (64 iters) = the loops' lane activity)
    unsigned full = 0;             // point-light shadow searches that fell back from the any-hit form to the closest-hit one
#if TRT_STAMP
    unsigned long long stamp_sum[24] = {0}, stamp_prev = 0;
#endif
};
#if TRT_STAMP
#define TRT_STAGE_STAMPS(t)                       \
    unsigned long long *const stamp_sum = (t).stamp_sum; \
    unsigned long long &stamp_prev = (t).stamp_prev
#else
#define TRT_STAGE_STAMPS(t) (void)(t)
#endif

// What the path ray of a round found (TRT.c:793-889).
struct PathHit
{
    Hit ph;     // closest hit as the intersection routine produced it (un-nudged point)
    bool hit;   // a sphere or the ground
    bool sky;   // nothing: the sample ends on the sky
    d3 back;    // hit: unit vector back along the ray (nudge direction, TRT.c:871-872); sky: the unit direction (TRT.c:702, :878)
    d3 normal;  // hit: unit surface normal (TRT.c:878)
    int mat;    // hit: index into L.mat (sphere i, n = ground even, n + 1 = ground odd; TRT.c:850-851)
};

// P: closest hit of the path ray (o, d) of every lane with `alive`, candidates from the table of the ray's family `fam`
// (trt_raygrid.h) unless some lane's ray is not a member of its family; then the surface record.  `fam` becomes the family of
// the NEXT path ray: it starts on the sphere that was hit, or it is the mirror image in the ground of a ray of this one's
// family -- of the patch of its sphere this ray started on -- (a ray from the ground cannot hit the ground again; if it
// does, it has no family).
template <bool COUNT, bool REFRACT = false, bool PATCHES = false>
TRT_DEV PathHit path_stage(const LdsImage &L, const CullView &cull, const GridView &grids, int n, d3 o, d3 d, int &fam, bool alive, d3 gp, d3 gn,
                           Tally &tally, int inside = -1)
{
    TRT_STAGE_STAMPS(tally);
    bool p_list = false;
    unsigned long long p_cell = 0;
    if (grids.path_enabled)
    {
        bool fallback;
        if constexpr (PATCHES)
            p_cell = path_cell_patches(L, grids, n, fam, o, d, alive, fallback); // fam: now what a reflection by the ground belongs to
        else
            p_cell = path_cell(L, grids, n, fam, o, d, alive, fallback);
        p_list = !__any(fallback);
    }
    if (COUNT && !p_list)
        tally.swept++;
    PathHit r;
#if TRT_STAMP
    r.ph = trace<false, REFRACT>(L, cull, n, o, d, alive, gp, gn, tally.iters[0], tally.tests[0], nullptr, p_list, p_cell, grids.pool, grids.list_bits, inside, nullptr, stamp_sum,
                                 &stamp_prev, 2);
#else
    r.ph = trace<false, REFRACT, 2>(L, cull, n, o, d, alive, gp, gn, tally.iters[0], tally.tests[0], nullptr, p_list, p_cell, grids.pool, grids.list_bits, inside);
#endif
    r.hit = alive && r.ph.i >= 0;
    r.sky = alive && r.ph.i < 0;
    if (r.hit)
        fam = r.ph.i < n ? 2 + r.ph.i : (PATCHES ? fam : (fam == 0 ? 1 : (fam >= 2 && fam < 2 + n ? fam + n : -1)));
    // one unit() for "back along the ray" (nudge, TRT.c:871-872) or the sky direction (TRT.c:702), one for the normal
    r.back = unit(r.hit ? sub(o, r.ph.p) : d);
    r.normal = d;
    r.mat = 0;
    if (__any(r.hit))
    {
        d3 raw = gn;
        r.mat = n + checker_odd(r.ph.p); // TRT.c:850-851 (only meaningful for a ground hit)
        if (r.ph.i >= 0 && r.ph.i < n)
        {
            raw = sub(r.ph.p, load3(L.sph + 4 * r.ph.i)); // TRT.c:824
            r.mat = r.ph.i;
        }
        r.normal = unit(raw); // TRT.c:878
    }
    TRT_STAMP_AT(6); // P post: nudge direction, normal
    return r;
}

// S(i): lighting of TRT.c:894-957 for the lanes with `lit_lanes`: one shadow ray per light from the (nudged) surface point
// `o` with unit normal `normal`, candidates from the light's table (trt_lightgrid.h) unless some lane's origin is outside
// its range; the lit colour is accumulated in the reference's light order, NOT yet clamped (TRT.c:960).
template <bool COUNT>
TRT_DEV d3 shadow_stage(const LdsImage &L, const CullView &cull, const GridView &grids, int n, int nd, int nl, d3 o, d3 normal, int mat,
                        bool lit_lanes, d3 gp, d3 gn, Tally &tally)
{
    TRT_STAGE_STAMPS(tally);
    d3 lit = d3{0.0, 0.0, 0.0};
    if (!__any(lit_lanes))
        return lit;
    for (int li = 0; li < nl; li++)
    {
        if (COUNT && lit_lanes)
            tally.shadow++;
        d3 sd, lcolor;
        bool is_lit;
        double factor;
        if (li < nd)
        { // directional light, TRT.c:900-923
            TRT_MARK_AT(24); // ISA profile: a directional light's pass starts
            sd = load3(L.dir + li * kDirRecord);
            lcolor = load3(L.dir + li * kDirRecord + 3);
            bool use_list = false;
            unsigned long long cell = 0;
            if (grids.enabled)
            {
                const trt_dirgrid *G = L.dirgrid + li; // header in LDS: a global read here would sit in front of the cell's load
                int far;
                const int c = trt_dirgrid_cell(G, o.x, o.y, o.z, &far); // a light without a unit direction: far (stage_lds_image)
                if (lit_lanes && !far)
                    cell = grids.dir_lists[(size_t)li * grids.dir_stride + (unsigned)c];
                use_list = !__any(lit_lanes && (far || (unsigned)(cell >> 56) == TRT_LIST_NONE));
            }
            if (COUNT && !use_list)
                tally.swept++;
            TRT_STAMP_AT(8); // look-up
#if TRT_STAMP
            const Hit sh = trace<true>(L, cull, n, o, sd, lit_lanes, gp, gn, tally.iters[1], tally.tests[1], L.cull_dir + li * cull.padded, use_list, cell, grids.pool,
                                       grids.list_bits, -1, L.dir + li * kDirRecord + 6, stamp_sum, &stamp_prev, 9);
#else
            const Hit sh = trace<true, false, 9>(L, cull, n, o, sd, lit_lanes, gp, gn, tally.iters[1], tally.tests[1], L.cull_dir + li * cull.padded, use_list, cell, grids.pool, grids.list_bits,
                                                 -1, L.dir + li * kDirRecord + 6);
#endif
            is_lit = sh.i < 0;
            factor = min1(dot(normal, sd));
            TRT_STAMP_AT(13); // directional shadow tail
        }
        else
        { // point light, TRT.c:926-957
            TRT_MARK_AT(25); // ISA profile: a point light's pass starts
            const double *pl = L.pt + (li - nd) * 7;
            const d3 to_light = sub(load3(pl), o);
            const double light_d2 = dot(to_light, to_light);
            const double strength = clampd(pl[6] / light_d2, 0.0, 1.0);
            sd = unit(to_light);
            lcolor = load3(pl + 3);
            bool use_list = false;
            unsigned long long cell = 0;
            if (grids.enabled)
            {
                const trt_pointgrid *G = L.pointgrid + (li - nd);
                int far;
                const int c = trt_pointgrid_cell(G, o.x, o.y, o.z, &far);
                far |= !(__builtin_fabs(dot(sd, sd) - 1.0) <= 9.094947017729282e-13);
                if (lit_lanes && !far)
                    cell = grids.point_lists[(size_t)(li - nd) * grids.point_stride + (unsigned)c];
                use_list = !__any(lit_lanes && (far || (unsigned)(cell >> 56) == TRT_LIST_NONE));
            }
            if (COUNT && !use_list)
                tally.swept++;
            TRT_STAMP_AT(14); // unit(to_light), strength, look-up
            // ANY-HIT search (trt_lightgrid.h (4)): a candidate that is provably hit nearer than the light proves "dark", whichever
            // hit is the closest; no hit at all is "lit"; anything else (a blocker about as far as the light) leaves the lane
            // unsure and sends its wave through the closest-hit search of TRT.c:937-942 below.
            bool full = !use_list;
            is_lit = true;
            if (use_list)
            {
                const trt_pointgrid *G = L.pointgrid + (li - nd);
                double lo, hi;
                trt_point_shadow_bounds(G, light_d2, dot(sd, sd), &lo, &hi);
                bool dark, unsure;
                point_light_search(L, n, o, sd, lit_lanes, gp, gn, tally.iters[2], tally.tests[2], cell, grids.pool, grids.list_bits, lo, hi, dark, unsure);
                is_lit = !dark;
                full = __any(lit_lanes && unsure);
                TRT_STAMP_AT(17); // any-hit search
            }
            if (TRT_EXPECT_RARE(full))
            {
                if (COUNT)
                    tally.full++;
#if TRT_STAMP
                const Hit sh = trace<false>(L, cull, n, o, sd, lit_lanes, gp, gn, tally.iters[2], tally.tests[2], nullptr, use_list, cell, grids.pool, grids.list_bits, -1, nullptr, stamp_sum, &stamp_prev, 15);
#else
                const Hit sh = trace<false, false, 26>(L, cull, n, o, sd, lit_lanes, gp, gn, tally.iters[2], tally.tests[2], nullptr, use_list, cell, grids.pool, grids.list_bits);
#endif
                is_lit = sh.i < 0;
                // A blocker: is it farther than the light?  The reference compares light_d2 with the squared distance to the
                // blocker point NUDGED 1e-6 back along the ray (TRT.c:871-874, :939-942): (D - 1e-6)^2 up to ~1e-14 relative
                // rounding, where D^2 = sh.d2 (1 +- 4u).  From D <= (D^2+1)/2:
                //     (D-1e-6)^2 (1-1e-14)  >=  sh.d2 (1 - 1.01e-6) - 1.01e-6      and, for D >= 1e-5,   (D-1e-6)^2 (1+1e-14) < sh.d2.
                // Outside that band the answer is certain without normalising anything; inside it the exact nudged point is
                // formed as the reference does.
                const bool surely_lit = light_d2 < sh.d2 * (1.0 - 1.01e-6) - 1.01e-6;
                const bool surely_dark = sh.d2 > 1e-10 && light_d2 >= sh.d2;
                if (sh.i >= 0)
                    is_lit = surely_lit;
                if (__any(lit_lanes && sh.i >= 0 && !surely_lit && !surely_dark))
                {
                    const d3 to_blocker = sub(add(sh.p, scale(unit(sub(o, sh.p)), 0.000001)), o);
                    if (sh.i >= 0 && !surely_lit && !surely_dark)
                        is_lit = light_d2 < dot(to_blocker, to_blocker);
                }
            }
            factor = strength * min1(dot(normal, sd));
            TRT_STAMP_AT(19); // point shadow tail
        }
        if (lit_lanes && is_lit)
            lit = add(lit, mulc(scale(lcolor, factor), load3(L.mat + mat * 5)));
    }
    return lit;
}

// Register budget: asking the allocator for only 3 waves/SIMD lets it settle at 127 VGPRs -- which still runs 4 waves/SIMD
// (<= 128) -- with a better schedule than when it is forced under 128 (measured 3.02 vs 3.15 ms).  The build records the
// compiler's resource report in build/resource_usage.txt and `make lib` warns if this kernel ever needs more than 128.
#ifndef TRT_ROUNDS_WAVES
#if TRT_BLOCK > 768
#define TRT_ROUNDS_WAVES 4 // a 1024-thread workgroup is 4 waves per SIMD by itself
#else
#define TRT_ROUNDS_WAVES 3
#endif
#endif

// REFRACT (EXTENSION, parity unpinned: the reference has no refraction): f.ior[i] > 0 makes sphere i a refractor.  A path
// ray that hits it from outside is shaded like any hit and continues along the refracted direction from a point 1e-6 PAST
// the surface; inside, that sphere is met at its far root; leaving it adds no colour and no weight but costs a bounce; total
// internal reflection mirrors the ray.  Restated operation for operation in oracle/trt_oracle.c (shade_pixel_refractive).
// The default instantiation (REFRACT = false) is the reference's path, unchanged.
//
// COMPACT: the shading of a hit (the shadow rays of S and the clamp, TRT.c:894-962) is decoupled from the lane that owns the
// sample.  A hit becomes a TASK {nudged point, normal, material, weight} in a ring in LDS that belongs to the wave; the wave
// shades tasks 64 at a time, whichever lanes they came from, and a task may wait for the NEXT round's hits to fill the pass
// it runs in -- never longer, so that a lane has at most two colours outstanding and they reach its sample in bounce order
// (TRT.c:1040 adds them in that order; the sky colour of TRT.c:1044 after them).  What is computed for a task is what the
// owner would have computed, operation for operation; only WHICH lane computes it, and when, changes.
constexpr int kRingTasks = 128;                            // >= 63 carried over + 64 new
constexpr int kRingDoubles = 7 * kRingTasks + kRingTasks / 4 + 3 * 64; // point(3) normal(3) weight(1) as arrays of doubles, material as 16-bit ints;
                                                                        // then per LANE the direction of its next path ray while the wave shades

// The kernel's arguments live in the kernarg segment (constant address space, read with scalar loads).  Held in SGPRs for the
// whole kernel they do not fit: the compiler spilt ~75 of them to VGPR lanes and read ~110 back per round with v_readlane, a
// tenth of the VALU instructions.  TRT_FRESH_ARGS re-reads, at the head of a stage, the arguments that stage uses (the copies
// shadow the kernel's parameters; unused fields are never loaded): a few s_load per stage instead of the v_readlanes, and
// nothing to keep alive between stages.  The empty asm makes the pointer opaque so that the loads stay where they are written.
constexpr size_t kArgScene = 0;
constexpr size_t kArgCull = (kArgScene + sizeof(SceneView) + alignof(CullView) - 1) / alignof(CullView) * alignof(CullView);
constexpr size_t kArgFrame = (kArgCull + sizeof(CullView) + alignof(FrameView) - 1) / alignof(FrameView) * alignof(FrameView);
constexpr size_t kArgGrids = (kArgFrame + sizeof(FrameView) + alignof(GridView) - 1) / alignof(GridView) * alignof(GridView);
// the explicit arguments of a kernel sit in the kernarg segment like the members of a struct, from offset 0
struct RenderKernelArguments
{
    SceneView s;
    CullView cull;
    FrameView f;
    GridView grids;
};
static_assert(offsetof(RenderKernelArguments, cull) == kArgCull && offsetof(RenderKernelArguments, f) == kArgFrame &&
                  offsetof(RenderKernelArguments, grids) == kArgGrids,
              "TRT_FRESH_ARGS reads the kernel's arguments at these offsets");
// A pointer that was read from memory is "generic" to the compiler: loads and stores through it would be FLAT instructions.
// Every pointer in the argument structs is a device-memory address: say so (a round trip through the global address space,
// from which the compiler's address-space inference takes it).
template <class T>
TRT_DEV T *in_device_memory(T *p)
{
    T __attribute__((address_space(1))) *g = (T __attribute__((address_space(1))) *)p;
    asm("" : "+s"(g)); // keeps the pair of casts from being folded back into the generic pointer
    return (T *)g;
}
TRT_DEV SceneView in_device_memory(SceneView v)
{
    v.spheres = in_device_memory(v.spheres), v.dir_lights = in_device_memory(v.dir_lights), v.point_lights = in_device_memory(v.point_lights);
    v.sky = in_device_memory(v.sky);
    return v;
}
TRT_DEV CullView in_device_memory(CullView v)
{
    v.table = in_device_memory(v.table);
    return v;
}
TRT_DEV FrameView in_device_memory(FrameView v)
{
    v.jitter = in_device_memory(v.jitter), v.col_x = in_device_memory(v.col_x), v.row_y = in_device_memory(v.row_y);
    v.samples = in_device_memory(v.samples), v.out = in_device_memory(v.out), v.ior = in_device_memory(v.ior);
    v.counters = in_device_memory(v.counters), v.queue = in_device_memory(v.queue);
    return v;
}
TRT_DEV GridView in_device_memory(GridView v)
{
    v.dir = in_device_memory(v.dir), v.point = in_device_memory(v.point);
    v.dir_lists = in_device_memory(v.dir_lists), v.point_lists = in_device_memory(v.point_lists);
    v.path_lists = in_device_memory(v.path_lists), v.pool = in_device_memory(v.pool), v.sphere_fam = in_device_memory(v.sphere_fam);
    v.patch_rec = in_device_memory(v.patch_rec);
    return v;
}
#define TRT_FRESH_ARGS                                                                             \
    const char CONSTANT_AS *fresh_ = (const char CONSTANT_AS *)__builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(fresh_));                                                               \
    const SceneView s = in_device_memory(load_kernel_argument<SceneView>(fresh_ + kArgScene));     \
    const CullView cull = in_device_memory(load_kernel_argument<CullView>(fresh_ + kArgCull));     \
    const FrameView f = in_device_memory(load_kernel_argument<FrameView>(fresh_ + kArgFrame));     \
    const GridView grids = in_device_memory(load_kernel_argument<GridView>(fresh_ + kArgGrids));   \
    const int n = s.num_spheres, nd = s.num_dir, nl = s.num_dir + s.num_point;                     \
    const unsigned pixels_here = (unsigned)f.local_rows * (unsigned)f.width;                       \
    const unsigned total = pixels_here * (unsigned)f.spp;                                          \
    const d3 gp = load3(s.ground), gn = load3(s.ground + 3);                                       \
    (void)n, (void)nd, (void)nl, (void)total, (void)pixels_here, (void)gp, (void)gn, (void)cull, (void)grids

#ifndef TRT_COMPACT_BLOCK
#define TRT_COMPACT_BLOCK 1024 // one workgroup per CU: 16 rings and one image share the CU's 160 KB of LDS
#endif
constexpr int kCompactBlock = TRT_COMPACT_BLOCK;
constexpr int kBigBlock = 1024;

// PATCHES: the path rays' tables have a family per PATCH of a sphere's surface (GridView::patch_m > 0, trt_raygrid.h) instead of
// one per sphere: another look-up (path_cell_patches), the same everything else.  Its own instantiation, so that scenes
// without patches run the code -- and the register allocation -- they ran before there were any.
// BIG (round 5): the plain rounds in workgroups of 1024 threads -- ONE LDS image per CU, shared by sixteen waves -- for scenes whose
// image no longer fits four times into a CU's 160 KB (above ~290 spheres: 136 bytes a sphere): with 256-thread workgroups such a
// scene runs three, two, one wave per SIMD (512 spheres: two).  Same code; the register allocator has to stay under 128.
template <bool COUNT, bool REFRACT = false, bool COMPACT = false, bool PATCHES = false, bool BIG = false>
__global__ __launch_bounds__(BIG ? kBigBlock : COMPACT ? kCompactBlock : kPersistentBlock, ((COMPACT && !COUNT) || BIG) ? 4 : TRT_ROUNDS_WAVES) void render_rounds_kernel(SceneView s, CullView cull, FrameView f, GridView grids)
{
    static_assert(!(REFRACT && COMPACT), "the refraction extension runs on the plain rounds");
    static_assert(!(PATCHES && COMPACT), "scenes with patches run the plain rounds");
    static_assert(!BIG || (PATCHES && !COUNT && !REFRACT && !COMPACT), "1024-thread workgroups: the shipping patch instantiation only");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const LdsImage L = stage_lds_image(lds, s, cull, f, grids);
    const int n = s.num_spheres, nd = s.num_dir, nl = s.num_dir + s.num_point;
    const int lane = threadIdx.x & 63;
    const unsigned pixels_here = (unsigned)f.local_rows * (unsigned)f.width;
    const unsigned total = pixels_here * (unsigned)f.spp; // work units, < 2^31
    (void)total; // TRT_FRESH_ARGS re-derives it where it is used
    const d3 gp = load3(s.ground), gn = load3(s.ground + 3);

    // ---- per-lane state ------------------------------------------------------------------------------
    bool alive = true;             // owns a unit (a sample being traced)
    bool want_unit = true;         // needs a (new) unit before the next round
    unsigned slot_id = 0;          // k*pixels + pixel: where the sample's colour goes in f.samples
    d3 sample = d3{0.0, 0.0, 0.0}; // pixel_color of the sample (TRT.c:1012)
    double weight = 1.0, weight_sum = 0.0;
    int bounces = 0;
    d3 o = d3{0.0, 0.0, 0.0}, d = d3{0.0, 0.0, -1.0}; // the pending path ray
    d3 next_dir = d;                                  // un-normalised direction of the next path ray
    int fam = 0;                                      // family of the pending path ray (trt_raygrid.h): 0 = it starts at the eye
    int inside = -1;                                  // REFRACT: the refractor the pending ray travels inside of
    Tally tally;
    unsigned pool_next = 0, pool_end = 0; // wave-uniform: units fetched from the queue, not yet handed out
    { // the queue (trt_common.hpp, kQueueStride): the first chunk is the wave's own -- 4096 waves asking for their first units at the
      // same moment would wait in line for them
        const unsigned group = blockIdx.x, waves = blockDim.x >> 6, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const unsigned j = (group >> f.queue_shift) * waves + wave;
        pool_next = ((j << f.queue_shift) + (group & ((1u << f.queue_shift) - 1u))) * f.chunk;
        pool_end = pool_next + f.chunk;
    }
    // COMPACT: the wave's ring of shading tasks and what this lane still expects from it
    double *const ring = lds + f.ring_at + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * kRingDoubles; // wave-uniform: a scalar
    unsigned short *const ring_mat = (unsigned short *)(ring + 7 * kRingTasks); // < 2^16 materials: the LDS image ends long before
    double *const parked = ring + 7 * kRingTasks + kRingTasks / 4;
    if (COMPACT)
    {
        for (int i = lane; i < kRingDoubles; i += 64)
            ring[i] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    unsigned q_head = 0, q_tail = 0, q_old = 0; // wave-uniform counters: [q_head, q_tail) waits, tasks below q_old are of the previous round
    unsigned my_task = 0, old_slot = 0;         // this round's task of the lane (its number), the previous round's (its place in the ring)
    bool my_open = false, old_open = false;     // ... whose colour has not arrived yet
    bool waiting = false;                       // the sample ended on a hit whose colour arrives in the next round: the lane sits that round out

    TRT_STAGE_STAMPS(tally);
#if TRT_STAMP == 2
    stamp_prev = 0; // the instruction count starts at 0 at the kernel's entry (tools/archive/count_isa.py); the prologue goes to the first slot
#elif TRT_STAMP
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    for (;;)
    {
        TRT_STAMP_AT(22); // loop edge
        // =============== hand out work units; primary rays of new samples (TRT.c:981-1016) ===============
        {
            TRT_FRESH_ARGS;
            const unsigned long long need = __ballot(want_unit);
            if (need)
            {
                const unsigned wanted = (unsigned)__builtin_popcountll(need);
                const unsigned rank = lanes_below(need);
                unsigned mine = pool_next + rank;
                if (pool_end - pool_next < wanted)
                { // refill the wave's pool with ONE atomic; what is left of the old chunk is served first
                    const unsigned left = pool_end - pool_next;
                    unsigned fresh = 0;
                    const int leader = __builtin_ctzll(need);
                    const unsigned word = blockIdx.x & ((1u << f.queue_shift) - 1u); // the workgroup's word of the queue: its XCD's
                    if (lane == leader)
                        fresh = atomicAdd(f.queue + word * kQueueStride, 1u);
                    fresh = (unsigned)__builtin_amdgcn_readlane((int)fresh, leader); // the leader is wave-uniform: no cross-lane permute, no lane id
                    fresh = ((fresh << f.queue_shift) + word) * f.chunk; // a chunk of this launch or one of the first few behind its end: < 2^32
                    if (rank >= left)
                        mine = fresh + (rank - left);
                    pool_next = fresh + (wanted - left);
                    pool_end = fresh + f.chunk;
                }
                else
                    pool_next += wanted;
                if (want_unit)
                {
                    alive = mine < total;
                    if (alive)
                    {
                        // unit -> (pixel, k) -> (row, column) by multiply-high with min(ceil(2^32/x), 2^32-1): off by at most one
                        // either way; the correction is two selects each, not branches (three nested exec regions per lane before)
                        unsigned pixel = __umulhi(mine, f.spp_magic);
                        int k = (int)(mine - pixel * (unsigned)f.spp);
                        {
                            const int under = k < 0, over = k >= f.spp;
                            pixel += (unsigned)(over - under);
                            k += (under - over) * f.spp;
                        }
                        slot_id = (unsigned)k * pixels_here + pixel; // sample-major scratch: the reduction streams it
                        unsigned row = __umulhi(pixel, f.width_magic);
                        int col = (int)(pixel - row * (unsigned)f.width);
                        {
                            const int under = col < 0, over = col >= f.width;
                            row += (unsigned)(over - under);
                            col += (under - over) * f.width;
                        }
                        const double sx = f.col_x[col] + L.jit[k];
                        // rows dealt from tile 0 with step 1 (a whole frame, a one-rank shard): the local row IS the frame row
                        const double sy = f.row_y[f.tile_first == 0 && f.tile_step == 1 ? (int)row : frame_row_of_magic(f, row)] + L.jit[f.spp + k];
                        d3 dir = d3{0.0, 0.0, 0.0};
                        dir = add(dir, scale(load3(L.cam + 0), sx));
                        dir = add(dir, scale(load3(L.cam + 3), sy));
#if TRT_OPT_BZSZ
                        dir = add(dir, load3(L.cam + 13)); // scale(basis z, sz), formed once per workgroup (stage_lds_image)
#else
                        dir = add(dir, scale(load3(L.cam + 6), L.cam[12]));
#endif
                        next_dir = sub(dir, load3(L.cam + 9)); // sic, TRT.c:1005
                        o = load3(L.cam + 9);
                        fam = 0;
                        inside = -1;
                        sample = d3{0.0, 0.0, 0.0};
                        weight = 1.0;
                        weight_sum = 0.0;
                        bounces = 0;
                    }
                    want_unit = false;
                }
            }
        }
        if (!__any(alive || (COMPACT && waiting)))
            break;
        if (COUNT)
            tally.rounds++;
        TRT_STAMP_AT(0); // units + primary rays
        d = unit(next_dir); // TRT.c:1008 for a primary ray, TRT.c:1055 for a reflected one

        // ======================================= P: the path ray =======================================
        if (COUNT && alive)
            tally.path++;
        TRT_STAMP_AT(1); // unit(next_dir)
        PathHit hit;
        {
            TRT_FRESH_ARGS;
            hit = path_stage<COUNT, REFRACT, PATCHES>(L, cull, grids, n, o, d, fam, alive, gp, gn, tally, inside);
        }
        if constexpr (COMPACT)
        {
            const bool path_hit = hit.hit, path_sky = hit.sky;
            bool end_sample = path_sky;
            uint32_t sky_t = 0;
            const double weight_before = weight;
            {
            TRT_FRESH_ARGS;
            if (__any(path_sky))
                sky_t = sky_texel_wave(s.sky, s.sky_dim, hit.back, s.sky_dim_f, path_sky); // TRT.c:858-867; added to the sample after the colours still on their way
            // ---- a hit becomes a task; what does not depend on its colour happens now (TRT.c:1036-1038, :1054) ----
            const unsigned long long hits = __ballot(path_hit);
            if (hits)
            {
                if (path_hit)
                {
                    my_task = q_tail + lanes_below(hits);
                    my_open = true;
                    const unsigned at = my_task & (kRingTasks - 1);
                    const d3 so = add(hit.ph.p, scale(hit.back, 0.000001)); // TRT.c:873-874
                    ring[0 * kRingTasks + at] = so.x, ring[1 * kRingTasks + at] = so.y, ring[2 * kRingTasks + at] = so.z;
                    ring[3 * kRingTasks + at] = hit.normal.x, ring[4 * kRingTasks + at] = hit.normal.y, ring[5 * kRingTasks + at] = hit.normal.z;
                    ring[6 * kRingTasks + at] = weight;
                    ring_mat[at] = (unsigned short)hit.mat;
                    const d3 nd3 = reflect(d, hit.normal); // TRT.c:1054; kept in LDS while the wave shades
                    parked[lane] = nd3.x, parked[64 + lane] = nd3.y, parked[128 + lane] = nd3.z;
                    weight *= L.mat[hit.mat * 5 + 3];
                    bounces++;
                    end_sample = !(bounces < f.bounce_limit && weight > 0.00001); // TRT.c:1018
                }
                q_tail += (unsigned)__builtin_popcountll(hits);
            }
            }
            if (alive)
                weight_sum += weight_before; // TRT.c:1034 (a sample that ends is normalised by this sum, one that goes on carries it)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // the ring is written by some lanes and read by others
            TRT_STAMP_AT(7); // P post: sky texel, reflection, nudge, enqueue
            // ---- S over tasks: ONE pass of up to 64 tasks from the head of the ring, if a task of the previous round waits (they
            // are at the head, fewer than 64) or 64 wait.  At most 63 stay behind, so 63 + 64 is the most the ring ever holds. ----
            if (q_head != q_tail && ((int)(q_old - q_head) > 0 || q_tail - q_head >= 64u))
            {
                TRT_FRESH_ARGS;
                const unsigned take = q_tail - q_head < 64u ? q_tail - q_head : 64u;
                const bool has = (unsigned)lane < take;
                const unsigned at = (q_head + (unsigned)lane) & (kRingTasks - 1);
                // lanes beyond `take` read whatever their slot holds -- an old task or the zeros the ring started with -- and
                // take no part in the stage (lit_lanes = has); nothing they compute is stored
                const d3 so = d3{ring[0 * kRingTasks + at], ring[1 * kRingTasks + at], ring[2 * kRingTasks + at]};
                const d3 sn = d3{ring[3 * kRingTasks + at], ring[4 * kRingTasks + at], ring[5 * kRingTasks + at]};
                const int sm = ring_mat[at];
                if (COUNT)
                    tally.passes++;
                const d3 lit = shadow_stage<COUNT>(L, cull, grids, n, nd, nl, so, sn, sm, has, gp, gn, tally);
                d3 color = d3{clampd(lit.x, 0.0, 1.0), clampd(lit.y, 0.0, 1.0), clampd(lit.z, 0.0, 1.0)}; // TRT.c:960-962
                color = scale(color, ring[6 * kRingTasks + at]);                                          // TRT.c:1035
                if (has) // the colour takes the place of the normal in the task's record; its owner collects it at the END of a round
                    ring[3 * kRingTasks + at] = color.x, ring[4 * kRingTasks + at] = color.y, ring[5 * kRingTasks + at] = color.z;
                q_head += take;
            }
            q_old = q_tail;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            TRT_STAMP_AT(20); // lit accumulate
            // ---- colours that have arrived, in bounce order (TRT.c:1040): last round's task is shaded by now, this round's may be ----
            if (old_open)
            {
                const unsigned at = old_slot;
                sample = add(sample, d3{ring[3 * kRingTasks + at], ring[4 * kRingTasks + at], ring[5 * kRingTasks + at]});
            }
            if (my_open && (int)(q_head - my_task) > 0)
            {
                const unsigned at = my_task & (kRingTasks - 1);
                sample = add(sample, d3{ring[3 * kRingTasks + at], ring[4 * kRingTasks + at], ring[5 * kRingTasks + at]});
                my_open = false;
            }
            // ======================================= END of the bounce =======================================
            if (path_hit)
            { // the next path ray: from the nudged point, along the reflection
                o = d3{ring[0 * kRingTasks + (my_task & (kRingTasks - 1))], ring[1 * kRingTasks + (my_task & (kRingTasks - 1))],
                       ring[2 * kRingTasks + (my_task & (kRingTasks - 1))]};
                next_dir = d3{parked[lane], parked[64 + lane], parked[128 + lane]};
            }
            // A sample that ended on this round's hit while the hit's colour is still on its way waits for it: the lane sits
            // the next round out (its task is then of the previous round: shaded whatever else happens) and stores the sample at
            // that round's END.
            bool finish = end_sample;
            if (waiting)
                finish = true, waiting = false; // the colour it waited for was added above
            else if (end_sample && my_open)
                finish = false, waiting = true, alive = false;
            old_slot = my_task & (kRingTasks - 1), old_open = my_open, my_open = false;
            if (path_sky)
            { // TRT.c:1044-1048: colour = texel, the sample ends here
                const d3 color = d3{L.b255[sky_t & 0xFF], L.b255[(sky_t >> 8) & 0xFF], L.b255[(sky_t >> 16) & 0xFF]};
                sample = add(sample, scale(color, weight));
            }
            if (__any(finish))
            { // TRT.c:1061
                const double q = 1.0 / weight_sum;
                if (finish)
                {
                    TRT_FRESH_ARGS;
                    unsigned slot = slot_id;
#if TRT_AB_DUMMY_STORES // diagnostic build (profiles/r03: what the sample scratch costs): every store lands in 48 KB
                    slot &= 2047u;
#endif
                    asm volatile("" : "+v"(slot)); // the address is formed here, not kept as 64 bits for the life of the sample
                    double *out = f.samples + (size_t)slot * 3;
                    out[0] = sample.x * q;
                    out[1] = sample.y * q;
                    out[2] = sample.z * q;
                    want_unit = true;
                }
            }
        }
        else
        {
        // REFRACT: the ray leaves the refractor it was inside of -- no shading, no weight, one bounce
        const bool leaving = REFRACT && hit.hit && hit.ph.i == inside;
        const bool path_hit = hit.hit && !leaving, path_sky = hit.sky;
        const d3 h_normal = hit.normal;
        const int h_mat = hit.mat;
        bool end_sample = false;
        double weight_sum_new = weight_sum + weight; // TRT.c:1034
        uint32_t sky_t = 0;
        if (__any(path_sky))
        { // TRT.c:858-867, :1044-1048: colour = texel, the sample ends here.  The texel is only LOADED here: it is a dependent read
          // from global memory, and what uses it (the sample's colour) is not needed before the END of the round, behind the other
          // lanes' shadow stages
            TRT_FRESH_ARGS;
            sky_t = sky_texel_wave(s.sky, s.sky_dim, hit.back, s.sky_dim_f, path_sky);
            end_sample = path_sky;
        }
        d3 lit;
        if (!REFRACT)
        {
            if (path_hit)
            {
                next_dir = reflect(d, h_normal);              // TRT.c:1054, normalised at the top of the next round
                o = add(hit.ph.p, scale(hit.back, 0.000001)); // TRT.c:873-874; origin of the shadow rays and of the next path ray
            }
            TRT_STAMP_AT(7); // P post: sky texel, reflection, nudge
            // ===================================== S(i): shadow rays =====================================
            if (COUNT && __any(path_hit))
                tally.passes++;
            {
                TRT_FRESH_ARGS;
                lit = shadow_stage<COUNT>(L, cull, grids, n, nd, nl, o, h_normal, h_mat, path_hit, gp, gn, tally);
            }
        }
        else
        {
            const d3 shade_at = add(hit.ph.p, scale(hit.back, 0.000001)); // TRT.c:873-874: where the surface is lit; origin of a reflected ray
            if (hit.hit)
            {
                bool bent = false;
                if (hit.ph.i < n)
                {
                    const double ior = f.ior[hit.ph.i];
                    if (ior > 0.0)
                    {
                        const d3 nn = leaving ? scale(h_normal, -1.0) : h_normal; // the normal facing the incoming ray
                        const double cosi = -dot(nn, d);
                        const double eta = leaving ? ior : 1.0 / ior;
                        const double kk = 1.0 - (eta * eta) * (1.0 - cosi * cosi);
                        bent = true;
                        if (!(kk < 0.0))
                        {
                            const double fac = eta * cosi - __builtin_sqrt(kk);
                            next_dir = d3{eta * d.x + fac * nn.x, eta * d.y + fac * nn.y, eta * d.z + fac * nn.z};
                            o = sub(hit.ph.p, scale(hit.back, 0.000001)); // 1e-6 past the surface
                            inside = leaving ? -1 : hit.ph.i;
                        }
                        else
                        { // total internal reflection: mirror about the facing normal, stay on this side
                            next_dir = reflect(d, nn);
                            o = shade_at;
                        }
                    }
                }
                if (!bent)
                {
                    next_dir = reflect(d, h_normal);
                    o = shade_at;
                }
            }
            lit = shadow_stage<COUNT>(L, cull, grids, n, nd, nl, shade_at, h_normal, h_mat, path_hit, gp, gn, tally);
        }

        TRT_STAMP_AT(20); // lit accumulate
        // ======================================= END of the bounce =======================================
        if (path_hit)
        { // TRT.c:960-962 then :1034-1048
            TRT_FRESH_ARGS;
            d3 color = d3{clampd(lit.x, 0.0, 1.0), clampd(lit.y, 0.0, 1.0), clampd(lit.z, 0.0, 1.0)};
            color = scale(color, weight);
            weight *= L.mat[h_mat * 5 + 3];
            bounces++;
            sample = add(sample, color);
            if (bounces < f.bounce_limit && weight > 0.00001) // TRT.c:1018
                weight_sum = weight_sum_new;
            else
                end_sample = true;
        }
        if (REFRACT && leaving)
        {
            bounces++;
            weight_sum_new = weight_sum; // nothing was contributed
            if (!(bounces < f.bounce_limit))
                end_sample = true;
        }
        if (path_sky)
        { // TRT.c:866, :1035-1051: colour = texel / 255, scaled by the sample's weight
            const d3 color = d3{L.b255[sky_t & 0xFF], L.b255[(sky_t >> 8) & 0xFF], L.b255[(sky_t >> 16) & 0xFF]};
            sample = add(sample, scale(color, weight));
        }
        if (__any(end_sample))
        { // TRT.c:1061: the sample's colour, normalised by the weights it gathered
            const double q = 1.0 / weight_sum_new;
            if (end_sample)
            {
                TRT_FRESH_ARGS;
#if TRT_AB_DUMMY_STORES
                double *out = f.samples + (size_t)(slot_id & 2047u) * 3;
#else
                double *out = f.samples + (size_t)slot_id * 3;
#endif
                out[0] = sample.x * q; // plain stores: non-temporal ones (keeping the 498 MB stream out of L2) measured no different
                out[1] = sample.y * q;
                out[2] = sample.z * q;
                want_unit = true;
            }
        }
        }
        TRT_STAMP_AT(21); // END of the bounce
    }

#if defined(TRT_MARKS) && TRT_MARKS == 2
    // ISA profile (tools/isa_profile.py): lane `slot` of v240 + kind holds the wave's executed instructions of that kind in
    // the intervals that START at stage boundary `slot` (63: before the first boundary)
    if (!COUNT && f.counters)
    {
#define TRT_PROFILE_DUMP(kind)                                                         \
    {                                                                                  \
        unsigned x_;                                                                   \
        asm volatile("v_mov_b32 %0, v" #kind : "=v"(x_));                              \
        atomicAdd(&f.counters[kProfileAt + 64 * ((kind) - 240) + lane], (unsigned long long)x_); \
    }
        TRT_PROFILE_DUMP(240) TRT_PROFILE_DUMP(241) TRT_PROFILE_DUMP(242) TRT_PROFILE_DUMP(243) TRT_PROFILE_DUMP(244) TRT_PROFILE_DUMP(245)
        TRT_PROFILE_DUMP(246) TRT_PROFILE_DUMP(247) TRT_PROFILE_DUMP(248) TRT_PROFILE_DUMP(249) TRT_PROFILE_DUMP(250)
#undef TRT_PROFILE_DUMP
    }
#endif
    if (COUNT && f.counters)
    {
        atomicAdd(&f.counters[0], (unsigned long long)tally.path);
        atomicAdd(&f.counters[1], (unsigned long long)tally.shadow);
        for (int k = 0; k < 3; k++)
            atomicAdd(&f.counters[33 + k], (unsigned long long)tally.tests[k]); // exact tests per lane: tests / (64 iterations) = lane activity
        if (lane == 0)
        {
            atomicAdd(&f.counters[2], (unsigned long long)tally.rounds);
            atomicAdd(&f.counters[3], (unsigned long long)tally.iters[0] + tally.iters[1] + tally.iters[2]);
            atomicAdd(&f.counters[28], (unsigned long long)tally.swept); // traces in which the wave fell back to the sweep
            atomicAdd(&f.counters[29], (unsigned long long)tally.passes); // times the wave ran the shadow stage
            for (int k = 0; k < 3; k++)
                atomicAdd(&f.counters[30 + k], (unsigned long long)tally.iters[k]); // wave-level iterations of the three exact-test loops
            atomicAdd(&f.counters[36], (unsigned long long)tally.full);            // point-light searches that fell back to the closest hit
#if TRT_STAMP
            for (int i = 0; i < 24; i++)
                atomicAdd(&f.counters[4 + i], stamp_sum[i]);
#endif
        }
    }
}

// trt_probe_rays through the PRODUCTION stages: one lane per ray, the very path_stage / shadow_stage of the render kernel
// (tables, fall-back sweep, exact tests, lighting).  families[i]: the family ray i is looked up in (trt_raygrid.h; < 0: none,
// the wave sweeps); nullptr: none for every ray.  Outputs as trace_ray / apply_lighting produce them (TRT.c:793-963).
template <bool PATCHES>
__global__ __launch_bounds__(kPersistentBlock) void probe_rounds_kernel(SceneView s, CullView cull, FrameView f, GridView grids, const double *rays,
                                                                        const int *families, long count, int *obj, double *point, double *normal,
                                                                        double *material, double *lit_out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const LdsImage L = stage_lds_image(lds, s, cull, f, grids);
    const int n = s.num_spheres, nd = s.num_dir, nl = s.num_dir + s.num_point;
    const d3 gp = load3(s.ground), gn = load3(s.ground + 3);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool alive = i < count;
    const long at = alive ? i : 0;
    const d3 o = load3(rays + 6 * at), d = load3(rays + 6 * at + 3);
    int fam = families ? families[at] : -1;
    Tally tally;
    const PathHit hit = path_stage<false, false, PATCHES>(L, cull, grids, n, o, d, fam, alive, gp, gn, tally);
    const d3 surface = hit.hit ? add(hit.ph.p, scale(hit.back, 0.000001)) : o; // TRT.c:873-874 / :860
    const d3 lit = shadow_stage<false>(L, cull, grids, n, nd, nl, surface, hit.normal, hit.mat, hit.hit, gp, gn, tally);
    const uint32_t sky_t = sky_texel_wave(s.sky, s.sky_dim, hit.back, s.sky_dim_f, alive && !hit.hit); // the render kernels' look-up
    if (!alive)
        return;
    d3 color = d3{0.0, 0.0, 0.0};
    double refl = 0.0, spec = 0.0; // TRT.c:866: the compound literal zero-fills
    d3 nrm = hit.back;             // a miss: the (normalised) ray direction, TRT.c:861, :878
    if (hit.hit)
    {
        color = load3(L.mat + hit.mat * 5);
        refl = L.mat[hit.mat * 5 + 3];
        spec = L.mat[hit.mat * 5 + 4];
        nrm = hit.normal;
    }
    else
    {
        const uint32_t t = sky_t;
        color = d3{L.b255[t & 0xFF], L.b255[(t >> 8) & 0xFF], L.b255[(t >> 16) & 0xFF]};
    }
    obj[i] = hit.hit ? (hit.ph.i < n ? 1 : 2) : 0;
    point[3 * i + 0] = surface.x, point[3 * i + 1] = surface.y, point[3 * i + 2] = surface.z;
    normal[3 * i + 0] = nrm.x, normal[3 * i + 1] = nrm.y, normal[3 * i + 2] = nrm.z;
    material[5 * i + 0] = color.x, material[5 * i + 1] = color.y, material[5 * i + 2] = color.z;
    material[5 * i + 3] = refl, material[5 * i + 4] = spec;
    const d3 c = hit.hit ? d3{clampd(lit.x, 0.0, 1.0), clampd(lit.y, 0.0, 1.0), clampd(lit.z, 0.0, 1.0)} : d3{0.0, 0.0, 0.0}; // TRT.c:960
    lit_out[3 * i + 0] = c.x, lit_out[3 * i + 1] = c.y, lit_out[3 * i + 2] = c.z;
}

} // namespace trt
