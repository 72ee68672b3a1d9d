// trt_persistent.hpp -- the EARLIER production design (kernel id 2): persistent waves with a per-lane state machine.
// Superseded by the mode-synchronous rounds of trt_rounds.hpp (kernel id 0), kept as an independent implementation for
// parity tests and A/B measurements; it also hosts what both share: LDS sizing, the work-queue constants, the ordered
// reduction kernel and the TRT_DUP / TRT_STAMP cost-attribution switches.
//
// Shape of the work: per pixel `spp` samples, per sample a bounce loop, per bounce
// one closest-hit trace plus one shadow trace per light; a trace = N sphere tests + 1 plane test.
// Path length varies 1..bounce_limit per sample, so neighbouring pixels differ ~10x in cost.
//
// Design:
//   * PERSISTENT waves, ONE LANE = ONE PIXEL AT A TIME.  A lane runs its pixel's samples in
//     order (the mean over samples must be accumulated in the reference's order, TRT.c:1063)
//     and, when the pixel is finished, pulls the next pixel index from a global queue
//     (wave-aggregated atomic).  Lanes never idle while pixels remain.
//   * ONE TRACE LOOP FOR EVERY KIND OF RAY.  Path rays (TRT.c:1024) and the shadow rays of
//     lighting (TRT.c:907, :937) are the same closest-hit search, so each lane keeps a small
//     state machine (mode = PATH | SHADOW(light i)) and all 64 lanes of a wave -- whatever
//     their mode -- go through the sphere sweep together.  The reference's recursion
//     (apply_lighting -> trace_ray) becomes "re-enter the loop with the shadow ray".
//   * TWO-PHASE SPHERE SWEEP.  Phase 1: wave-uniform loop over a FP32 culling table read with
//     scalar loads (the sphere index is uniform, so the table rides in SGPRs): 7 FP32 VALU ops
//     + 2 compares per sphere, result = per-lane candidate bit mask (trt_filter.h;
//     conservative, never decides a hit).  Phase 2: each lane pops its own candidates in
//     ascending index order and runs the EXACT FP64 test in the reference's operation order
//     against sphere records in LDS (per-lane index -> LDS gather).
//   * ONE SHARED NORMALISATION STAGE PER ITERATION.  FP64 sqrt and division are ~18 and ~11
//     instructions each and the reference normalises vectors everywhere (3 divisions each).
//     A lane's post-trace work is therefore split into POST (cheap, per mode: decide what
//     happens next, emit up to two vectors to normalise and one quotient), NORM (all lanes
//     together: two unit() slots, one division) and FINISH (cheap, per mode: consume them).
//     Divergent per-mode code holds only adds/multiplies; the long sequences run convergent.
//   * Scene records are staged once per workgroup into LDS (spheres SoA, materials, lights,
//     a byte/255.0 table); the cubemap (1.5 MB at 256^2) does not fit the 160 KB LDS and stays
//     in global memory / L2, touched once per sample.
//   * FP64 throughout, contraction off: results are bit-identical to the reference.
#pragma once

#include "trt_device.hpp"
#include "trt_filter.h"

namespace trt
{

constexpr int kPersistentBlock = 256;
// TRT_DUP: cost-attribution diagnostic builds.  One stage is executed TWICE (the second result is merged so
// that nothing can be optimised away, and it is identical, so the frame stays correct); the slowdown against
// the normal build is that stage's cost.  1 = filter set-up, 2 = phase-1 sweep, 4 = plane test, 8 = NORM stage, 16 = phase-2 exact tests
// (idempotent: strict '<' keeps the first result).
#ifndef TRT_DUP
#define TRT_DUP 0
#endif
// TRT_STAMP=1: diagnostic build with s_memtime stamps between the stages of the main loop; per-stage wave-cycle
// sums go to counters[4..11] (read SHARES from it, never its run time: the stamps fence the schedule).
#ifndef TRT_STAMP
#define TRT_STAMP 0
#endif
#if TRT_STAMP
#define TRT_STAMP_AT(slot)                                                                   \
    do                                                                                       \
    {                                                                                        \
        unsigned long long now_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        stamp_sum[slot] += now_ - stamp_prev;                                                \
        stamp_prev = now_;                                                                   \
    } while (0)
#else
#define TRT_STAMP_AT(slot) \
    do                     \
    {                      \
    } while (0)
#endif
// TRT_SWEEP_LDS: where phase 1 reads the culling table from.  1 (default) = from LDS, one broadcast ds_read_b128
// per sphere, so that every VALU operand is a VGPR; 0 = scalar loads, coefficients as SGPR operands.  On gfx950 an
// SGPR source roughly doubles the issue cost of v_fmac/v_mul/v_fma (tools/ubench_valu.hip), 25.7 vs ~16 cycles per test.
#ifndef TRT_SWEEP_LDS
#define TRT_SWEEP_LDS 1
#endif
#ifndef TRT_PERSISTENT_WAVES
#define TRT_PERSISTENT_WAVES 4 // min waves per SIMD the register allocator must leave room for (= 256-thread blocks per CU)
#endif
constexpr unsigned kQueueChunkPixels = 64;   // work units fetched from the global queue per atomic (one wave's worth)
#ifndef TRT_QUEUE_CHUNK
#define TRT_QUEUE_CHUNK 256
#endif
constexpr unsigned kQueueChunkSamples = TRT_QUEUE_CHUNK;
static_assert(kQueueChunkSamples >= 64, "a chunk must hold the 64 units the lanes of a wave can ask for in one round");
#ifndef TRT_CULL_GROUP
#define TRT_CULL_GROUP 8
#endif
constexpr int kCullGroup = TRT_CULL_GROUP; // culling-table entries fetched per scalar-load batch (table padded to this)

struct PersistentLaunch
{
    unsigned grid, block;
};

inline PersistentLaunch persistent_launch_shape(int compute_units, int blocks_per_cu, long pixels)
{
    long want = (pixels + kPersistentBlock - 1) / kPersistentBlock;
    long cap = (long)compute_units * (blocks_per_cu > 0 ? blocks_per_cu : 1);
    return PersistentLaunch{(unsigned)(want < cap ? (want > 0 ? want : 1) : cap), (unsigned)kPersistentBlock};
}

// LDS image: cx[n] cy[n] cz[n] r2[n] | mat[(n+2)*5] (spheres, ground even, ground odd) |
//            dir lights: unit to-light(3) colour(3) | point lights: pos(3) colour(3) intensity | byte/255.0 [256] |
//            camera: basis x,y,z (9) eye (3) -screen_distance (1) | jitter x[spp] y[spp]
constexpr int kLdsCameraDoubles = 13;
// TRT_SWEEP_MFMA == 2 (rounds kernel, experiment): per wave, the rays' filter vectors (64 x 9 floats) and the exchange
// buffer of the verdict words (64 lanes x 4 x 8 bytes) behind the LDS image
#ifndef TRT_SWEEP_MFMA
#define TRT_SWEEP_MFMA 0
#endif
constexpr int kMfma16RayFloats = 9, kMfma16WaveFloats = 64 * kMfma16RayFloats + 64 * 8;
constexpr int kDirGridDoubles = 10, kPointGridDoubles = 6; // sizeof(trt_dirgrid) / 8, sizeof(trt_pointgrid) / 8 (asserted in trt_rounds.hpp)
inline size_t persistent_lds_bytes(const SceneView &s, int spp)
{
    const size_t padded = ((size_t)s.num_spheres + kCullGroup - 1) / kCullGroup * kCullGroup;
    return sizeof(double) * ((size_t)s.num_spheres * 4 + ((size_t)s.num_spheres + 2) * 5 + (size_t)s.num_dir * 6 +
                             (size_t)s.num_point * 7 + 256 + kLdsCameraDoubles + 2 * (size_t)spp + 1 /* 16-B alignment */ +
                             (TRT_SWEEP_LDS ? padded * 2 : 0) /* culling table first: 4 floats per sphere, 16-B aligned */ +
                             (size_t)s.num_dir * padded * 2 /* one fixed-direction table per directional light (rounds kernel) */ +
                             (2 + (size_t)s.num_dir) * (((size_t)s.num_spheres + 63) / 64 * 64) /* MFMA A-operand images, 2*padded64 floats each (rounds kernel) */ +
                             (size_t)s.num_dir * kDirGridDoubles + (size_t)s.num_point * kPointGridDoubles /* light-table headers (rounds kernel) */ +
                             (TRT_SWEEP_MFMA == 2 ? (size_t)(kPersistentBlock / 64) * kMfma16WaveFloats / 2 : 0));
}

struct CullView
{
    const float *table; // padded to a multiple of kCullGroup entries of {Cx,Cy,Cz,kk}
    int padded;
    double c0x, c0y, c0z;
    float cn, rm;
};

typedef const float __attribute__((address_space(4))) *const_float_ptr;

// modes of a lane between two traces
enum : int
{
    kModeBoot = -1, // has no pixel yet
    kModePath = 0   // 1 + i: shadow ray of light i (directional lights first, then point lights)
};

// SAMPLE_UNITS = false: a work unit is a PIXEL; the lane keeps the running mean and writes the pixel itself.
// SAMPLE_UNITS = true : a work unit is ONE SAMPLE of a pixel (unit = pixel*spp + k); the lane writes the sample's
//   normalised colour (TRT.c:1061) to f.samples[unit] and reduce_samples_kernel forms the mean in the reference's
//   order (TRT.c:1063-1065).  Ten times finer balancing: chosen by the host when there are few pixels per lane
//   (small frames, or one frame sharded over several GPUs).
template <bool COUNT, bool SAMPLE_UNITS>
__global__ __launch_bounds__(kPersistentBlock, TRT_PERSISTENT_WAVES) void render_persistent_kernel(SceneView s, CullView cull, FrameView f)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = s.num_spheres, nd = s.num_dir, np = s.num_point, nl = nd + np;
    float4 *const l_cull = (float4 *)lds; // culling table image, read with broadcast ds_read_b128
    double *const l_cx = lds + (TRT_SWEEP_LDS ? cull.padded * 2 : 0), *const l_cy = l_cx + n, *const l_cz = l_cy + n, *const l_r2 = l_cz + n;
    double *const l_mat = l_r2 + n;
    double *const l_dir = l_mat + (n + 2) * 5;
    double *const l_pt = l_dir + nd * 6;
    double *const l_255 = l_pt + np * 7;
    double *const l_cam = l_255 + 256;         // rarely-used frame constants live in LDS, not in SGPRs
    double *const l_jit = l_cam + kLdsCameraDoubles;

    for (int i = threadIdx.x; i < n; i += blockDim.x)
    {
        const double *sp = s.spheres + (long)i * kSphereDoubles;
        l_cx[i] = sp[0];
        l_cy[i] = sp[1];
        l_cz[i] = sp[2];
        l_r2[i] = sp[3] * sp[3]; // radius*radius exactly as TRT.c:648 forms it
        for (int j = 0; j < 5; j++)
            l_mat[i * 5 + j] = sp[4 + j];
    }
    for (int i = threadIdx.x; i < 10; i += blockDim.x)
        l_mat[n * 5 + i] = s.ground[6 + i]; // even material, odd material
    for (int i = threadIdx.x; i < nd; i += blockDim.x)
    {
        const double *li = s.dir_lights + i * kDirLightDoubles;
        d3 tl = unit(scale(load3(li), -1.0)); // TRT.c:903-904, the same value for every hit point
        l_dir[i * 6 + 0] = tl.x, l_dir[i * 6 + 1] = tl.y, l_dir[i * 6 + 2] = tl.z;
        l_dir[i * 6 + 3] = li[3], l_dir[i * 6 + 4] = li[4], l_dir[i * 6 + 5] = li[5];
    }
    for (int i = threadIdx.x; i < np * 7; i += blockDim.x)
        l_pt[i] = s.point_lights[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x)
        l_255[i] = (double)i / 255.0; // TRT.c:866
    for (int i = threadIdx.x; i < 12; i += blockDim.x)
        l_cam[i] = f.cam[i];
    if (threadIdx.x == 0)
        l_cam[12] = -f.cam[12]; // TRT.c:989
    for (int i = threadIdx.x; i < 2 * f.spp; i += blockDim.x)
        l_jit[i] = f.jitter[i];
    if (TRT_SWEEP_LDS)
        for (int i = threadIdx.x; i < cull.padded; i += blockDim.x)
            l_cull[i] = ((const float4 *)cull.table)[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    // number of work units; < 2^31, checked by the host
    const unsigned total = (unsigned)f.local_rows * (unsigned)f.width * (SAMPLE_UNITS ? (unsigned)f.spp : 1u);
    const d3 gp = load3(s.ground), gn = load3(s.ground + 3);
    const const_float_ptr table = (const_float_ptr)(uintptr_t)cull.table;

    // ---- per-lane state --------------------------------------------------------------------------------
    bool alive = true;   // false once the queue is empty and this lane has written its last pixel
    bool has_ray = false;
    int mode = kModeBoot;
    unsigned pix = 0;
    double sx_base = 0.0, sy_base = 0.0; // screen coordinates of the pixel before jitter (TRT.c:987-988)
    int k = 0;                     // sample index within the pixel
    d3 mean = d3{0.0, 0.0, 0.0};   // sum over samples (average_pixel_color, TRT.c:977)
    d3 sample = d3{0.0, 0.0, 0.0}; // pixel_color of the current sample (TRT.c:1012)
    double weight = 1.0, weight_sum = 0.0;
    int bounces = 0;
    d3 o = d3{0.0, 0.0, 0.0}, d = d3{0.0, 0.0, -1.0};
    // surface found by the path ray, kept while its shadow rays are traced
    d3 h_point = o, h_normal = d, path_dir = d, lit = d3{0.0, 0.0, 0.0};
    int h_mat = 0; // index into l_mat (sphere i, n = ground even, n+1 = ground odd)
    double light_d2 = 0.0, strength = 0.0;
    unsigned n_path = 0, n_shadow = 0, n_trips = 0, n_phase2 = 0;
    unsigned pool_next = 0, pool_end = 0; // wave-uniform: units [pool_next, pool_end) fetched from the queue, not yet handed out

#if TRT_STAMP
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    while (__any(alive))
    {
        TRT_STAMP_AT(7); // loop back-edge / exit test
        if (COUNT)
            n_trips++;
        // =================================== TRACE (TRT.c:793-856) ===================================
        const double a = dot(d, d);
        double best_d2 = __builtin_inf();
        d3 best_p = o;
        int best_i = -1; // sphere index, or n for the ground
        if (COUNT && has_ray)
        {
            if (mode == kModePath)
                n_path++;
            else
                n_shadow++;
        }
        // a directional-light shadow ray only asks "anything hit?" (TRT.c:908): any hit ends its search
        const bool any_hit_suffices = mode >= 1 && mode <= nd;

        trt_ray_filter flt;
        trt_filter_setup(&flt, o.x, o.y, o.z, d.x, d.y, d.z, a, cull.c0x, cull.c0y, cull.c0z, cull.cn, cull.rm);
        if (TRT_DUP & 1)
        {
            d3 o2 = o, d2 = d;
            asm volatile("" : "+v"(o2.x), "+v"(o2.y), "+v"(o2.z), "+v"(d2.x), "+v"(d2.y), "+v"(d2.z));
            trt_ray_filter f2;
            trt_filter_setup(&f2, o2.x, o2.y, o2.z, d2.x, d2.y, d2.z, dot(d2, d2), cull.c0x, cull.c0y, cull.c0z, cull.cn, cull.rm);
            if (f2.neg_thr != flt.neg_thr || f2.cd_min != flt.cd_min)
                flt.ok = 0; // never taken
        }

        TRT_STAMP_AT(0); // filter set-up
        for (int base = 0; base < cull.padded; base += 64)
        {
            // phase 1: uniform sweep of up to 64 table entries, scalar-loaded.  Each test leaves its verdict in a
            // sign bit that v_alignbit shifts into a per-lane word: sphere base+j ends up at bit 63-j of `cand`.
            const int chunk = (cull.padded - base) < 64 ? (cull.padded - base) : 64;
            unsigned word[2] = {~0u, ~0u}; // all "reject" until shifted in
            unsigned word_dup = 0;
            for (int rep = 0; rep < ((TRT_DUP & 2) ? 2 : 1); rep++)
            {
            if (TRT_DUP & 2)
                asm volatile("" : "+v"(flt.dx), "+v"(flt.wx), "+v"(flt.neg_thr));
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
                        if (TRT_SWEEP_LDS)
                        {
                            const float4 e = l_cull[first + g + j]; // same address in every lane: LDS broadcast
                            bits = __builtin_amdgcn_alignbit(bits, trt_filter_sign(&flt, e.x, e.y, e.z, e.w), 31);
                        }
                        else
                        {
                            const const_float_ptr e = table + (long)(first + g + j) * 4;
                            bits = __builtin_amdgcn_alignbit(bits, trt_filter_sign(&flt, e[0], e[1], e[2], e[3]), 31);
                        }
                    }
                }
                // the first sphere of this half must sit at bit 31: shift out the `32 - count` untouched bits
                if (rep == 1)
                    word_dup |= bits ^ ~word[h]; // identical sweep: contributes nothing
                word[h] = count > 0 ? ~(bits << (32 - count)) & (count == 32 ? ~0u : ~((1u << (32 - count)) - 1u)) : 0u;
            }
            }
            if ((TRT_DUP & 2) && word_dup == 0x12345u)
                word[0] = 0; // never taken
            unsigned long long cand = ((unsigned long long)word[0] << 32) | word[1];
            if (!flt.ok)
                cand = chunk == 64 ? ~0ull : ~((1ull << (64 - chunk)) - 1ull); // degenerate ray: every sphere of the chunk
            if (!has_ray)
                cand = 0;
            TRT_STAMP_AT(1); // phase 1
            // phase 2: exact FP64 tests of this lane's candidates, ascending index (first index wins ties)
            const unsigned long long cand_again = cand;
            for (int rep2 = 0; rep2 < ((TRT_DUP & 16) ? 2 : 1); rep2++)
            {
            if (rep2 == 1)
                cand = cand_again;
            while (__any(cand != 0))
            {
                if (COUNT)
                    n_phase2++;
                if (cand != 0)
                {
                    const int lead = __builtin_clzll(cand); // highest bit = lowest sphere index
                    const int i = base + lead;
                    cand &= ~(0x8000000000000000ull >> lead);
                    if (i < n)
                    {
                        const d3 c = d3{l_cx[i], l_cy[i], l_cz[i]};
                        const d3 oc = sub(o, c);
                        const double b = 2.0 * dot(oc, d);
                        const double cc = dot(oc, oc) - l_r2[i];
                        const double disc = b * b - 4.0 * a * cc;
                        if (!(disc < 0.0))
                        {
                            const double t0 = (-b - __builtin_sqrt(disc)) / (2.0 * a);
                            if (t0 > 0.0)
                            {
                                const d3 p = d3{o.x + t0 * d.x, o.y + t0 * d.y, o.z + t0 * d.z};
                                const double d2 = dist2(o, p);
                                if (d2 < best_d2)
                                {
                                    best_d2 = d2;
                                    best_p = p;
                                    best_i = i;
                                }
                                if (any_hit_suffices)
                                    cand = 0;
                            }
                        }
                    }
                }
            }
            }
            TRT_STAMP_AT(2); // phase 2
        }
        // ground plane (TRT.c:831-853)
        if (has_ray && !(any_hit_suffices && best_i >= 0))
        {
            d3 p;
            if (hit_plane(o, d, gp, gn, p))
            {
                const double d2 = dist2(o, p);
                if (d2 < best_d2)
                {
                    best_d2 = d2;
                    best_p = p;
                    best_i = n;
                }
            }
        }

        TRT_STAMP_AT(3); // plane
        if (TRT_DUP & 4)
        {
            d3 o2 = o, d2 = d;
            asm volatile("" : "+v"(o2.x), "+v"(o2.y), "+v"(o2.z), "+v"(d2.x), "+v"(d2.y), "+v"(d2.z));
            d3 p;
            if (has_ray && !(any_hit_suffices && best_i >= 0) && hit_plane(o2, d2, gp, gn, p))
                if (dist2(o2, p) < best_d2)
                    best_i = n; // never changes anything: the first test already took it
        }

        // =================================== POST ===================================
        // Decide what this lane does next; emit the vectors to normalise (uA, uB) and one quotient (qn/qd).
        d3 uA = d3{0.0, 0.0, 1.0}, uB = d3{0.0, 0.0, 1.0};
        double qn = 0.0, qd = 1.0;
        bool needA = false, needB = false;
        bool path_hit = false, path_sky = false, shadow_back = false, lighting_done = false;
        bool end_sample = false, end_pixel = false, want_pixel = false;
        double weight_sum_new = weight_sum, light_d2_next = 0.0;
        const bool hit = best_i >= 0;
        if (has_ray && mode == kModePath)
        {
            if (hit)
            {
                path_hit = true;
                uA = sub(o, best_p); // nudge direction, TRT.c:871-872
                needA = true;
                if (best_i < n)
                {
                    uB = sub(best_p, d3{l_cx[best_i], l_cy[best_i], l_cz[best_i]}); // TRT.c:824
                    h_mat = best_i;
                }
                else
                {
                    uB = gn;
                    h_mat = n + checker_odd(best_p); // TRT.c:850-851
                }
                needB = true;
            }
            else
            {
                path_sky = true; // TRT.c:858-867, :1044-1048: the sample ends on the sky
                uA = d;          // get_skybox_color normalises the direction again, TRT.c:702
                needA = true;
                end_sample = true;
            }
        }
        else if (has_ray)
        {
            shadow_back = true;
            const int li = mode - 1;
            if (li >= nd && hit)
            { // point light with a blocker: is the blocker farther than the light?  needs the nudged point (TRT.c:939-942)
                uA = sub(o, best_p);
                needA = true;
            }
            if (li + 1 < nl)
            {
                if (li + 1 >= nd)
                { // next light is a point light: TRT.c:929-933
                    const double *pl = l_pt + (li + 1 - nd) * 7;
                    uB = sub(load3(pl), h_point);
                    light_d2_next = dot(uB, uB);
                    qn = pl[6];
                    qd = light_d2_next;
                    needB = true;
                }
            }
            else
                lighting_done = true;
        }
        else if (alive && mode >= 1)
            lighting_done = true; // a scene without lights: the surface found last iteration is shaded black
        if (lighting_done)
        { // does the bounce loop go on after this surface?  TRT.c:1018, :1041-1042
            const double w_next = weight * l_mat[h_mat * 5 + 3];
            if (bounces + 1 < f.bounce_limit && w_next > 0.00001)
            {
                uB = reflect(path_dir, h_normal); // TRT.c:1054
                needB = true;
            }
            else
                end_sample = true;
        }
        int k_next = k;
        if (end_sample)
        {
            weight_sum_new = weight_sum + weight; // TRT.c:1034
            qn = 1.0;
            qd = weight_sum_new; // TRT.c:1061
            k_next = k + 1;
            if (SAMPLE_UNITS || k_next == f.spp)
            {
                end_pixel = true; // the unit (pixel, or single sample) is finished
                k_next = 0;
            }
        }
        want_pixel = alive && (end_pixel || mode == kModeBoot);
        unsigned pix_next = pix; // index of the lane's work unit
        double sx_next = sx_base, sy_next = sy_base;
        {
            const unsigned long long need = __ballot(want_pixel);
            if (need)
            { // serve the lanes from the wave's pool; refill the pool with ONE atomic per kQueueChunk units
                const unsigned wanted = (unsigned)__builtin_popcountll(need);
                unsigned first = pool_next; // wave-uniform
                if (pool_end - pool_next < wanted)
                { // not enough left: fetch a fresh chunk (what is left of the old one is served first)
                    const unsigned left = pool_end - pool_next;
                    const unsigned chunk = SAMPLE_UNITS ? kQueueChunkSamples : kQueueChunkPixels;
                    unsigned fresh = 0;
                    const int leader = __builtin_ctzll(need);
                    if (lane == leader)
                        fresh = atomicAdd(f.queue, chunk);
                    fresh = __shfl(fresh, leader);
                    // lanes ranked < left take the old units, the others the first units of the new chunk
                    const unsigned rank = (unsigned)__builtin_popcountll(need & ((1ull << lane) - 1ull));
                    if (want_pixel)
                        pix_next = rank < left ? pool_next + rank : fresh + (rank - left);
                    pool_next = fresh + (wanted - left);
                    pool_end = fresh + chunk;
                    first = 0xffffffffu; // marks "already assigned"
                }
                else
                    pool_next += wanted;
                if (want_pixel)
                {
                    if (first != 0xffffffffu)
                        pix_next = first + (unsigned)__builtin_popcountll(need & ((1ull << lane) - 1ull));
                    if (pix_next < total)
                    {
                        unsigned pixel = pix_next;
                        if (SAMPLE_UNITS)
                        { // unit -> (pixel, k) by multiply-high with min(ceil(2^32/spp), 2^32-1): off by at most one
                            pixel = __umulhi(pix_next, f.spp_magic);
                            int kk = (int)(pix_next - pixel * (unsigned)f.spp);
                            if (kk < 0)
                            {
                                pixel--;
                                kk += f.spp;
                            }
                            else if (kk >= f.spp)
                            {
                                pixel++;
                                kk -= f.spp;
                            }
                            k_next = kk;
                        }
                        // row = pixel / width the same way
                        unsigned row = __umulhi(pixel, f.width_magic);
                        int col = (int)(pixel - row * (unsigned)f.width);
                        if (col < 0)
                        {
                            row--;
                            col += f.width;
                        }
                        else if (col >= f.width)
                        {
                            row++;
                            col -= f.width;
                        }
                        sx_next = f.col_x[col];
                        sy_next = f.row_y[frame_row_of(f, (int)row)];
                    }
                }
            }
        }
        const bool new_sample = (end_sample || mode == kModeBoot) && alive && pix_next < total;
        if (new_sample)
        { // primary ray of sample k_next of pixel pix_next (TRT.c:987-1005); normalised in NORM
            const double sx = sx_next + l_jit[k_next];
            const double sy = sy_next + l_jit[f.spp + k_next];
            d3 dir = d3{0.0, 0.0, 0.0};
            dir = add(dir, scale(load3(l_cam + 0), sx));
            dir = add(dir, scale(load3(l_cam + 3), sy));
            dir = add(dir, scale(load3(l_cam + 6), l_cam[12]));
            uB = sub(dir, load3(l_cam + 9)); // sic, TRT.c:1005
            needB = true;
        }

        TRT_STAMP_AT(4); // POST
        // =================================== NORM (convergent) ===================================
        d3 nA = uA, nB = uB;
        double q = 0.0;
        if (__any(needA))
            nA = unit(uA);
        if (__any(needB))
            nB = unit(uB);
        if (__any(end_sample || (shadow_back && needB)))
            q = qn / qd;
        if (TRT_DUP & 8)
        { // same work again on opaque copies of the inputs
            d3 vA = uA, vB = uB;
            double vn = qn, vd = qd;
            asm volatile("" : "+v"(vA.x), "+v"(vA.y), "+v"(vA.z), "+v"(vB.x), "+v"(vB.y), "+v"(vB.z), "+v"(vn), "+v"(vd));
            d3 mA = vA, mB = vB;
            double q2 = 0.0;
            if (__any(needA))
                mA = unit(vA);
            if (__any(needB))
                mB = unit(vB);
            if (__any(end_sample || (shadow_back && needB)))
                q2 = vn / vd;
            if (mA.x != nA.x || mB.y != nB.y || q2 != q)
                nA.z = mA.z; // never taken: identical by construction
        }

        TRT_STAMP_AT(5); // NORM
        // =================================== FINISH ===================================
        if (path_hit)
        {
            h_point = add(best_p, scale(nA, 0.000001)); // TRT.c:873-874
            h_normal = nB;                              // TRT.c:878
            path_dir = d;
            lit = d3{0.0, 0.0, 0.0};
            o = h_point;
            if (nl == 0)
            {
                has_ray = false; // nothing to trace: shaded black next iteration
                mode = 1;
            }
            else if (nd > 0)
            {
                d = load3(l_dir); // TRT.c:903-907
                mode = 1;
            }
            else
            { // first light is a point light: its direction depends on h_point, which only now exists
                const double *pl = l_pt;
                d3 to_light = sub(load3(pl), h_point);
                light_d2 = dot(to_light, to_light);
                strength = clampd(pl[6] / light_d2, 0.0, 1.0);
                d = unit(to_light);
                mode = 1;
            }
        }
        else if (shadow_back)
        { // TRT.c:908-922 / :939-956; d is the unit vector to the light
            const int li = mode - 1;
            bool is_lit = !hit;
            double factor = min1(dot(h_normal, d));
            d3 lcolor;
            if (li < nd)
                lcolor = load3(l_dir + li * 6 + 3);
            else
            {
                if (hit)
                {
                    const d3 to_blocker = sub(add(best_p, scale(nA, 0.000001)), o);
                    is_lit = light_d2 < dot(to_blocker, to_blocker);
                }
                factor = strength * factor;
                lcolor = load3(l_pt + (li - nd) * 7 + 3);
            }
            if (is_lit)
                lit = add(lit, mulc(scale(lcolor, factor), load3(l_mat + h_mat * 5)));
            if (!lighting_done)
            { // next shadow ray from the same surface point (o stays h_point)
                if (li + 1 < nd)
                    d = load3(l_dir + (li + 1) * 6);
                else
                {
                    d = nB;
                    light_d2 = light_d2_next;
                    strength = clampd(q, 0.0, 1.0);
                }
                mode = li + 2;
            }
        }
        if (lighting_done)
        { // TRT.c:960-962 then :1034-1056
            d3 color = d3{clampd(lit.x, 0.0, 1.0), clampd(lit.y, 0.0, 1.0), clampd(lit.z, 0.0, 1.0)};
            if (!end_sample)
                weight_sum += weight;
            color = scale(color, weight);
            weight *= l_mat[h_mat * 5 + 3];
            bounces++;
            sample = add(sample, color);
            if (!end_sample)
            {
                d = nB;
                o = h_point;
                mode = kModePath;
                has_ray = true;
            }
        }
        if (path_sky)
        { // colour of the sky texel (TRT.c:865-866); the sample ends
            const uint32_t t = sky_texel_unit(s.sky, s.sky_dim, nA);
            const d3 color = d3{l_255[t & 0xFF], l_255[(t >> 8) & 0xFF], l_255[(t >> 16) & 0xFF]};
            sample = add(sample, scale(color, weight));
        }
        if (end_sample)
        { // TRT.c:1061-1066
            sample = scale(sample, q);
            if (SAMPLE_UNITS)
            {
                const unsigned px_ = __umulhi(pix, f.spp_magic); // unit -> (pixel, k) once more for the sample-major slot
                int k_ = (int)(pix - px_ * (unsigned)f.spp);
                const unsigned pixel_ = k_ < 0 ? px_ - 1 : (k_ >= f.spp ? px_ + 1 : px_);
                k_ = (int)(pix - pixel_ * (unsigned)f.spp);
                double *out = f.samples + ((size_t)k_ * ((size_t)f.local_rows * f.width) + pixel_) * 3;
                out[0] = sample.x;
                out[1] = sample.y;
                out[2] = sample.z;
            }
            else
            {
                mean = add(mean, sample);
                if (end_pixel)
                {
                    mean = scale(mean, f.inv_spp);
                    double *out = f.out + (size_t)pix * 3;
                    out[0] = mean.x;
                    out[1] = mean.y;
                    out[2] = mean.z;
                    mean = d3{0.0, 0.0, 0.0};
                }
            }
        }
        if (end_sample || mode == kModeBoot)
        {
            if (new_sample)
            {
                pix = pix_next;
                sx_base = sx_next;
                sy_base = sy_next;
                k = k_next;
                d = nB;
                o = load3(l_cam + 9);
                sample = d3{0.0, 0.0, 0.0};
                weight = 1.0;
                weight_sum = 0.0;
                bounces = 0;
                mode = kModePath;
                has_ray = true;
            }
            else
            {
                alive = false;
                has_ray = false;
            }
        }
        TRT_STAMP_AT(6); // FINISH
    }

    if (COUNT && f.counters)
    {
        atomicAdd(&f.counters[0], (unsigned long long)n_path);
        atomicAdd(&f.counters[1], (unsigned long long)n_shadow);
        if (lane == 0)
        { // diagnostics: loop trips and phase-2 rounds per wave (lane utilisation = traces / (64 * trips))
            atomicAdd(&f.counters[2], (unsigned long long)n_trips);
            atomicAdd(&f.counters[3], (unsigned long long)n_phase2);
#if TRT_STAMP
            for (int i = 0; i < 8; i++)
                atomicAdd(&f.counters[4 + i], stamp_sum[i]);
#endif
        }
    }
}

// TRT.c:1063-1066 for frames rendered with samples as work units: pixel = (((0 + s0) + s1) + ...) * (1/spp),
// samples in index order.  The scratch is sample-major, samples[(k*pixels + pixel)*3 + channel], so that for every k
// consecutive threads read consecutive doubles (a pure streaming kernel: spp*24 B read + 24 B written per pixel).
__global__ __launch_bounds__(256) void reduce_samples_kernel(const double *samples, double *out, long values, int spp, double inv_spp)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; // one thread per colour channel of a pixel
    if (i >= values)
        return;
    double mean = 0.0;
    for (int k = 0; k < spp; k++)
        mean += samples[(long)k * values + i];
    out[i] = mean * inv_spp;
}

} // namespace trt
