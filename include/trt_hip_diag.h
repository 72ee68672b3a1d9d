/*
 * trt_hip_diag.h -- DIAGNOSTIC and TEST entries of libtrt_hip.so.  Not part of the drop-in boundary (include/trt_hip.h): nothing
 * here is needed to produce a frame, a maintainer of the reference would not bind any of it, and any of it may change with the
 * kernels.  Used by tests/, bench.py (ray counts, loop diagnostics) and tools/.
 *
 *   counters      trace_ray calls of the last frame as the kernel counted them (the metric's numerator), loop diagnostics
 *   read-backs    the candidate tables the device built, for comparison with the host reference builders
 *   self-tests    device division / square root / normalisation / cube instructions / skybox estimate against their references
 *   probes        single rays through the reference-order kernel and through the production kernel's stages
 *   test hooks    the pool cap of the long candidate lists; a stand-in for RCCL so that several ranks can share one GPU
 */
#ifndef TRT_HIP_DIAG_H
#define TRT_HIP_DIAG_H

#include "trt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tests: cap the scene's part of the pool of long candidate lists, and the part of every eye slot (the eye's two tables are
 * rebuilt per camera into a part of their own), at `words` 64-bit words (0 = automatic) from the next trt_set_scene on; lists
 * that find no room leave their cell without a list and its rays sweep -- frames stay bit-identical. */
int trt_set_list_pool_words(trt_context *ctx, size_t words);

/* Work counters of the LAST rendered frame (device atomics, only when enabled; off by default
 * so timed runs carry no atomics).  path = trace calls from the bounce loop (TRT.c:1024),
 * shadow = trace calls from lighting (TRT.c:907, :937). */
int trt_enable_counters(trt_context *ctx, int enable);

int trt_read_counters(trt_context *ctx, unsigned long long *path_rays, unsigned long long *shadow_rays);

/* Diagnostics of the last counted frame (valid after trt_read_counters, production kernel only):
 * iterations of the per-wave main loop summed over waves, and exact-test (phase 2) rounds.
 * Lane utilisation of the trace loop = (path + shadow) / (64 * wave_loop_trips). */
int trt_read_diagnostics(trt_context *ctx, unsigned long long *wave_loop_trips, unsigned long long *phase2_rounds);

/* The family code of the production kernel for a path ray (trt_probe_rays_production): kind 0 = the ray starts at the eye,
 * 1 = reflected by the ground, parent from the eye, 2 = the ray starts on `sphere`, 3 = reflected by the ground, its parent
 * started on `sphere` at parent_origin (3 doubles: the patch of the sphere follows from it).  -1: no family. */
int trt_path_family_code(trt_context *ctx, int kind, int sphere, const double *parent_origin);

/* Copy the path rays' tables to the host (tests: the device-built lists must equal the host reference builder's): list
 * cells (2*6*eye_cells^2, then 2NP*6*sphere_cells^2 for P patches per sphere: patch k of sphere i at i P + k, then the mirror
 * images in the same order) and the pool of long lists, as built for `camera`'s eye.
 * info: {enabled, eye_cells, sphere_cells, N, cells, pool words used by the scene's tables, by the eye's, pool capacity}.
 * Returns the number of cells copied, 0 when the tables are off, or a negative TRT_ERR_*. */
long trt_read_path_tables(trt_context *ctx, const Camera *camera, unsigned long long *cells, size_t capacity_cells,
                          unsigned long long *pool, size_t capacity_pool, long info[8]);

/* After trt_read_counters: the number of wave-level traces of the last counted frame in which some ray failed its table's
 * membership / range test and the whole wave swept the culling table instead (the slow path). */
int trt_read_sweep_fallbacks(trt_context *ctx, unsigned long long *swept_traces);

/* Likewise: how many times a wave ran the shadow stage (over up to 64 hits each time).  hits / (64 * passes) is the lane
 * activity of the shadow stage; the hits of a frame are shadow_rays / number of lights. */
int trt_read_shading_passes(trt_context *ctx, unsigned long long *passes);

/* Likewise, the exact-test loops of the three kinds of trace (path rays TRT.c:1024, directional-light shadow rays TRT.c:907,
 * point-light shadow rays TRT.c:937): out[0..2] = wave-level iterations of each loop, out[3..5] = exact sphere tests summed over
 * lanes (tests / (64 * iterations) = the loop's lane activity: a wave iterates as long as its busiest lane), out[6] = point-light
 * shadow searches in which some lane's any-hit search was inconclusive and the wave ran the closest-hit search, out[7] = 0. */
int trt_read_loop_diagnostics(trt_context *ctx, unsigned long long out[8]);

/* Copy one light's table to the host (tests: the device-built table must equal the host reference builder's).
 * point_light: 0 = directional light `index`, 1 = point light `index`.  Returns the number of 64-bit words copied
 * (cells * ceil(N/64); cells = slabs * g^2 resp. shells * 6 g^2), 0 when the tables are off, or a negative TRT_ERR_*. */
long trt_read_light_grid(trt_context *ctx, int point_light, int index, unsigned long long *masks, size_t capacity_words);

/* Device rounding self-test: quot[i] = a[i] / b[i], root[i] = sqrt(a[i]) computed by the same
 * device instructions sequences the kernels use (host arrays in/out). */
int trt_selftest_div_sqrt(trt_context *ctx, const double *a, const double *b, size_t n, double *quot, double *root);

/* The kernels normalise vectors (TRT.c:439-450) with one shared reciprocal refinement for the three divisions and a
 * square root without the compiler's range handling whenever a whole wave's operands are far from the ends of the
 * exponent range.  This entry runs that code (`fast`) and the compiler's plain `/` and sqrt (`reference`) on n
 * records {x, y, z, w}: out = {unit(x,y,z), sqrt(w)}.  The two outputs must be identical bit for bit. */
int trt_selftest_unit(trt_context *ctx, const double *xyzw, size_t n, double *fast, double *reference);

/* The candidate tables over cube maps (point lights, ray families, surface patches) take face and face coordinates of a
 * direction from gfx9's v_cubeid / v_cubesc / v_cubetc / v_cubema; the host-side builders and checkers use a C restatement of the
 * four instructions (csrc/trt_lightgrid.h, trt_cube_lookup).  n directions {x, y, z} in, {face, sc, tc, 2 * major} per direction out:
 * as the device computes them and as the host restatement does.  The two must agree bit for bit. */
int trt_selftest_cube(trt_context *ctx, const float *xyz, size_t n, float *device_out, float *host_out);

/* Tests: the skybox look-up of the production kernel (get_skybox_color, TRT.c:700-789) takes the texel's index from an FP32
 * estimate of the texel coordinates whenever that provably truncates as the reference's FP64 value does, and from the FP64 form
 * otherwise (csrc/trt_device.hpp: sky_index_estimate).  For n unit directions (3 doubles each) and a cubemap of side dim:
 * exact[i] = face dim^2 + vi dim + ui by the FP64 form, estimate[i] = by the estimate, ambiguous[i] != 0 where the estimate
 * does not vouch for itself.  The claim under test: ambiguous[i] == 0  =>  estimate[i] == exact[i]. */
int trt_selftest_sky(trt_context *ctx, const double *dirs, size_t n, int dim, long long *exact, long long *estimate, int *ambiguous);

/* Single-ray probe for tests: closest hit of TRT.c:793 for n rays (host arrays): obj[n],
 * point[3n], normal[3n], material[5n] (colour, reflectivity, specularity); lit[3n] = colour after
 * the lighting of TRT.c:894 for hits. */
int trt_probe_rays(trt_context *ctx, const Ray *rays, size_t n, int *obj, double *point, double *normal,
                   double *material, double *lit);

/* The same probe through the PRODUCTION kernel's stages (csrc/trt_rounds.hpp: candidate tables, fall-back sweep, exact
 * tests, surface record, lighting), so that trace_ray (TRT.c:793), ray_intersects_sphere/plane (:638, :677),
 * get_skybox_color (:700) and apply_lighting (:894) are each checked on the code that ships.  families[i] = the kernel's
 * family code of ray i (trt_path_family_code: 0 eye, 1 mirror eye, 2 + s starts on sphere s, codes from 2 + N on: mirror image
 * of a patch of a sphere; negative or families == NULL: none, the ray's wave sweeps); a ray that is not a member of the
 * family named falls back by itself.
 * camera: its origin is the eye the tables of families 0 and 1 are built for. */
int trt_probe_rays_production(trt_context *ctx, const Camera *camera, const Ray *rays, const int *families, size_t n, int *obj,
                              double *point, double *normal, double *material, double *lit);

/* Measurement: the number of frames a context has launched so far (launch numbers count from 0), and the DEVICE time in ms from the
 * start of launch `first_launch` of context `first` to the end -- render kernel and ordered mean -- of launch `last_launch` of
 * context `last` (HIP events on the contexts' streams; both of one device; each among the last 256 launches of its context).
 * A pipelined loop deals its frames over several contexts: span / frames is its device time per frame, which bench.py prints
 * beside the host-clocked ms_per_step. */
long trt_launch_count(trt_context *ctx);
int trt_launch_span_ms(trt_context *first, long first_launch, trt_context *last, long last_launch, float *ms);

/* TEST HOOK: allow (1) or forbid (0, the default) the environment variable TRT_RCCL_LIB to name the library that trt_dist_* binds in
 * RCCL's place (tests/rccl_stub.cpp: send / recv through shared memory, so that several ranks can share the one GPU of a test
 * box).  Must be called before the process's first use of RCCL: fails with TRT_ERR_NOT_INITIALISED once RCCL has been bound.  A
 * process that never calls it ignores the variable.  trt_dist_rccl_library() (trt_hip.h) says what was bound. */
int trt_dist_allow_rccl_override(int allow);

#ifdef __cplusplus
}
#endif
#endif
