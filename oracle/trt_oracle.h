/*
 * trt_oracle.h -- CPU restatement of the reference's frame producer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker.  The product path (libtrt_hip.so) never
 * links or calls it.
 *
 * Parity pinning: this restatement is checked bit-for-bit (FNV-1a-64 of the
 * double framebuffer, and trace_ray call counts) against the genuine
 * reference compiled from /root/reference by oracle/Makefile (oracle/_ref),
 * and against the known-answer table of SURVEY.md section 8c; the vectors are
 * committed under tests/golden/ (tests/golden/make_golden.py).
 */
#ifndef TRT_ORACLE_H
#define TRT_ORACLE_H

#include "trt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
    unsigned long long path_rays;    /* trace calls issued by the bounce loop (primary + secondary), TRT.c:1024 */
    unsigned long long shadow_rays;  /* trace calls issued by lighting, TRT.c:907 and :937 */
    unsigned long long sky_lookups;  /* skybox samples whose result is used (path rays that hit nothing) */
    unsigned long long samples;      /* W*H*rays_per_pixel */
} trt_oracle_stats;

/* whole frame; num_threads<=1 is the faithful single-threaded order, >1 splits rows with OpenMP
 * (pixels are independent, so the framebuffer is identical either way) */
void trt_oracle_project_scene(const Scene *scene, Screen *screen, int bounce_limit, int rays_per_pixel,
                              int num_threads, trt_oracle_stats *stats);

/* rows [row_begin,row_end) of a W x H frame into out[(row-row_begin)*W + col] */
void trt_oracle_render_rows(const Scene *scene, Vector *out, int width, int height, int row_begin, int row_end,
                            int bounce_limit, int rays_per_pixel, int num_threads, trt_oracle_stats *stats);

/* TRT.c:793 -- outputs may be NULL */
ObjectType trt_oracle_trace_ray(const Scene *scene, const Ray *ray, Point *intersection, Vector *normal, Material *material);

/* TRT.c:700 -- writes the linear texel index it would read (may equal dim*dim: reference's latent overrun) */
int trt_oracle_skybox_lookup(const Scene *scene, const Vector *direction, int *face, long *texel_index);

/* TRT.c:894 */
void trt_oracle_apply_lighting(const Scene *scene, const Point *intersection, const Vector *normal, Material *material,
                               trt_oracle_stats *stats);

/* TRT.c:225 */
double trt_oracle_triangle_wave(double t);

/* (int)(c*255) per channel, TRT.c:1157-1163 */
void trt_oracle_rgb8(const Vector *pixels, size_t count, unsigned char *rgb);

/* FNV-1a-64, offset 1469598103934665603, prime 1099511628211 */
unsigned long long trt_oracle_fnv1a64(const void *data, size_t bytes);

/* Ray log for tests: while a log is installed (per thread), every traced ray is appended as
 * 6 doubles (origin, direction) with kind 0 = path ray, 1 = directional-light shadow ray,
 * 2 = point-light shadow ray, until capacity is reached.  NULL uninstalls. */
typedef struct
{
    double *rays;
    unsigned char *kinds;
    size_t capacity;
    size_t count;
} trt_oracle_ray_log;
void trt_oracle_set_ray_log(trt_oracle_ray_log *log);

/* EXTENSION, PARITY UNPINNED (the reference has no refraction): the semantics of the HIP kernel's refraction variant
 * restated on the CPU, so that the GPU has something to be checked against bit for bit.  ior[i] > 0: sphere i refracts
 * with that index (relative to the outside); 0: the reference's opaque sphere.  See the block comment in trt_oracle.c. */
void trt_oracle_project_scene_refractive(const Scene *scene, const double *ior, Screen *screen, int bounce_limit, int rays_per_pixel,
                                         int num_threads, trt_oracle_stats *stats);

/* exact a/b and sqrt(a) tables for the device rounding self-test */
void trt_oracle_div_sqrt(const double *a, const double *b, size_t n, double *quot, double *root);

#ifdef __cplusplus
}
#endif
#endif
