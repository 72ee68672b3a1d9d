/*
 * trt_host.h -- host-side C companions of the frame producer: the pieces of the reference that sit
 * either side of project_scene() and stay on the CPU.
 *
 *   camera   per-frame orbit that produces the Camera input        TRT.c:290-305, 558-624, 1327-1336
 *   skybox   P6 reader + cubemap loader that produce the Skybox    TRT.c:309-436
 *   emitter  framebuffer -> 24-bit ANSI background-colour cells    TRT.c:1084-1172
 *
 * Plain C, no GPU calls; compiled into libtrt_hip.so next to the C-ABI of trt_hip.h.
 * Unlike the reference, nothing here calls exit(): errors come back as TRT_HOST_* codes.
 */
#ifndef TRT_HOST_H
#define TRT_HOST_H

#include <stdio.h>

#include "trt.h"

#ifdef __cplusplus
extern "C" {
#endif

enum
{
    TRT_HOST_OK = 0,
    TRT_HOST_ERR_OPEN = -101,    /* TRT.c:318-322  "Error opening file" */
    TRT_HOST_ERR_FORMAT = -102,  /* TRT.c:327-332  not a P6 file / malformed header */
    TRT_HOST_ERR_MAXVAL = -103,  /* TRT.c:351-356  max colour value is not 255 */
    TRT_HOST_ERR_MEMORY = -104,  /* TRT.c:363-368 */
    TRT_HOST_ERR_SHAPE = -105,   /* TRT.c:413-417  faces must be square and equal */
    TRT_HOST_ERR_ARGUMENT = -106,
    TRT_HOST_ERR_TRUNCATED = -107 /* fewer than width*height*3 data bytes (the reference stores EOF bytes silently) */
};

/* ---- camera ------------------------------------------------------------------------------- */
void trt_init_frame(Frame *frame);                                    /* TRT.c:290 */
/* TRT.c:299: distance 1, height 5, width 5*aspect_w/aspect_h (the reference bakes 480/280) */
void trt_init_camera(Camera *camera, int aspect_w, int aspect_h);
void trt_rotate_basis(Basis *basis, const Basis *rotation);          /* TRT.c:558 */
void trt_rotate_basis_x(Basis *basis, double angle);                  /* TRT.c:576 */
void trt_rotate_basis_y(Basis *basis, double angle);                  /* TRT.c:586 */
void trt_rotate_basis_z(Basis *basis, double angle);                  /* TRT.c:596 */
void trt_transform_frame(Frame *frame, const Frame *transform);      /* TRT.c:607 */
/* TRT.c:1327-1336: frame of the orbiting camera at wall-clock second t (screen_* members untouched) */
void trt_orbit_camera(Camera *camera, double t);

/* ---- skybox ------------------------------------------------------------------------------- */
/* TRT.c:309.  *colors is malloc'ed (caller frees). */
int trt_read_ppm(const char *filename, Color **colors, int *width, int *height);
/* TRT.c:388.  directory holds +X.ppm -X.ppm +Y.ppm -Y.ppm +Z.ppm -Z.ppm (the reference's
 * "skybox/<name>").  On error nothing is left allocated and skybox->dim is -1. */
int trt_load_skybox(Skybox *skybox, const char *directory);
void trt_free_skybox(Skybox *skybox);                                 /* TRT.c:430 */

/* ---- emitter ------------------------------------------------------------------------------ */
typedef struct trt_emitter trt_emitter;
/* TRT.c:1102-1131: "\033[0;0H" then per row width cells "\033[48;2;RRR;GGG;BBBm  \033[0m" and '\n'.
 * The buffer has the reference's sizeof(screenbuffer) = 8 + (25*width+1)*height + 1 bytes. */
int trt_emitter_create(int width, int height, trt_emitter **out);
void trt_emitter_destroy(trt_emitter *e);
const char *trt_emitter_buffer(const trt_emitter *e);
size_t trt_emitter_size(const trt_emitter *e);
/* TRT.c:1142-1168: patch the 9 digits of every cell with (int)(c*255) (no output yet).  The Screen must have the
 * emitter's width and height (in the reference both come from SCREEN_WIDTH/HEIGHT): TRT_HOST_ERR_ARGUMENT otherwise. */
int trt_emitter_patch(trt_emitter *e, const Screen *screen);
/* the same from bytes already quantised on the GPU (trt_quantize_device): 3 bytes per pixel, width*height pixels */
int trt_emitter_patch_rgb8(trt_emitter *e, const unsigned char *rgb);
/* TRT.c:1171: one fwrite of the whole buffer (trailing NULs included, as the reference does) */
int trt_emitter_write(const trt_emitter *e, FILE *stream);
/* TRT.c:1084-1099: the unbuffered printf form */
int trt_draw_screen(const Screen *screen, FILE *stream);

/* ---- frame fingerprint ------------------------------------------------------------------ */
/* FNV-1a-64 with the offset SURVEY.md 8c states, 1469598103934665603 (NOT the textbook 14695981039346656037), and prime
 * 1099511628211, over raw bytes: the hash the golden frames are recorded with, e.g. over screen->pixels[0 .. W*H) after project_scene (TRT.c:966). */
unsigned long long trt_fnv1a64(const void *data, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
