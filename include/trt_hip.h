/*
 * trt_hip.h -- C-ABI of libtrt_hip.so, the MI355X (gfx950) frame producer.
 *
 * Plain C: pointers, ints and the structs of trt.h.  No C++/torch types cross it.
 * Each entry cites the reference interface it replaces or serves
 * ("TRT.c:N" = TerminalRayTracer.c line N of david-andrew/TerminalRayTracer).
 *
 * Two layers:
 *   1. drop-in          project_scene()            -- same symbol, same signature as TRT.c:966
 *   2. extended/context trt_*                      -- run-time bounce limit / rays per pixel
 *      (macros in the reference, TRT.c:54,58), device-resident framebuffers for off-screen
 *      rendering, row-tile sharding for multi-GPU, RGB8 quantisation for the emitter.
 *
 * This header is the PRODUCT boundary: the drop-in entries, the context API a host renders through, the tuning setters, the
 * multi-GPU entries.  What exists for the tests and the measurements only -- counters, table read-backs, self-tests, single-ray
 * probes, the RCCL stand-in hook -- is declared in include/trt_hip_diag.h; a host must not depend on it.
 *
 * All functions returning int return TRT_OK (0) or a negative TRT_ERR_*; trt_last_error()
 * gives the message.  A context is bound to one device and one stream and is not
 * thread-safe; separate contexts are independent.  The drop-in layer (project_scene, trt_render_frame,
 * trt_init/shutdown, trt_upload/invalidate_skybox) shares one default context and takes a lock: like the
 * reference's pure function it may be called from several threads, the calls then take turns.
 */
#ifndef TRT_HIP_H
#define TRT_HIP_H

#include "trt.h"

#ifdef __cplusplus
extern "C" {
#endif

enum
{
    TRT_OK = 0,
    TRT_ERR_HIP = -1,         /* a HIP runtime call failed */
    TRT_ERR_ARGUMENT = -2,    /* NULL / out-of-range argument */
    TRT_ERR_NO_SCENE = -3,    /* render before trt_set_scene */
    TRT_ERR_CAPACITY = -4,    /* output buffer too small, or scene too large for LDS staging */
    TRT_ERR_NOT_INITIALISED = -5
};

/* ---- 1. drop-in ------------------------------------------------------------------------------ */

/* Replaces `void project_scene(Scene *scene, Screen *screen)` (TRT.c:966; caller TRT.c:1339).
 * Same semantics at the reference's compile-time constants BOUNCE_LIMIT=10, RAYS_PER_PIXEL=10:
 * reads *scene (host pointers inside), overwrites screen->pixels[0 .. width*height).
 * Lazily creates a default context on device 0.  void like the original: a HIP failure prints
 * the error and aborts (the reference cannot fail here either). */
void project_scene(Scene *scene, Screen *screen);

/* project_scene with the two macros as run-time values (TRT.c:54, TRT.c:58).  Host in, host out. */
int trt_render_frame(const Scene *scene, Screen *screen, int bounce_limit, int rays_per_pixel);
/* The same entry under the name BASELINE.json's north_star gives it ("render_frame() entry"; the reference itself has no such
 * symbol: its frame producer is project_scene, TRT.c:966). */
int render_frame(const Scene *scene, Screen *screen, int bounce_limit, int rays_per_pixel);

/* The same frame as the bytes the emitter makes of it: rgb[(row*width + col)*3 + channel] = (int)(colour*255), the
 * conversion of buffered_draw_screen (TRT.c:1157-1163) done on the device, so that 3 bytes per pixel cross PCIe instead of 24.
 * For hosts whose only consumer of the frame is the terminal emitter (trt_emitter_patch_rgb8, trt_host.h).  Colours outside
 * [0, 1) convert as the reference's cast does on x86-64. */
int trt_render_frame_rgb8(const Scene *scene, int width, int height, int bounce_limit, int rays_per_pixel, unsigned char *rgb);

/* project_scene is a pure function of *scene (TRT.c:966): a caller may move a sphere before every call.  The drop-in entries
 * compare the primitives with the previous call's; a scene that has changed on `moving_after` consecutive calls counts as MOVING
 * and its candidate tables are rebuilt per call the cheap way (one family per sphere instead of 24 patches: ~4 ms instead of ~100 ms
 * at 256 spheres), and after `still_after` consecutive unchanged calls the full tables are built once.  moving_after = 0: every
 * change builds the full tables.  Defaults 2 and 3.  Frames are bit-identical whichever tables serve them.
 * trt_scene_is_moving: 1 while the default context treats its scene as moving. */
int trt_set_scene_policy(int moving_after, int still_after);
int trt_scene_is_moving(void);

/* Default-context management for the two calls above.  trt_init is optional (device 0 otherwise). */
int trt_init(int device);
int trt_shutdown(void);

/* Pre-upload a cubemap for the default context (what load_skybox, TRT.c:388, produced).  The
 * default context otherwise uploads on first use and re-uploads only when the face pointers,
 * the dimension or trt_invalidate_skybox() say the texels changed. */
int trt_upload_skybox(const Skybox *skybox);
int trt_invalidate_skybox(void);

/* ---- 2. context API (device-resident) --------------------------------------------------------- */

typedef struct trt_context trt_context;

/* Rows of a W x H frame owned by one renderer, as interleaved row tiles: tiles of `tile_rows`
 * rows, this renderer owns tiles tile_first, tile_first+tile_step, ...  The renderer's
 * framebuffer is compact: owned rows in ascending order, `width` pixels each.
 * Whole frame: {W, H, H, 0, 1}. */
typedef struct
{
    int width;
    int height;
    int tile_rows;
    int tile_first;
    int tile_step;
} trt_rowset;

/* number of frame rows a rowset owns (0 on invalid input) */
int trt_rowset_rows(const trt_rowset *rows);
/* frame row of local row i, or -1 */
int trt_rowset_frame_row(const trt_rowset *rows, int local_row);

int trt_create(int device, trt_context **out);
int trt_destroy(trt_context *ctx);

/* Use an existing HIP stream (hipStream_t passed as void*; NULL = the context's own stream).
 * PyTorch callers pass torch.cuda.current_stream().cuda_stream. */
int trt_set_stream(trt_context *ctx, void *hip_stream);

/* Keep `reserved` compute units free of this context's kernels: its own stream is re-created with a CU mask and the
 * persistent kernel is sized for the remaining CUs.  The frame producer's workgroups are persistent and hold every
 * wave slot they are given until the frame ends; a collective that has to run beside them (the RCCL gather of the
 * previous frame in a multi-GPU run) would otherwise wait for a frame to drain.  Applies to the context's own stream
 * (not to one handed in with trt_set_stream); 0 removes the mask.  trt_get_stream returns the stream in use, e.g. to
 * wrap it for event synchronisation (torch.cuda.ExternalStream). */
int trt_reserve_cus(trt_context *ctx, int reserved);
int trt_get_stream(trt_context *ctx, void **hip_stream);

/* Upload everything of *scene except the camera: spheres, ground, lights, skybox texels.
 * (Scene layout TRT.c:196-208.)  Synchronous; call once per scene, not per frame. */
int trt_set_scene(trt_context *ctx, const Scene *scene);

/* `dst` renders the scene `src` was given, FROM src's tables: spheres, lights, cubemap and every candidate table of the scene stay
 * one copy per device however many contexts render it (the frame slots of a trt_dist render different cameras of one scene at the
 * same time: TRT.c:1296-1306 builds the scene once, TRT.c:1327-1339 moves the camera per frame).  Only what depends on the camera
 * -- the two tables of the eye's families -- and the per-frame buffers are dst's own.  Both contexts must be of the same device;
 * up to 8 contexts per scene, all of them driven from ONE thread (the slots a scene's sharers take and give back are not locked: a
 * trt_dist drives its frame slots from the thread that calls it).  While tables are shared, the table setters (trt_set_light_grids / _slabs, trt_set_path_grids /
 * _patches / _min_spheres) refuse on every sharer; trt_set_scene gives a context tables of its own again.  The refraction
 * extension's indices are per context (dst starts with none). */
int trt_share_scene(trt_context *dst, trt_context *src);
/* Device memory held by the scene's primitives and tables (bytes), how many contexts share them, host seconds of the last build. */
int trt_scene_info(trt_context *ctx, unsigned long long *table_bytes, int *sharers, double *build_seconds);

/* Render the owned rows into DEVICE memory: d_pixels[local_row*width + col] = 3 doubles
 * (the Screen layout of TRT.c:188-193).  Asynchronous on the context's stream.
 * camera: TRT.c:178-184, by value per frame as main() does (TRT.c:1327-1339). */
int trt_render_device(trt_context *ctx, const Camera *camera, const trt_rowset *rows, int bounce_limit,
                      int rays_per_pixel, void *d_pixels, size_t capacity_bytes);

/* (int)(c*255) per channel (TRT.c:1157-1163) on the device: 3 bytes per pixel. Asynchronous. */
int trt_quantize_device(trt_context *ctx, const void *d_pixels, size_t num_pixels, void *d_rgb8);

/* Same as trt_render_device but into HOST memory (synchronous; pinned staging inside). */
int trt_render_host(trt_context *ctx, const Camera *camera, const trt_rowset *rows, int bounce_limit,
                    int rays_per_pixel, Vector *pixels);

/* trt_render_host followed by the emitter's (int)(c*255) on the device: 3 bytes per pixel to HOST memory (synchronous). */
int trt_render_host_rgb8(trt_context *ctx, const Camera *camera, const trt_rowset *rows, int bounce_limit, int rays_per_pixel,
                         unsigned char *rgb);

int trt_synchronize(trt_context *ctx);

/* HIP-event durations (ms) of the most recent render-kernel launches on this context, newest
 * last; returns how many were written (<= max).  Synchronises the stream. */
int trt_kernel_times(trt_context *ctx, float *ms, int max);
/* The same launches split at the HIP event recorded between the two kernels of a frame: the render kernel's own
 * duration and that of the ordered mean over a pixel's samples (TRT.c:1063-1065); reduce_ms may be NULL. */
int trt_render_kernel_times(trt_context *ctx, float *render_ms, float *reduce_ms, int max);

/* Kernel selection: 0 = production kernel (persistent waves, mode-synchronous rounds over single samples, then the
 * ordered mean per pixel); 1 = reference-order kernel (one lane per pixel, loops exactly as TRT.c:966-1069, every
 * sphere tested, no culling): an independent implementation kept as the on-device parity anchor.  Both are HIP; there
 * is no CPU path. */
int trt_set_kernel(trt_context *ctx, int which);

/* Light-space candidate masks of the production kernel (csrc/trt_lightgrid.h): a shadow ray reads the spheres it can
 * touch from a table of its light -- a directional_cells^2 grid across a directional light's direction, a cube map
 * of 6 * point_cells^2 direction cells about a point light -- instead of sweeping all spheres.  Built on the host
 * whenever the spheres or the lights change.  0, 0 turns the tables off (every shadow ray sweeps); results are
 * bit-identical either way.  Defaults: 128 and 64; environment TRT_LIGHTGRID="d,p" overrides the defaults. */
int trt_set_light_grids(trt_context *ctx, int directional_cells, int point_cells);
/* The tables' third coordinate (csrc/trt_lightgrid.h (5)): `directional_slabs` slabs of depth along a directional light's
 * direction, `point_shells` shells of distance from a point light (1..64 each); a cell then lists only the spheres that can lie
 * between an origin of its slab and the light -- about half the column in a dense scene.  1, 1: the two-parameter tables of
 * rounds 1-3.  Defaults 16 and 16; TRT_LIGHTGRID="d,p,slabs,shells" sets all four.  Frames are bit-identical for every value. */
int trt_set_light_slabs(trt_context *ctx, int directional_slabs, int point_shells);

/* Candidate tables of the PATH rays (csrc/trt_raygrid.h): path rays come in families that pass (nearly) through one point --
 * the eye, its mirror image in the ground, a sphere, a sphere's mirror image -- and each family has a cube map of
 * 6 * cells^2 direction cells listing the spheres its rays can touch; a path ray reads ONE cell instead of sweeping all
 * spheres (TRT.c:805-828 tests every sphere).  eye_cells: per face side for the two families of the eye (rebuilt on the GPU
 * when the eye moves), sphere_cells: for the 2N families of the spheres (rebuilt when spheres or ground change).
 * 0, 0 turns them off (every path ray sweeps); frames are bit-identical either way.  Defaults 64 and 32; environment
 * TRT_PATHGRID="e,s" overrides the defaults.  Scenes with more than 256 spheres render without any candidate table. */
int trt_set_path_grids(trt_context *ctx, int eye_cells, int sphere_cells);
/* Sub-families of the spheres (csrc/trt_raygrid.h).  "Every ray that starts anywhere on sphere i" is a fat family: in a dense
 * scene (256 spheres) its direction cells list 6.5 candidates per path ray for 1.1 exact hits.  With m > 0 the surface of
 * every sphere is cut into 6 m^2 patches (a cube map of the direction centre -> origin) and each patch -- and its mirror image
 * in the ground -- gets its own family and table: apex under the middle of the patch, membership radius 0.82 / 0.52 / 0.44 /
 * 0.33 of the sphere's for m = 1..4, hence narrower cones and shorter lists (3.6 candidates at m = 2) for 6 m^2 times the
 * table memory (sphere_cells = 32, 256 spheres: 25 MB at m = 0, 604 MB at m = 2).  The membership of every ray in the family
 * it is looked up in is still checked in FP64 per ray; frames are bit-identical for every m.  m = -1 (default): 2 for scenes of
 * 128 spheres or more, else 0.  Environment TRT_PATHGRID="e,s,m" sets all three defaults. */
int trt_set_path_patches(trt_context *ctx, int m);
/* m and the patches per sphere (1, or 6 m^2) of the tables the current scene renders with; 0, 0 when the tables are off */
int trt_get_path_patches(trt_context *ctx, int *m, int *patches_per_sphere);
/* Scenes with fewer than `min_spheres` spheres keep the sweep for their path rays: below about a dozen spheres 9 VALU per
 * sphere are cheaper than a look-up with its membership test (default 12; 0 = tables for every scene). */
int trt_set_path_grids_min_spheres(trt_context *ctx, int min_spheres);

/* Shading decoupled from the lane that owns the sample (render_rounds_kernel<.., .., true>): the hits of a
 * wave become tasks in a ring in LDS and are shaded 64 at a time, whichever lanes they came from, so that the shadow rays run
 * with ~96 % of the lanes busy instead of ~61 % (the share of path rays that hit something).  The ring costs about what one
 * light's idle lanes cost, and its 1024-thread workgroups suit large launches: mode -1 (default) uses it for scenes with two
 * lights or more and launches of 16 M samples or more (a 1080p frame at 10 rays per pixel is 20.7 M; row shards of 1/2 and
 * less stay on the plain rounds), when the rings fit in LDS beside the scene without costing a resident wave (up to ~100
 * spheres); 0: never; 1: whenever the rings fit.  Frames are bit-identical in every mode.  Environment TRT_COMPACTION=-1|0|1
 * sets the default of new contexts. */
int trt_set_compaction(trt_context *ctx, int mode);
/* Which form a frame of the current scene and of the most recent launch's size runs (a whole large frame before the first
 * launch): *decoupled 1 / 0, and the threads of a workgroup (1024 / 256). */
int trt_render_variant(trt_context *ctx, int *decoupled, int *workgroup_threads);

/* EXTENSION, PARITY UNPINNED.  The reference has no refraction (Material is {color, reflectivity, specularity},
 * TRT.c:114-119, and only the near root of a sphere is ever used, TRT.c:657); BASELINE config 3 names "refractive
 * materials" all the same.  ior[i] > 0 makes sphere i of the current scene a refractor with that index of refraction
 * (relative to the outside), 0 leaves it the reference's opaque sphere; the Material layout is untouched.  A path ray that
 * hits a refractor from outside is shaded like any hit (lighting, weight *= reflectivity, one bounce) and continues along
 * the refracted direction; inside, the sphere is met at its far root; leaving it adds no colour and no weight but costs a
 * bounce; total internal reflection mirrors the ray; shadow rays are the reference's.  count must equal the scene's
 * sphere count when a frame is rendered; count = 0 turns the extension off.  Frames are checked bit for bit against
 * oracle/trt_oracle.c's restatement of these very semantics -- not against the reference, which has none.  With the
 * extension off (the default) a different kernel instantiation runs and nothing of this is on the path. */
int trt_set_refraction(trt_context *ctx, const double *ior, int count);

/* Resource usage of the render kernel the next frame runs (hipFuncGetAttributes / occupancy query; max_blocks_per_cu counts
 * workgroups of trt_render_variant's size). */
int trt_kernel_info(trt_context *ctx, int *vgprs, int *sgprs, int *static_lds_bytes, int *max_blocks_per_cu,
                    int *compute_units);

const char *trt_last_error(void);
const char *trt_version(void);

/* ---- 3. one frame over the GPUs of a node (csrc/trt_dist.hip) --------------------------------- */

/* The reference renders a frame with one call, project_scene (TRT.c:966), from its single call site TRT.c:1339; every
 * pixel is independent.  Here one rank = one process (or thread) = one GPU; the ranks render interleaved tiles of
 * `tile_rows` rows each (tile t -> rank t mod world) and ONE gather per frame brings the rows to rank 0 over RCCL
 * (ncclSend / ncclRecv inside one group on the library's own stream; over xGMI every peer has its own link to the root).
 * Frames are pipelined over `frames_in_flight` renderer contexts.  RCCL is loaded at run time (librccl.so.1) and only for
 * world > 1.  (Test hook, include/trt_hip_diag.h: a process that asks for it before its first use of RCCL binds the library the
 * environment variable TRT_RCCL_LIB names instead -- the tests' stand-in, which lets several ranks share one GPU; every other
 * process ignores the variable.)
 * The host program carries the 128-byte id from rank 0 to the other ranks however it likes (MPI, a file, a socket,
 * torch.distributed): it is what ncclGetUniqueId produced. */
typedef struct trt_dist trt_dist;
#define TRT_DIST_ID_BYTES 128

int trt_dist_unique_id(void *id_out); /* rank 0 */
/* The library this process bound for the collectives ("" before the first use; "STAND-IN (TRT_RCCL_LIB): <path>" when the test
 * hook of trt_hip_diag.h took effect). */
const char *trt_dist_rccl_library(void);

/* Collective over the `world` ranks (same id, world, frame size, tile_rows everywhere).  scene: as for trt_set_scene (host
 * pointers inside; every rank holds the whole scene, ~1.5 MB with a 256^2 cubemap).  reserved_cus: compute units the
 * renderers leave to the collective's kernels (trt_reserve_cus; 0 = none).  id may be NULL when world == 1 (RCCL is then
 * not touched at all; with an id a one-rank communicator is created and the gather path runs with nobody to receive from). */
int trt_dist_create(int device, const Scene *scene, const void *id, int rank, int world, int width, int height, int tile_rows,
                    int frames_in_flight, int reserved_cus, trt_dist **out);
int trt_dist_set_scene(trt_dist *d, const Scene *scene);

/* Collective, asynchronous: this rank's rows of the frame seen from `camera` (TRT.c:1327-1339: a new camera per frame),
 * then the gather.  On rank 0 *d_frame is the DEVICE address of the assembled frame, height x width x 3 doubles in the
 * Screen layout of TRT.c:188-193 (NULL on the other ranks); it is complete after trt_dist_synchronize and is reused
 * `frames_in_flight` calls later. */
int trt_dist_render(trt_dist *d, const Camera *camera, int bounce_limit, int rays_per_pixel, void **d_frame);
/* If a collective fails on a rank (a HIP or RCCL error after the call has started to enqueue work), that rank is out of step
 * with its peers: the trt_dist is poisoned, every later render / synchronize call on it fails with TRT_ERR_NOT_INITIALISED,
 * and the host should destroy it on every rank. */
int trt_dist_synchronize(trt_dist *d);
/* The same frame as the 3 bytes per pixel the emitter makes of it ((int)(c*255), TRT.c:1157-1163): every rank
 * quantises its rows on the device and the gather moves bytes -- 8 times less over xGMI than doubles -- into an assembled
 * height x width x 3 byte frame on rank 0, bit-exact for buffered_draw_screen.  trt_dist_enable_rgb8 allocates the byte
 * buffers (once, on every rank, before the first trt_dist_render_rgb8); the two kinds of frame may be mixed. */
int trt_dist_enable_rgb8(trt_dist *d);
int trt_dist_render_rgb8(trt_dist *d, const Camera *camera, int bounce_limit, int rays_per_pixel, void **d_frame_rgb8);
int trt_dist_fetch_rgb8(trt_dist *d, const void *d_frame_rgb8, unsigned char *rgb);
/* synchronise, then copy an assembled frame into screen->pixels-like host memory (width*height Vectors) */
int trt_dist_fetch(trt_dist *d, const void *d_frame, Vector *pixels);
/* The root's assembly map, pure host arithmetic (no GPU): source_row[frame row] = row of the rank-major gather buffer in
 * which rank r's shard starts at row r * max_rows.  Returns max_rows, the height every shard is padded to. */
int trt_dist_source_rows(int width, int height, int tile_rows, int world, int *source_row);
int trt_dist_info(const trt_dist *d, int *rank, int *world, int *local_rows, int *max_rows, int *frames_in_flight);
/* How many ranks the communicator that was created has, as RCCL itself reports it (ncclCommCount; trt_dist_create fails unless
 * it equals `world`): the record that "RCCL saw N ranks".  0 when no communicator exists (world == 1 without an id). */
int trt_dist_comm_ranks(trt_dist *d);
/* Diagnostics, averaged over the frame slots' most recent frames (synchronises): this rank's render time (render kernel + ordered
 * mean) and the time its part of the gather took on the communicator's stream (send, or the receives and the assembly kernel on
 * the root); 0 where there was nothing to measure.  One number pair per rank makes a multi-GPU run diagnosable from one line. */
int trt_dist_frame_times(trt_dist *d, float *render_ms, float *gather_ms);
/* the renderer context of a slot (counters, kernel times, kernel selection); owned by d */
trt_context *trt_dist_context(trt_dist *d, int slot);
int trt_dist_destroy(trt_dist *d);
const char *trt_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
