/* trt_dist_demo.c -- the reference's frame loop (TRT.c:1317-1366) with the frame sharded over the GPUs of a node.
 * One process per GPU; every process runs this same program:
 *
 *   trt_dist_demo <skybox-directory> <rank> <world> <id-file> [frames=60] [width=1920] [height=1080] [tile-rows=8]
 *                 [frames-in-flight=2] [device=<rank>] [rgb8=0] [allow-rccl-stand-in=0]
 *
 * Rank 0 writes the communicator id (what ncclGetUniqueId produced) to <id-file>, the other ranks wait for the file: any
 * other way of carrying 128 bytes to the ranks (MPI, a socket) does as well.  Each rank renders its interleaved row tiles
 * on GPU <rank>, one RCCL gather per frame brings them to rank 0 (all of it inside libtrt_hip.so, trt_dist_*), and rank 0
 * -- the only process that owns a terminal -- would hand the frame to the emitter exactly as the single-GPU demo does;
 * here it prints the frame's fingerprint instead.  Scene and camera are the reference's (TRT.c:1256-1288, :1327-1336).
 * With world = 1 the whole path (communicator, group, assembly kernel) runs on one GPU.  rgb8 = 1: the ranks gather the
 * frame as the 3 bytes per pixel the emitter makes of it (trt_dist_render_rgb8) and rank 0 prints that frame's fingerprint.
 * device: the GPU of this rank (several ranks on one GPU only work with the tests' stand-in for RCCL: the environment variable
 * TRT_RCCL_LIB names it, and the last argument must be 1 -- the library ignores the variable unless the process asks for the
 * override, trt_dist_allow_rccl_override). */
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "trt_hip.h"
#include "trt_hip_diag.h"
#include "trt_host.h"

static double now_seconds(void)
{
    struct timespec t;
    timespec_get(&t, TIME_UTC);
    return (double)t.tv_sec + (double)t.tv_nsec / 1e9;
}

int main(int argc, char **argv)
{
    if (argc < 5)
    {
        fprintf(stderr, "usage: %s <skybox-directory> <rank> <world> <id-file> [frames] [width] [height] [tile-rows] [frames-in-flight] [device] [rgb8]\n", argv[0]);
        return 2;
    }
    const int rank = atoi(argv[2]), world = atoi(argv[3]);
    const char *id_file = argv[4];
    const int frames = argc > 5 ? atoi(argv[5]) : 60, width = argc > 6 ? atoi(argv[6]) : 1920, height = argc > 7 ? atoi(argv[7]) : 1080;
    const int tile_rows = argc > 8 ? atoi(argv[8]) : 8, in_flight = argc > 9 ? atoi(argv[9]) : 2, device = argc > 10 ? atoi(argv[10]) : rank;
    const int rgb8 = argc > 11 ? atoi(argv[11]) : 0;
    if (argc > 12 && atoi(argv[12]) && trt_dist_allow_rccl_override(1) != TRT_OK) /* TEST HOOK: before the first use of RCCL */
    {
        fprintf(stderr, "trt_dist_allow_rccl_override: %s\n", trt_dist_last_error());
        return 1;
    }

    Skybox sky;
    int rc = trt_load_skybox(&sky, argv[1]);
    if (rc != TRT_HOST_OK)
    {
        fprintf(stderr, "cannot load skybox from %s (error %d)\n", argv[1], rc);
        return 1;
    }
    Sphere spheres[6] = {
        {{1.0, 0.0, 0.0}, 0.5, {{1.0, 0.0, 0.0}, 1.0, 100.0}},  {{0.0, 1.0, 0.0}, 0.5, {{0.0, 1.0, 0.0}, 0.8, 100.0}},
        {{0.0, 0.0, 1.0}, 0.5, {{0.0, 0.0, 1.0}, 0.8, 100.0}},  {{-1.0, 0.0, 0.0}, 0.5, {{0.0, 1.0, 1.0}, 0.8, 100.0}},
        {{0.0, -1.0, 0.0}, 0.5, {{1.0, 0.0, 1.0}, 0.8, 100.0}}, {{0.0, 0.0, -1.0}, 0.5, {{1.0, 1.0, 0.0}, 0.8, 100.0}},
    };
    DirectionalLight sun[1] = {{{-1.0, -1.0, -1.0}, {1.0, 1.0, 1.0}}};
    PointLight lamp[1] = {{{0.0, 0.0, 0.0}, {1.0, 1.0, 1.0}, 10.0}};
    Scene scene;
    memset(&scene, 0, sizeof scene);
    scene.spheres = spheres;
    scene.num_spheres = 6;
    scene.ground = (Plane){{0.0, -2.0, 0.0}, {0.0, 1.0, 0.0}, {{1.0, 1.0, 1.0}, 0.2, 100.0}, {{1.0, 0.0, 0.0}, 0.2, 100.0}};
    scene.directional_lights = sun;
    scene.num_directional_lights = 1;
    scene.point_lights = lamp;
    scene.num_point_lights = 1;
    scene.skybox = sky;
    trt_init_camera(&scene.camera, width, height);

    /* the communicator id: rank 0 makes it, everybody else picks it up */
    unsigned char id[TRT_DIST_ID_BYTES];
    if (rank == 0)
    {
        if (trt_dist_unique_id(id) != TRT_OK)
        {
            fprintf(stderr, "trt_dist_unique_id: %s\n", trt_dist_last_error());
            return 1;
        }
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
        FILE *fh = fopen(tmp, "wb");
        if (!fh || fwrite(id, 1, sizeof id, fh) != sizeof id || fclose(fh) != 0 || rename(tmp, id_file) != 0)
        {
            fprintf(stderr, "cannot write %s\n", id_file);
            return 1;
        }
    }
    else
    {
        FILE *fh = NULL;
        for (int tries = 0; tries < 600 && !(fh = fopen(id_file, "rb")); tries++)
            usleep(100000);
        if (!fh || fread(id, 1, sizeof id, fh) != sizeof id)
        {
            fprintf(stderr, "rank %d: cannot read %s\n", rank, id_file);
            return 1;
        }
        fclose(fh);
    }

    trt_dist *dist = NULL;
    if (trt_dist_create(device, &scene, id, rank, world, width, height, tile_rows, in_flight, 0, &dist) != TRT_OK ||
        (rgb8 && trt_dist_enable_rgb8(dist) != TRT_OK))
    {
        fprintf(stderr, "rank %d: trt_dist_create: %s\n", rank, trt_dist_last_error());
        return 1;
    }
    Vector *pixels = rank == 0 ? (Vector *)malloc(sizeof(Vector) * (size_t)width * height) : NULL;
    const double start = now_seconds();
    void *frame = NULL;
    for (int f = 0; f < frames; f++)
    {
        trt_orbit_camera(&scene.camera, f / 60.0); /* TRT.c:1327-1336 at a fixed 60 frames per second of scene time */
        if ((rgb8 ? trt_dist_render_rgb8(dist, &scene.camera, TRT_REF_BOUNCE_LIMIT, TRT_REF_RAYS_PER_PIXEL, &frame)
                  : trt_dist_render(dist, &scene.camera, TRT_REF_BOUNCE_LIMIT, TRT_REF_RAYS_PER_PIXEL, &frame)) != TRT_OK)
        {
            fprintf(stderr, "rank %d: trt_dist_render: %s\n", rank, trt_dist_last_error());
            return 1;
        }
    }
    if (trt_dist_synchronize(dist) != TRT_OK)
    {
        fprintf(stderr, "rank %d: %s\n", rank, trt_dist_last_error());
        return 1;
    }
    const double elapsed = now_seconds() - start;
    if (rank == 0)
    {
        const size_t bytes = rgb8 ? (size_t)width * height * 3 : sizeof(Vector) * (size_t)width * height;
        if ((rgb8 ? trt_dist_fetch_rgb8(dist, frame, (unsigned char *)pixels) : trt_dist_fetch(dist, frame, pixels)) != TRT_OK)
        {
            fprintf(stderr, "trt_dist_fetch: %s\n", trt_dist_last_error());
            return 1;
        }
        printf("%d frames %dx%d on %d GPU(s): %.3f ms/frame, last frame %sfnv %016llx (gather through %s)\n", frames, width, height, world,
               1e3 * elapsed / frames, rgb8 ? "rgb8 " : "", trt_fnv1a64(pixels, bytes), world > 1 ? trt_dist_rccl_library() : "nothing: one rank");
        free(pixels);
    }
    trt_dist_destroy(dist);
    trt_free_skybox(&sky);
    return 0;
}
