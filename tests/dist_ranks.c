/* dist_ranks.c -- TEST HOST for the C-ABI multi-GPU path (csrc/trt_dist.hip): one or more ranks of a `world`, each a THREAD of
 * this process, all on device 0 -- a GPU box admits six processes, so an 8-way split (BASELINE configs[3] and [4] name 8 GPUs)
 * runs as three processes: rank 0 alone, ranks 1-4, ranks 5-7.  The gather goes through the tests' stand-in for RCCL
 * (tests/rccl_stub.cpp; TRT_RCCL_LIB, allowed with trt_dist_allow_rccl_override).  Not an example and not product code: the
 * scene, the cameras and the cubemap come from a file the test writes, so that any scene of the Python side can be split.
 *
 *   dist_ranks <scene-file> <ranks, e.g. 1,2,3,4> <world> <id-file> <width> <height> <bounce-limit> <tile-rows> <frames-in-flight> <rgb8>
 *
 * scene-file: int32 {spheres, directional lights, point lights, cubemap dim, cameras}, then doubles: spheres x 9, ground 16,
 * directional lights x 6, point lights x 7, cameras x 15; then 6 x dim x dim x 3 bytes of texels.
 * Rank 0 writes the communicator id to <id-file>; it fetches every frame (the peers run on) and prints "frame <i> fnv <hash>". */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "trt_hip.h"
#include "trt_hip_diag.h"
#include "trt_host.h"

typedef struct
{
    int rank, world, width, height, bounces, tile_rows, depth, rgb8, cameras;
    const Scene *scene;
    const double *cams;
    const char *id_file;
    int result;
} job;

static void *run_rank(void *arg)
{
    job *j = (job *)arg;
    j->result = 1;
    unsigned char id[TRT_DIST_ID_BYTES];
    if (j->rank == 0)
    {
        if (trt_dist_unique_id(id) != TRT_OK)
        {
            fprintf(stderr, "rank 0: trt_dist_unique_id: %s\n", trt_dist_last_error());
            return NULL;
        }
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", j->id_file);
        FILE *fh = fopen(tmp, "wb");
        if (!fh || fwrite(id, 1, sizeof id, fh) != sizeof id || fclose(fh) != 0 || rename(tmp, j->id_file) != 0)
            return NULL;
    }
    else
    {
        FILE *fh = NULL;
        for (int tries = 0; tries < 1200 && !(fh = fopen(j->id_file, "rb")); tries++)
            usleep(100000);
        if (!fh || fread(id, 1, sizeof id, fh) != sizeof id)
        {
            fprintf(stderr, "rank %d: cannot read %s\n", j->rank, j->id_file);
            return NULL;
        }
        fclose(fh);
    }
    trt_dist *d = NULL;
    if (trt_dist_create(0, j->scene, id, j->rank, j->world, j->width, j->height, j->tile_rows, j->depth, 0, &d) != TRT_OK ||
        (j->rgb8 && trt_dist_enable_rgb8(d) != TRT_OK))
    {
        fprintf(stderr, "rank %d: trt_dist_create: %s\n", j->rank, trt_dist_last_error());
        return NULL;
    }
    const size_t bytes = j->rgb8 ? (size_t)j->width * j->height * 3 : sizeof(Vector) * (size_t)j->width * j->height;
    void *pixels = j->rank == 0 ? malloc(bytes) : NULL;
    for (int f = 0; f < j->cameras; f++)
    {
        Camera cam;
        memcpy(&cam, j->cams + 15 * f, sizeof cam);
        void *frame = NULL;
        if ((j->rgb8 ? trt_dist_render_rgb8(d, &cam, j->bounces, TRT_REF_RAYS_PER_PIXEL, &frame)
                     : trt_dist_render(d, &cam, j->bounces, TRT_REF_RAYS_PER_PIXEL, &frame)) != TRT_OK)
        {
            fprintf(stderr, "rank %d: trt_dist_render: %s\n", j->rank, trt_dist_last_error());
            return NULL;
        }
        if (j->rank == 0)
        {
            if ((j->rgb8 ? trt_dist_fetch_rgb8(d, frame, (unsigned char *)pixels) : trt_dist_fetch(d, frame, (Vector *)pixels)) != TRT_OK)
            {
                fprintf(stderr, "rank 0: trt_dist_fetch: %s\n", trt_dist_last_error());
                return NULL;
            }
            printf("frame %d fnv %016llx\n", f, (unsigned long long)trt_fnv1a64(pixels, bytes));
            fflush(stdout);
        }
    }
    if (trt_dist_synchronize(d) != TRT_OK)
    {
        fprintf(stderr, "rank %d: %s\n", j->rank, trt_dist_last_error());
        return NULL;
    }
    float render_ms = 0.0f, gather_ms = 0.0f;
    (void)trt_dist_frame_times(d, &render_ms, &gather_ms);
    fprintf(stderr, "rank %d of %d: render %.3f ms, gather %.3f ms (%s)\n", j->rank, j->world, render_ms, gather_ms, trt_dist_rccl_library());
    free(pixels);
    trt_dist_destroy(d);
    j->result = 0;
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 11)
    {
        fprintf(stderr, "usage: %s <scene-file> <ranks> <world> <id-file> <width> <height> <bounce-limit> <tile-rows> <frames-in-flight> <rgb8>\n", argv[0]);
        return 2;
    }
    if (trt_dist_allow_rccl_override(1) != TRT_OK) /* TEST HOOK: several ranks share the one GPU through the stand-in */
    {
        fprintf(stderr, "trt_dist_allow_rccl_override: %s\n", trt_dist_last_error());
        return 1;
    }
    FILE *fh = fopen(argv[1], "rb");
    int32_t head[5];
    if (!fh || fread(head, sizeof head, 1, fh) != 1)
    {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 1;
    }
    const int n = head[0], nd = head[1], np = head[2], dim = head[3], cameras = head[4];
    const size_t doubles = (size_t)n * 9 + 16 + (size_t)nd * 6 + (size_t)np * 7 + (size_t)cameras * 15, face = (size_t)dim * dim * 3;
    double *data = (double *)malloc(doubles * sizeof(double));
    unsigned char *texels = (unsigned char *)malloc(6 * face);
    if (!data || !texels || fread(data, sizeof(double), doubles, fh) != doubles || fread(texels, 1, 6 * face, fh) != 6 * face)
    {
        fprintf(stderr, "%s is short\n", argv[1]);
        return 1;
    }
    fclose(fh);
    Scene scene;
    memset(&scene, 0, sizeof scene);
    double *at = data;
    scene.spheres = (Sphere *)at, scene.num_spheres = n, at += (size_t)n * 9;
    memcpy(&scene.ground, at, sizeof(Plane)), at += 16;
    scene.directional_lights = (DirectionalLight *)at, scene.num_directional_lights = nd, at += (size_t)nd * 6;
    scene.point_lights = (PointLight *)at, scene.num_point_lights = np, at += (size_t)np * 7;
    const double *cams = at;
    for (int f = 0; f < 6; f++)
        scene.skybox.colors[f] = (Color *)(texels + (size_t)f * face);
    scene.skybox.dim = dim;
    memcpy(&scene.camera, cams, sizeof(Camera));

    job jobs[8];
    pthread_t threads[8];
    int count = 0;
    for (char *tok = strtok(argv[2], ","); tok && count < 8; tok = strtok(NULL, ","))
    {
        job *j = &jobs[count++];
        j->rank = atoi(tok), j->world = atoi(argv[3]), j->id_file = argv[4];
        j->width = atoi(argv[5]), j->height = atoi(argv[6]), j->bounces = atoi(argv[7]), j->tile_rows = atoi(argv[8]), j->depth = atoi(argv[9]);
        j->rgb8 = atoi(argv[10]), j->cameras = cameras, j->scene = &scene, j->cams = cams, j->result = 1;
    }
    for (int i = 0; i < count; i++)
        pthread_create(&threads[i], NULL, run_rank, &jobs[i]);
    int failed = 0;
    for (int i = 0; i < count; i++)
    {
        pthread_join(threads[i], NULL);
        failed |= jobs[i].result;
    }
    free(data);
    free(texels);
    return failed;
}
