/* trt_demo.c -- the reference's frame loop (TRT.c:1235-1370) on top of libtrt_hip.so:
 * host C builds the demo scene and the orbiting camera, the MI355X produces the frame through
 * the drop-in project_scene(), and the ANSI emitter stays on the host.
 *
 *   trt_demo <skybox-directory> [frames=0 (until Ctrl-C)] [width=160] [height=48] [--no-draw] [--rgb8]
 *
 * --rgb8: the frame crosses PCIe as the 3 bytes per pixel the emitter makes of it (trt_render_frame_rgb8: the (int)(c*255) of
 * TRT.c:1157-1163 done on the device) instead of as 24-byte doubles; what reaches the terminal is the same.
 *
 * The scene literals are the reference's (TRT.c:1256-1288). */
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "trt_hip.h"
#include "trt_host.h"

static volatile sig_atomic_t stop_requested = 0;
static void on_sigint(int sig)
{
    (void)sig;
    stop_requested = 1;
}

static double seconds_since(const struct timespec *start)
{
    struct timespec now;
    timespec_get(&now, TIME_UTC);
    return (double)(now.tv_sec - start->tv_sec) + (double)(now.tv_nsec - start->tv_nsec) / 1e9;
}

int main(int argc, char **argv)
{
    if (argc < 2)
    {
        fprintf(stderr, "usage: %s <skybox-directory> [frames] [width] [height] [--no-draw] [--rgb8]\n", argv[0]);
        return 2;
    }
    const long frames = argc > 2 ? atol(argv[2]) : 0;
    const int width = argc > 3 ? atoi(argv[3]) : 160, height = argc > 4 ? atoi(argv[4]) : 48;
    int draw = 1, bytes_only = 0;
    for (int i = 5; i < argc; i++)
    {
        if (strcmp(argv[i], "--no-draw") == 0)
            draw = 0;
        else if (strcmp(argv[i], "--rgb8") == 0)
            bytes_only = 1;
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
    scene.ground.point.y = -2.0;
    scene.ground.normal.y = 1.0;
    scene.ground.even_material = (Material){{1.0, 1.0, 1.0}, 0.2, 100.0};
    scene.ground.odd_material = (Material){{1.0, 0.0, 0.0}, 0.2, 100.0};
    scene.directional_lights = sun;
    scene.num_directional_lights = 1;
    scene.point_lights = lamp;
    scene.num_point_lights = 1;
    scene.skybox = sky;
    trt_init_camera(&scene.camera, width, height);

    Screen screen = {(Vector *)malloc(sizeof(Vector) * (size_t)width * height), width, height};
    unsigned char *rgb = (unsigned char *)malloc((size_t)width * height * 3);
    trt_emitter *emitter = NULL;
    if (!screen.pixels || !rgb || trt_emitter_create(width, height, &emitter) != TRT_HOST_OK)
        return 1;

    signal(SIGINT, on_sigint);
    struct timespec start;
    timespec_get(&start, TIME_UTC);
    double producer_seconds = 0.0, first_call_seconds = 0.0;
    long frame = 0;
    for (; !stop_requested && (frames == 0 || frame < frames); frame++)
    {
        const double t = seconds_since(&start);
        trt_orbit_camera(&scene.camera, t);

        const double before = seconds_since(&start);
        if (!bytes_only)
            project_scene(&scene, &screen); /* the GPU frame producer, same call as TRT.c:1339 */
        else if (trt_render_frame_rgb8(&scene, width, height, TRT_REF_BOUNCE_LIMIT, TRT_REF_RAYS_PER_PIXEL, rgb) != TRT_OK)
        {
            fprintf(stderr, "trt_render_frame_rgb8: %s\n", trt_last_error());
            return 1;
        }
        if (frame == 0) /* device initialisation, cubemap upload and pinned staging happen in the first call */
            first_call_seconds = seconds_since(&start) - before;
        else
            producer_seconds += seconds_since(&start) - before;

        if (draw)
        {
            if (bytes_only)
                trt_emitter_patch_rgb8(emitter, rgb);
            else
                trt_emitter_patch(emitter, &screen);
            trt_emitter_write(emitter, stdout);
            fputs("\033[0;0H", stdout);
            printf("%.02f fps\n", 1.0 / (seconds_since(&start) - t));
            fputs("\033[0;0H", stdout);
        }
    }
    fprintf(stderr, "%ld frames %dx%d, frame producer %.3f ms/frame after a first call of %.1f ms (host-in/host-out%s, 10 bounces, 10 rays per pixel)\n",
            frame, width, height, frame > 1 ? 1e3 * producer_seconds / (frame - 1) : 0.0, 1e3 * first_call_seconds,
            bytes_only ? " as RGB8" : "");

    trt_emitter_destroy(emitter);
    free(screen.pixels);
    free(rgb);
    trt_free_skybox(&sky);
    trt_shutdown();
    return 0;
}
