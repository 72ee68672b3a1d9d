/* mt_drop_in.c -- test helper: project_scene() from four threads at once (the drop-in layer shares one context and takes
 * turns); compiled and run by tests/test_gpu_parity.py. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "trt.h"
#include "trt_hip.h"
#include "trt_host.h"
static Scene scene; static int W = 96, H = 54;
static void *work(void *arg) { Screen s = {malloc(sizeof(Vector) * W * H), W, H}; for (int i = 0; i < 5; i++) project_scene(&scene, &s); *(Vector **)arg = s.pixels; return NULL; }
int main(int argc, char **argv) {
    static Sphere sp[2] = {{{0, 0, 0}, 1.0, {{1, 0, 0}, 0.3, 100}}, {{2, 0.5, -1}, 0.7, {{0, 1, 0}, 0.8, 100}}};
    static DirectionalLight dl = {{-1, -1, -1}, {1, 1, 1}};
    static Color tex[6][16]; Skybox sky; for (int f = 0; f < 6; f++) { for (int i = 0; i < 16; i++) tex[f][i] = (Color){(unsigned char)(40 * f), 100, (unsigned char)(10 * i)}; sky.colors[f] = tex[f]; } sky.dim = 4;
    memset(&scene, 0, sizeof scene); scene.spheres = sp; scene.num_spheres = 2; scene.directional_lights = &dl; scene.num_directional_lights = 1;
    scene.ground.point = (Point){0, -2, 0}; scene.ground.normal = (Vector){0, 1, 0}; scene.ground.even_material = (Material){{1, 1, 1}, 0.2, 100}; scene.ground.odd_material = (Material){{0, 0, 0}, 0.2, 100};
    scene.skybox = sky; trt_init_camera(&scene.camera, W, H); trt_orbit_camera(&scene.camera, 1.0);
    pthread_t t[4]; Vector *out[4];
    for (int i = 0; i < 4; i++) pthread_create(&t[i], NULL, work, &out[i]);
    for (int i = 0; i < 4; i++) pthread_join(t[i], NULL);
    int same = 1; for (int i = 1; i < 4; i++) same &= memcmp(out[0], out[i], sizeof(Vector) * W * H) == 0;
    printf("4 threads x 5 project_scene calls: frames %s\n", same ? "identical" : "DIFFER"); return !same;
}
