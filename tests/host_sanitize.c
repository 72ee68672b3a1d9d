/* host_sanitize.c -- the host-side C of include/trt_host.h (csrc/host/: camera orbit, PPM / cubemap loader, ANSI emitter, fingerprint)
 * driven through good, malformed and hostile inputs; built by tests/test_host.py with -fsanitize=address,undefined so that an
 * out-of-bounds access, a leak on an error path or undefined arithmetic ends the run.  usage: host_sanitize <scratch directory>
 * Prints "ok <checks>" and exits 0 when every status code is the expected one. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "trt_host.h"

static int checks = 0, failures = 0;
#define EXPECT(cond)                                                              \
    do                                                                            \
    {                                                                             \
        checks++;                                                                 \
        if (!(cond))                                                              \
        {                                                                         \
            failures++;                                                           \
            fprintf(stderr, "host_sanitize.c:%d: %s\n", __LINE__, #cond);         \
        }                                                                         \
    } while (0)

static void write_file(const char *path, const void *bytes, size_t n)
{
    FILE *fp = fopen(path, "wb");
    if (!fp || fwrite(bytes, 1, n, fp) != n)
    {
        perror(path);
        exit(2);
    }
    fclose(fp);
}

/* header text + `texels` data bytes (a ramp) */
static void write_ppm(const char *path, const char *header, size_t data_bytes)
{
    const size_t h = strlen(header);
    unsigned char *buf = (unsigned char *)malloc(h + data_bytes + 1);
    memcpy(buf, header, h);
    for (size_t i = 0; i < data_bytes; i++)
        buf[h + i] = (unsigned char)(i * 7 + 3);
    write_file(path, buf, h + data_bytes);
    free(buf);
}

static int read_status(const char *path, int *w, int *h, Color **c)
{
    *w = *h = -7;
    *c = (Color *)(size_t)0x10; /* must be overwritten: NULL on every failure */
    return trt_read_ppm(path, c, w, h);
}

int main(int argc, char **argv)
{
    if (argc < 2)
        return 2;
    char path[4096], dir[4096];
    Color *c;
    int w, h;

    /* ---- PPM reader (TRT.c:309-380) ---- */
    snprintf(path, sizeof path, "%s/good.ppm", argv[1]);
    write_ppm(path, "P6\n4 3\n255\n", 36);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_OK && w == 4 && h == 3 && c && c[0].r == 3 && c[11].b == (unsigned char)(35 * 7 + 3));
    free(c);
    write_ppm(path, "P6\n# made by GIMP\n4 3\n# another\n255\n", 36); /* comment lines as GIMP writes them, and elsewhere */
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_OK && w == 4 && h == 3);
    free(c);
    write_ppm(path, "P6 2\t2\r255\n", 12 + 5); /* any single whitespace separates; trailing bytes are ignored */
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_OK && w == 2 && h == 2);
    free(c);
    write_ppm(path, "P6\n4 3\n255\n", 35); /* one byte short */
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_TRUNCATED && c == NULL);
    write_ppm(path, "P6\n4 3\n255\n", 0);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_TRUNCATED && c == NULL);
    write_ppm(path, "P5\n4 3\n255\n", 36);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_FORMAT && c == NULL);
    write_ppm(path, "", 0);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_FORMAT && c == NULL);
    write_ppm(path, "P", 0);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_FORMAT && c == NULL);
    write_ppm(path, "P6", 0);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_FORMAT && c == NULL);
    write_ppm(path, "P6\n# a comment that never ends", 0);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_FORMAT && c == NULL);
    write_ppm(path, "P6\n4 3\n65535\n", 72);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_MAXVAL && c == NULL);
    write_ppm(path, "P6\n4 3\n254\n", 36);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_MAXVAL && c == NULL);
    write_ppm(path, "P6\n0 3\n255\n", 0);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_FORMAT && c == NULL);
    write_ppm(path, "P6\n-4 3\n255\n", 36);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_FORMAT && c == NULL);
    write_ppm(path, "P6\n4x 3\n255\n", 36);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_FORMAT && c == NULL);
    write_ppm(path, "P6\n99999999999999999999 3\n255\n", 36); /* a number no int holds */
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_FORMAT && c == NULL);
    write_ppm(path, "P6\n16777216 16777216\n255\n", 36); /* 2^48 texels: no such allocation, or a short read */
    {
        const int s = read_status(path, &w, &h, &c);
        EXPECT((s == TRT_HOST_ERR_MEMORY || s == TRT_HOST_ERR_TRUNCATED) && c == NULL);
    }
    write_ppm(path, "P6\n4 3\n255", 0); /* the header ends without its separator */
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_FORMAT && c == NULL);
    snprintf(path, sizeof path, "%s/absent.ppm", argv[1]);
    EXPECT(read_status(path, &w, &h, &c) == TRT_HOST_ERR_OPEN && c == NULL);
    EXPECT(trt_read_ppm(NULL, &c, &w, &h) == TRT_HOST_ERR_ARGUMENT);
    EXPECT(trt_read_ppm(path, NULL, &w, &h) == TRT_HOST_ERR_ARGUMENT);

    /* ---- cubemap loader (TRT.c:388-427): six square faces of one size ---- */
    static const char *const faces[6] = {"+X.ppm", "-X.ppm", "+Y.ppm", "-Y.ppm", "+Z.ppm", "-Z.ppm"};
    Skybox sky;
    for (int variant = 0; variant < 4; variant++)
    { /* 0: fine | 1: the last face of another size | 2: a face that is not square | 3: the fourth face missing */
        snprintf(dir, sizeof dir, "%s/sky%d", argv[1], variant);
        char cmd[4200];
        snprintf(cmd, sizeof cmd, "mkdir -p '%s'", dir);
        if (system(cmd) != 0)
            return 2;
        for (int f = 0; f < 6; f++)
        {
            snprintf(path, sizeof path, "%s/%s", dir, faces[f]);
            if (variant == 3 && f == 3)
            {
                remove(path);
                continue;
            }
            if (variant == 1 && f == 5)
                write_ppm(path, "P6\n4 4\n255\n", 48);
            else if (variant == 2 && f == 2)
                write_ppm(path, "P6\n8 4\n255\n", 96);
            else
                write_ppm(path, "P6\n8 8\n255\n", 192);
        }
        const int s = trt_load_skybox(&sky, dir);
        if (variant == 0)
        {
            EXPECT(s == TRT_HOST_OK && sky.dim == 8 && sky.colors[5] && sky.colors[5][63].g == (unsigned char)((63 * 3 + 1) * 7 + 3));
            trt_free_skybox(&sky);
            EXPECT(sky.dim == -1 && sky.colors[0] == NULL);
            trt_free_skybox(&sky); /* twice is harmless */
        }
        else
        { /* a failure frees what was loaded and leaves an empty skybox */
            EXPECT(s == (variant == 3 ? TRT_HOST_ERR_OPEN : TRT_HOST_ERR_SHAPE));
            EXPECT(sky.dim == -1);
            for (int f = 0; f < 6; f++)
                EXPECT(sky.colors[f] == NULL);
        }
    }
    EXPECT(trt_load_skybox(NULL, dir) == TRT_HOST_ERR_ARGUMENT && trt_load_skybox(&sky, NULL) == TRT_HOST_ERR_ARGUMENT);
    trt_free_skybox(NULL);

    /* ---- emitter (TRT.c:1107-1172) ---- */
    trt_emitter *e = NULL;
    EXPECT(trt_emitter_create(0, 4, &e) != TRT_HOST_OK && e == NULL);
    EXPECT(trt_emitter_create(4, -1, &e) != TRT_HOST_OK && e == NULL);
    EXPECT(trt_emitter_create(4, 4, NULL) != TRT_HOST_OK);
    EXPECT(trt_emitter_create(7, 5, &e) == TRT_HOST_OK && e != NULL);
    if (e)
    {
        const size_t size = trt_emitter_size(e);
        EXPECT(size > 7u * 5u * 10u && trt_emitter_buffer(e) != NULL);
        Vector px[35];
        for (int i = 0; i < 35; i++)
            px[i].x = i / 35.0, px[i].y = 1.0, px[i].z = 0.0; /* in [0, 1] as project_scene leaves them (TRT.c:960-962, :1065) */
        Screen screen = {px, 7, 5}, wrong = {px, 5, 7};
        EXPECT(trt_emitter_patch(e, &screen) == TRT_HOST_OK);
        EXPECT(trt_emitter_patch(e, &wrong) == TRT_HOST_ERR_ARGUMENT && trt_emitter_patch(e, NULL) == TRT_HOST_ERR_ARGUMENT);
        unsigned char rgb[105];
        for (int i = 0; i < 105; i++)
            rgb[i] = (unsigned char)(i * 5);
        rgb[0] = 0, rgb[1] = 255, rgb[2] = 9, rgb[3] = 10, rgb[4] = 99, rgb[5] = 100; /* one, two and three digits */
        const unsigned long long before = trt_fnv1a64(trt_emitter_buffer(e), size);
        EXPECT(trt_emitter_patch_rgb8(e, rgb) == TRT_HOST_OK && trt_emitter_patch_rgb8(e, NULL) == TRT_HOST_ERR_ARGUMENT);
        EXPECT(trt_emitter_size(e) == size && trt_fnv1a64(trt_emitter_buffer(e), size) != before);
        snprintf(path, sizeof path, "%s/frame.ansi", argv[1]);
        FILE *out = fopen(path, "wb");
        EXPECT(out && trt_emitter_write(e, out) == TRT_HOST_OK);
        EXPECT(out && trt_draw_screen(&screen, out) == TRT_HOST_OK);
        if (out)
            fclose(out);
        EXPECT(trt_emitter_write(e, NULL) != TRT_HOST_OK);
        trt_emitter_destroy(e);
    }
    trt_emitter_destroy(NULL);

    /* ---- camera (TRT.c:290-306, 558-624, 1327-1336) ---- */
    Camera cam;
    trt_init_camera(&cam, 480, 280);
    EXPECT(cam.screen_height == 5.0 && fabs(cam.screen_width - 5.0 * 480 / 280) < 1e-12);
    for (int i = 0; i < 200; i++)
    {
        trt_orbit_camera(&cam, i * 0.37 - 20.0);
        const Basis *b = &cam.frame.basis;
        const double lx = b->x.x * b->x.x + b->x.y * b->x.y + b->x.z * b->x.z, xy = b->x.x * b->y.x + b->x.y * b->y.y + b->x.z * b->y.z;
        EXPECT(fabs(lx - 1.0) < 1e-9 && fabs(xy) < 1e-9);
    }
    trt_orbit_camera(&cam, 1e300); /* sin / cos of anything finite are finite */
    EXPECT(isfinite(cam.frame.origin.x) && isfinite(cam.frame.basis.z.z));
    Frame fr, tr;
    trt_init_frame(&fr);
    trt_init_frame(&tr);
    trt_rotate_basis_x(&tr.basis, 0.3), trt_rotate_basis_y(&tr.basis, -1.1), trt_rotate_basis_z(&tr.basis, 2.0);
    tr.origin.x = 1, tr.origin.y = 2, tr.origin.z = 3;
    trt_transform_frame(&fr, &tr);
    trt_transform_frame(&fr, &fr); /* a frame transformed by itself: the arguments alias */
    EXPECT(isfinite(fr.origin.x) && isfinite(fr.basis.y.y));
    trt_rotate_basis(&fr.basis, &fr.basis);
    EXPECT(isfinite(fr.basis.x.x));

    EXPECT(trt_fnv1a64("", 0) == 1469598103934665603ull && trt_fnv1a64(NULL, 0) == 1469598103934665603ull);
    if (failures)
        return 1;
    printf("ok %d\n", checks);
    return 0;
}
