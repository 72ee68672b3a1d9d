/* trt_ppm.c -- binary PPM (P6) reader and cubemap loader: what read_ppm / load_skybox /
 * free_skybox (TRT.c:309-436) give the frame producer, with status codes instead of exit(1).
 * Accepts what the reference accepts (GIMP-style '#' comment lines after the magic number,
 * maxval 255 only) and, like any PNM reader, comments anywhere in the header. */
#include <stdlib.h>
#include <string.h>

#include "trt_host.h"

/* next header token as a non-negative integer; skips whitespace and '#' comments */
static int header_int(FILE *fp, int *value)
{
    int ch = fgetc(fp);
    for (;;)
    {
        while (ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n')
            ch = fgetc(fp);
        if (ch != '#')
            break;
        while (ch != '\n' && ch != EOF)
            ch = fgetc(fp);
    }
    if (ch < '0' || ch > '9')
        return 0;
    long v = 0;
    while (ch >= '0' && ch <= '9')
    {
        v = v * 10 + (ch - '0');
        if (v > 1 << 24)
            return 0;
        ch = fgetc(fp);
    }
    /* exactly one whitespace character separates a token from what follows (TRT.c:343, :348) */
    if (ch != ' ' && ch != '\t' && ch != '\r' && ch != '\n')
        return 0;
    *value = (int)v;
    return 1;
}

int trt_read_ppm(const char *filename, Color **colors, int *width, int *height)
{
    if (!filename || !colors || !width || !height)
        return TRT_HOST_ERR_ARGUMENT;
    *colors = NULL;
    FILE *fp = fopen(filename, "rb");
    if (!fp)
        return TRT_HOST_ERR_OPEN;
    int status = TRT_HOST_OK, maxval = 0;
    if (fgetc(fp) != 'P' || fgetc(fp) != '6')
        status = TRT_HOST_ERR_FORMAT;
    else if (!header_int(fp, width) || !header_int(fp, height) || !header_int(fp, &maxval) || *width <= 0 || *height <= 0)
        status = TRT_HOST_ERR_FORMAT;
    else if (maxval != 255)
        status = TRT_HOST_ERR_MAXVAL;
    if (status == TRT_HOST_OK)
    {
        const size_t count = (size_t)*width * (size_t)*height;
        /* a header that promises more texels than the file holds is refused before anything of that size is allocated
         * (a seekable file knows its length; a pipe is read until it ends) */
        int short_file = 0;
        const long at = ftell(fp);
        if (at >= 0 && fseek(fp, 0, SEEK_END) == 0)
        {
            const long end = ftell(fp);
            short_file = end >= at && (unsigned long long)(end - at) < (unsigned long long)count * sizeof(Color);
            if (fseek(fp, at, SEEK_SET) != 0)
                short_file = 1;
        }
        Color *texels = short_file ? NULL : (Color *)malloc(sizeof(Color) * count);
        if (short_file)
            status = TRT_HOST_ERR_TRUNCATED;
        else if (!texels)
            status = TRT_HOST_ERR_MEMORY;
        else if (fread(texels, sizeof(Color), count, fp) != count) /* Color is 3 packed bytes r,g,b */
        {
            free(texels);
            status = TRT_HOST_ERR_TRUNCATED;
        }
        else
            *colors = texels;
    }
    fclose(fp);
    return status;
}

void trt_free_skybox(Skybox *skybox)
{
    if (!skybox)
        return;
    for (int f = 0; f < 6; f++)
    {
        free(skybox->colors[f]);
        skybox->colors[f] = NULL;
    }
    skybox->dim = -1;
}

int trt_load_skybox(Skybox *skybox, const char *directory)
{
    static const char *const faces[6] = {"+X.ppm", "-X.ppm", "+Y.ppm", "-Y.ppm", "+Z.ppm", "-Z.ppm"}; /* TRT.c:390 */
    if (!skybox || !directory)
        return TRT_HOST_ERR_ARGUMENT;
    for (int f = 0; f < 6; f++)
        skybox->colors[f] = NULL;
    skybox->dim = -1;
    const size_t len = strlen(directory) + 1 + 6 + 1;
    char *path = (char *)malloc(len);
    if (!path)
        return TRT_HOST_ERR_MEMORY;
    int status = TRT_HOST_OK, dim = -1;
    for (int f = 0; f < 6 && status == TRT_HOST_OK; f++)
    {
        int w = 0, h = 0;
        snprintf(path, len, "%s/%s", directory, faces[f]);
        status = trt_read_ppm(path, &skybox->colors[f], &w, &h);
        if (status == TRT_HOST_OK)
        {
            if (dim == -1)
                dim = w;
            if (dim != w || dim != h) /* TRT.c:411-417 */
                status = TRT_HOST_ERR_SHAPE;
        }
    }
    free(path);
    if (status != TRT_HOST_OK)
    {
        trt_free_skybox(skybox);
        return status;
    }
    skybox->dim = dim;
    return TRT_HOST_OK;
}
