/* trt_emit.c -- terminal emitter: framebuffer -> 24-bit ANSI background-colour cells, two spaces
 * per pixel (TRT.c:1084-1172).  The emitter stays on the host; it only fixes the integer
 * quantisation (int)(c*255) that the GPU frame must reproduce bit-exactly. */
#include <stdlib.h>
#include <string.h>

#include "trt_host.h"

static const char k_home[] = "\033[0;0H";                      /* reset_str, TRT.c:1102 */
static const char k_cell[] = "\033[48;2;000;000;000m  \033[0m"; /* pixel_str, TRT.c:1103 */
enum
{
    HOME_LEN = sizeof(k_home) - 1,  /* 6 */
    CELL_LEN = sizeof(k_cell) - 1,  /* 25 */
    RED_AT = 7,                     /* "\033[48;2;" */
    GREEN_AT = 11,
    BLUE_AT = 15
};

struct trt_emitter
{
    int width, height;
    size_t size;
    char *text;
};

int trt_emitter_create(int width, int height, trt_emitter **out)
{
    if (!out || width <= 0 || height <= 0)
        return TRT_HOST_ERR_ARGUMENT;
    *out = NULL;
    trt_emitter *e = (trt_emitter *)malloc(sizeof *e);
    if (!e)
        return TRT_HOST_ERR_MEMORY;
    e->width = width;
    e->height = height;
    /* sizeof(screenbuffer), TRT.c:1104: (sizeof(reset_str)+1) + ((sizeof(pixel_str)-1)*W + 1)*H + 1 */
    e->size = (sizeof(k_home) + 1) + ((size_t)CELL_LEN * width + 1) * height + 1;
    e->text = (char *)calloc(e->size, 1); /* static storage in the reference: the tail stays NUL */
    if (!e->text)
    {
        free(e);
        return TRT_HOST_ERR_MEMORY;
    }
    char *p = e->text;
    memcpy(p, k_home, HOME_LEN);
    p += HOME_LEN;
    for (int row = 0; row < height; row++)
    {
        for (int col = 0; col < width; col++, p += CELL_LEN)
            memcpy(p, k_cell, CELL_LEN);
        *p++ = '\n';
    }
    *out = e;
    return TRT_HOST_OK;
}

void trt_emitter_destroy(trt_emitter *e)
{
    if (e)
    {
        free(e->text);
        free(e);
    }
}

const char *trt_emitter_buffer(const trt_emitter *e) { return e ? e->text : NULL; }
size_t trt_emitter_size(const trt_emitter *e) { return e ? e->size : 0; }

/* byte_to_digits, TRT.c:1134-1139 (plain int arithmetic, also for out-of-range values) */
static void three_digits(char *at, int value)
{
    at[0] = (char)(value / 100 + '0');
    at[1] = (char)((value / 10) % 10 + '0');
    at[2] = (char)(value % 10 + '0');
}

int trt_emitter_patch(trt_emitter *e, const Screen *screen)
{
    /* the reference's buffer and Screen share SCREEN_WIDTH/HEIGHT (TRT.c:47-48, :1104); here they are run-time values
     * and must agree, or the walk below would leave the buffer */
    if (!e || !screen || !screen->pixels || screen->width != e->width || screen->height != e->height)
        return TRT_HOST_ERR_ARGUMENT;
    char *row_text = e->text + HOME_LEN;
    for (int row = 0; row < screen->height; row++, row_text += (size_t)CELL_LEN * e->width + 1)
    {
        char *cell = row_text;
        for (int col = 0; col < screen->width; col++, cell += CELL_LEN)
        {
            const Vector px = screen->pixels[row * screen->width + col];
            three_digits(cell + RED_AT, (int)(px.x * 255));
            three_digits(cell + GREEN_AT, (int)(px.y * 255));
            three_digits(cell + BLUE_AT, (int)(px.z * 255));
        }
    }
    return TRT_HOST_OK;
}

int trt_emitter_patch_rgb8(trt_emitter *e, const unsigned char *rgb)
{
    if (!e || !rgb)
        return TRT_HOST_ERR_ARGUMENT;
    char *row_text = e->text + HOME_LEN;
    for (int row = 0; row < e->height; row++, row_text += (size_t)CELL_LEN * e->width + 1)
    {
        char *cell = row_text;
        for (int col = 0; col < e->width; col++, cell += CELL_LEN, rgb += 3)
        {
            three_digits(cell + RED_AT, rgb[0]);
            three_digits(cell + GREEN_AT, rgb[1]);
            three_digits(cell + BLUE_AT, rgb[2]);
        }
    }
    return TRT_HOST_OK;
}

int trt_emitter_write(const trt_emitter *e, FILE *stream)
{
    if (!e || !stream)
        return TRT_HOST_ERR_ARGUMENT;
    return fwrite(e->text, 1, e->size, stream) == e->size ? TRT_HOST_OK : TRT_HOST_ERR_OPEN;
}

int trt_draw_screen(const Screen *screen, FILE *stream)
{
    if (!screen || !screen->pixels || !stream)
        return TRT_HOST_ERR_ARGUMENT;
    if (fputs(k_home, stream) < 0)
        return TRT_HOST_ERR_OPEN;
    for (int row = 0; row < screen->height; row++)
    {
        for (int col = 0; col < screen->width; col++)
        {
            const Vector px = screen->pixels[row * screen->width + col];
            fprintf(stream, "\033[48;2;%d;%d;%dm  \033[0m", (int)(px.x * 255), (int)(px.y * 255), (int)(px.z * 255));
        }
        fputc('\n', stream);
    }
    return TRT_HOST_OK;
}
