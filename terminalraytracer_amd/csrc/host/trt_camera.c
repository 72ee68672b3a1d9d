/* trt_camera.c -- the camera side of the reference's frame loop (host, once per frame).
 * Follows TRT.c:290-305 (init), :558-603 (basis rotations), :607-624 (frame transform) and
 * :1327-1336 (orbit).  Every sum is written left to right exactly as there, so that with the
 * same libm sin/cos the Camera bytes are identical. */
#include <math.h>

#include "trt_host.h"

#define TRT_PI 3.14159265358979323846 /* TRT.c:43 */

static Vector vec(double x, double y, double z)
{
    Vector v = {x, y, z};
    return v;
}

static double along(const Vector *a, const Vector *b) { return a->x * b->x + a->y * b->y + a->z * b->z; }

void trt_init_frame(Frame *frame)
{
    frame->basis.x = vec(1.0, 0.0, 0.0);
    frame->basis.y = vec(0.0, 1.0, 0.0);
    frame->basis.z = vec(0.0, 0.0, 1.0);
    frame->origin.x = frame->origin.y = frame->origin.z = 0.0;
}

void trt_init_camera(Camera *camera, int aspect_w, int aspect_h)
{
    trt_init_frame(&camera->frame);
    camera->screen_distance = 1.0;
    camera->screen_width = 5 * (double)aspect_w / (double)aspect_h;
    camera->screen_height = 5 * 1.0;
}

/* every axis of the basis is re-expressed against the three rows of the rotation */
void trt_rotate_basis(Basis *basis, const Basis *rotation)
{
    const Vector *axes[3] = {&basis->x, &basis->y, &basis->z};
    Vector out[3];
    for (int i = 0; i < 3; i++)
        out[i] = vec(along(axes[i], &rotation->x), along(axes[i], &rotation->y), along(axes[i], &rotation->z));
    basis->x = out[0];
    basis->y = out[1];
    basis->z = out[2];
}

void trt_rotate_basis_x(Basis *basis, double angle)
{
    Basis r = {vec(1.0, 0.0, 0.0), vec(0.0, cos(angle), -sin(angle)), vec(0.0, sin(angle), cos(angle))};
    trt_rotate_basis(basis, &r);
}

void trt_rotate_basis_y(Basis *basis, double angle)
{
    Basis r = {vec(cos(angle), 0.0, sin(angle)), vec(0.0, 1.0, 0.0), vec(-sin(angle), 0.0, cos(angle))};
    trt_rotate_basis(basis, &r);
}

void trt_rotate_basis_z(Basis *basis, double angle)
{
    Basis r = {vec(cos(angle), -sin(angle), 0.0), vec(sin(angle), cos(angle), 0.0), vec(0.0, 0.0, 1.0)};
    trt_rotate_basis(basis, &r);
}

/* (basis, origin) pushed through `transform` as one homogeneous 4x4 */
void trt_transform_frame(Frame *frame, const Frame *transform)
{
    const Basis *t = &transform->basis;
    const Vector *rows[3] = {&frame->basis.x, &frame->basis.y, &frame->basis.z};
    Vector out[3];
    for (int i = 0; i < 3; i++)
    {
        const Vector *a = rows[i];
        out[i] = vec(a->x * t->x.x + a->y * t->y.x + a->z * t->z.x, a->x * t->x.y + a->y * t->y.y + a->z * t->z.y,
                     a->x * t->x.z + a->y * t->y.z + a->z * t->z.z);
    }
    const Point o = frame->origin;
    Point moved = {o.x * t->x.x + o.y * t->y.x + o.z * t->z.x + transform->origin.x,
                   o.x * t->x.y + o.y * t->y.y + o.z * t->z.y + transform->origin.y,
                   o.x * t->x.z + o.y * t->y.z + o.z * t->z.z + transform->origin.z};
    frame->basis.x = out[0];
    frame->basis.y = out[1];
    frame->basis.z = out[2];
    frame->origin = moved;
}

void trt_orbit_camera(Camera *camera, double t)
{
    Frame spin, offset;
    trt_init_frame(&spin);
    trt_init_frame(&offset);
    trt_init_frame(&camera->frame);
    trt_rotate_basis_x(&spin.basis, 2.0 * TRT_PI * t * -0.03);
    trt_rotate_basis_y(&spin.basis, 2.0 * TRT_PI * t * 0.05);
    offset.origin.x += 0.0;
    offset.origin.y += 0.0;
    offset.origin.z += 1.99; /* root_to_camera, TRT.c:1333 */
    trt_transform_frame(&camera->frame, &offset);
    trt_transform_frame(&camera->frame, &spin);
}
