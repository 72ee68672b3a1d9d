/*
 * trt.h -- data layout at the drop-in boundary of the MI355X frame producer.
 *
 * These are the value types that cross the boundary of the reference's frame
 * producer `void project_scene(Scene*, Screen*)` (TerminalRayTracer.c:966).
 * The reference has no FFI or plugin registry: its "interface" is that one C
 * function plus the struct definitions at TerminalRayTracer.c:61-208, so the
 * names, member order and member types below are kept field-for-field; the
 * reference's own main() (TerminalRayTracer.c:1235) links against this header
 * unchanged.  x86-64 SysV sizes/offsets are pinned by the static asserts at
 * the bottom.
 *
 * All arithmetic on the path is IEEE-754 binary64.
 */
#ifndef TRT_H
#define TRT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* TerminalRayTracer.c:61-67 -- what a traced ray ended on */
typedef enum
{
    NONE,
    SPHERE,
    GROUND,
} ObjectType;

/* TerminalRayTracer.c:70-75 */
typedef struct
{
    double x;
    double y;
    double z;
} Point;

/* TerminalRayTracer.c:78-83 */
typedef struct
{
    double x;
    double y;
    double z;
} Vector;

/* TerminalRayTracer.c:92-97 -- the three axes of a reference frame */
typedef struct
{
    Vector x;
    Vector y;
    Vector z;
} Basis;

/* TerminalRayTracer.c:100-104 */
typedef struct
{
    Basis basis;
    Point origin;
} Frame;

/* TerminalRayTracer.c:107-111 */
typedef struct
{
    Point origin;
    Vector direction;
} Ray;

/* TerminalRayTracer.c:114-119 -- specularity is carried but never read by live code */
typedef struct
{
    Vector color;
    double reflectivity;
    double specularity;
} Material;

/* TerminalRayTracer.c:122-127 */
typedef struct
{
    unsigned char r;
    unsigned char g;
    unsigned char b;
} Color;

/* TerminalRayTracer.c:130-134 -- cubemap, face order +X,-X,+Y,-Y,+Z,-Z; each face dim*dim texels, row-major */
typedef struct
{
    Color *colors[6];
    int dim;
} Skybox;

/* TerminalRayTracer.c:146-150 */
typedef struct
{
    Vector direction;
    Vector color;
} DirectionalLight;

/* TerminalRayTracer.c:153-158 */
typedef struct
{
    Point position;
    Vector color;
    double intensity;
} PointLight;

/* TerminalRayTracer.c:161-166 */
typedef struct
{
    Point center;
    double radius;
    Material material;
} Sphere;

/* TerminalRayTracer.c:169-175 -- checkerboard ground plane */
typedef struct
{
    Point point;
    Vector normal;
    Material even_material;
    Material odd_material;
} Plane;

/* TerminalRayTracer.c:178-184 */
typedef struct
{
    Frame frame;
    double screen_distance;
    double screen_width;
    double screen_height;
} Camera;

/* TerminalRayTracer.c:188-193 -- pixels[row * width + column], one Vector (r,g,b in [0,1]) per pixel */
typedef struct
{
    Vector *pixels;
    int width;
    int height;
} Screen;

/* TerminalRayTracer.c:196-208 */
typedef struct
{
    Sphere *spheres;
    int num_spheres;
    Plane ground;
    DirectionalLight *directional_lights;
    int num_directional_lights;
    PointLight *point_lights;
    int num_point_lights;
    Camera camera;
    Skybox skybox;
} Scene;

/* compile-time constants of the reference's frame producer (TerminalRayTracer.c:54, 58);
 * project_scene() uses them, the extended entries take them at run time */
#define TRT_REF_BOUNCE_LIMIT 10
#define TRT_REF_RAYS_PER_PIXEL 10

#ifdef __cplusplus
}
#define TRT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define TRT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

TRT_STATIC_ASSERT(sizeof(Vector) == 24 && sizeof(Point) == 24, "Vector/Point are 3 doubles");
TRT_STATIC_ASSERT(sizeof(Basis) == 72 && sizeof(Frame) == 96 && sizeof(Ray) == 48, "frame types");
TRT_STATIC_ASSERT(sizeof(Material) == 40 && sizeof(Color) == 3, "material/color");
TRT_STATIC_ASSERT(sizeof(Skybox) == 56 && offsetof(Skybox, dim) == 48, "Skybox");
TRT_STATIC_ASSERT(sizeof(DirectionalLight) == 48 && sizeof(PointLight) == 56, "lights");
TRT_STATIC_ASSERT(sizeof(Sphere) == 72 && offsetof(Sphere, radius) == 24 && offsetof(Sphere, material) == 32, "Sphere");
TRT_STATIC_ASSERT(sizeof(Plane) == 128 && offsetof(Plane, even_material) == 48 && offsetof(Plane, odd_material) == 88, "Plane");
TRT_STATIC_ASSERT(sizeof(Camera) == 120 && sizeof(Screen) == 16, "Camera/Screen");
TRT_STATIC_ASSERT(sizeof(Scene) == 352 && offsetof(Scene, ground) == 16 && offsetof(Scene, directional_lights) == 144 &&
                      offsetof(Scene, point_lights) == 160 && offsetof(Scene, camera) == 176 && offsetof(Scene, skybox) == 296,
                  "Scene");

#endif /* TRT_H */
