/*
 * trt_oracle.c -- CPU restatement of TerminalRayTracer.c's frame producer
 * (project_scene, trace_ray, apply_lighting, get_skybox_color and the leaf
 * math they inline), with bounce limit / rays per pixel as run-time values.
 *
 * TEST INFRASTRUCTURE ONLY (see trt_oracle.h).  Build with
 *     gcc -O3 -ffp-contract=off -fno-fast-math
 * Every function cites the reference lines it follows ("TRT.c:N" =
 * /root/reference/TerminalRayTracer.c line N).  The order of every
 * floating-point operation is part of the contract: the framebuffer must be
 * bit-identical to the reference's.
 */
#include "trt_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TRT_PI 3.14159265358979323846 /* TRT.c:43 */
#define TRT_NUDGE 0.000001            /* TRT.c:44 EPSILON */

typedef struct
{
    double x, y, z;
} v3;

static inline v3 v3_of(const Vector *p) { return (v3){p->x, p->y, p->z}; }
static inline v3 v3_ofp(const Point *p) { return (v3){p->x, p->y, p->z}; }
static inline void v3_to(Vector *dst, v3 a) { dst->x = a.x, dst->y = a.y, dst->z = a.z; }
static inline void v3_top(Point *dst, v3 a) { dst->x = a.x, dst->y = a.y, dst->z = a.z; }

/* TRT.c:461-464: ((x*x)+(y*y))+(z*z), left to right */
static inline double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 sub(v3 a, v3 b) { return (v3){a.x - b.x, a.y - b.y, a.z - b.z}; }  /* TRT.c:499-504 */
static inline v3 add(v3 a, v3 b) { return (v3){a.x + b.x, a.y + b.y, a.z + b.z}; }  /* TRT.c:483-488 */
static inline v3 mulc(v3 a, v3 b) { return (v3){a.x * b.x, a.y * b.y, a.z * b.z}; } /* TRT.c:515-520 */
static inline v3 scale(v3 a, double s) { return (v3){a.x * s, a.y * s, a.z * s}; }  /* TRT.c:467-472 */

/* TRT.c:439-450: three divisions by the length, skipped when the length is <= 1e-4 */
static inline v3 unit(v3 a)
{
    double len = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    if (len > 0.0001)
    {
        a.x /= len;
        a.y /= len;
        a.z /= len;
    }
    return a;
}

/* TRT.c:523-530 */
static inline double clampd(double v, double lo, double hi)
{
    if (v < lo)
        return lo;
    if (v > hi)
        return hi;
    return v;
}

/* TRT.c:627-633: v - ((2.0*dot)*n) per component */
static inline v3 reflect(v3 v, v3 n)
{
    double d = dot(v, n);
    return (v3){v.x - 2.0 * d * n.x, v.y - 2.0 * d * n.y, v.z - 2.0 * d * n.z};
}

/* TRT.c:225-228 */
double trt_oracle_triangle_wave(double t)
{
    double m = fmod(t, 2 * TRT_PI);
    return (m < TRT_PI) ? (m / TRT_PI) : (2 - (m / TRT_PI));
}

/* TRT.c:638-672: near root only; a ray that starts inside a sphere misses it */
static inline int hit_sphere(v3 o, v3 d, const Sphere *s, v3 *p)
{
    v3 oc = {o.x - s->center.x, o.y - s->center.y, o.z - s->center.z};
    double a = dot(d, d);
    double b = 2.0 * dot(oc, d);
    double c = dot(oc, oc) - s->radius * s->radius;
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0)
        return 0;
    double t0 = (-b - sqrt(disc)) / (2.0 * a);
    if (!(t0 > 0.0))
        return 0;
    p->x = o.x + t0 * d.x;
    p->y = o.y + t0 * d.y;
    p->z = o.z + t0 * d.z;
    return 1;
}

/* TRT.c:677-695 */
static inline int hit_plane(v3 o, v3 d, const Plane *g, v3 *p)
{
    v3 n = v3_of(&g->normal);
    double denom = dot(d, n);
    if (!(fabs(denom) > 0.00001))
        return 0;
    v3 to_plane = sub(v3_ofp(&g->point), o);
    double t = dot(to_plane, n) / denom;
    if (!(t > 0.00001))
        return 0;
    p->x = o.x + t * d.x;
    p->y = o.y + t * d.y;
    p->z = o.z + t * d.z;
    return 1;
}

/* cubemap axes in face order +X,-X,+Y,-Y,+Z,-Z (TRT.c:137-143) */
static const v3 k_axes[6] = {{1.0, 0.0, 0.0}, {-1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, -1.0, 0.0}, {0.0, 0.0, 1.0}, {0.0, 0.0, -1.0}};

/* TRT.c:700-789.  Returns 1 when the texel index lies inside the face. */
int trt_oracle_skybox_lookup(const Scene *scene, const Vector *direction, int *face_out, long *texel_index)
{
    v3 dir = unit(v3_of(direction));

    /* face = argmax of dir . axis, strict >, first wins (TRT.c:703-713) */
    int face = -1;
    double best = -1.0;
    for (int f = 0; f < 6; f++)
    {
        double t = dot(dir, k_axes[f]);
        if (t > best)
        {
            best = t;
            face = f;
        }
    }

    /* scale the direction so that it touches the face plane (TRT.c:717-719) */
    v3 touching = mulc(dir, k_axes[face]);
    double scale_by = touching.x + touching.y + touching.z;
    dir = scale(dir, 1.0 / scale_by);

    /* in-plane part, halved because the faces sit at distance 0.5 (TRT.c:720-723) */
    double along = dot(dir, k_axes[face]);
    v3 in_plane = scale(sub(dir, scale(k_axes[face], along)), 0.5);

    double u = dot(in_plane, k_axes[(face + 2) % 6]); /* TRT.c:726 */
    double v = dot(in_plane, k_axes[(face + 4) % 6]); /* TRT.c:727 */

    /* per-face mirror / rotation table (TRT.c:730-761) */
    if (face % 2 == 1)
        u *= -1.0;
    if (face == 0 || face == 1)
    {
        double t = u;
        u = v;
        v = -t;
    }
    else if (face == 2 || face == 3)
    {
        double t = u;
        u = -v;
        v = t;
    }
    else if (face == 4)
    {
        u *= -1.0;
        v *= -1.0;
    }

    u = clampd(u, -0.5, 0.5); /* TRT.c:778-779 */
    v = clampd(v, -0.5, 0.5);

    int dim = scene->skybox.dim;
    int ui = (int)((u + 0.5) * dim); /* TRT.c:782-783; u == 0.5 gives ui == dim (reference's latent overrun) */
    int vi = (int)((v + 0.5) * dim);

    *face_out = face;
    *texel_index = (long)ui + (long)vi * dim;
    return *texel_index >= 0 && *texel_index < (long)dim * dim;
}

static _Thread_local trt_oracle_ray_log *g_ray_log = NULL;

void trt_oracle_set_ray_log(trt_oracle_ray_log *log) { g_ray_log = log; }

static inline void log_ray(v3 o, v3 d, int kind)
{
    trt_oracle_ray_log *l = g_ray_log;
    if (l && l->count < l->capacity)
    {
        double *r = l->rays + 6 * l->count;
        r[0] = o.x, r[1] = o.y, r[2] = o.z, r[3] = d.x, r[4] = d.y, r[5] = d.z;
        l->kinds[l->count++] = (unsigned char)kind;
    }
}

typedef struct
{
    ObjectType what;
    v3 point;  /* nudged hit point, or the ray origin on a miss */
    v3 normal; /* unit normal, or the unit ray direction on a miss */
    Material material;
} surface;

/* TRT.c:793-889.  want_surface==0 is the shadow-ray form (normal/material pointers NULL in the
 * reference): the skybox sample of a miss is skipped because its value is discarded there. */
static inline surface closest_hit(const Scene *scene, v3 o, v3 d, int want_surface)
{
    surface best;
    best.what = NONE;
    double best_d2 = INFINITY;
    v3 best_point = o, best_normal = d;
    memset(&best.material, 0, sizeof best.material);

    for (int i = 0; i < scene->num_spheres; i++) /* TRT.c:805-828 */
    {
        const Sphere *s = &scene->spheres[i];
        v3 p;
        if (hit_sphere(o, d, s, &p))
        {
            v3 back = sub(o, p);
            double d2 = dot(back, back); /* squared distance recomputed from the rounded hit point */
            if (d2 < best_d2)            /* strict: the first index wins ties */
            {
                best.what = SPHERE;
                best_d2 = d2;
                best_point = p;
                best_normal = sub(p, v3_ofp(&s->center));
                best.material = s->material;
            }
        }
    }

    {
        v3 p;
        if (hit_plane(o, d, &scene->ground, &p)) /* TRT.c:831-853 */
        {
            v3 back = sub(o, p);
            double d2 = dot(back, back);
            if (d2 < best_d2)
            {
                best.what = GROUND;
                best_d2 = d2;
                best_point = p;
                best_normal = v3_of(&scene->ground.normal);
                int odd = (int)(floor(p.x) + floor(p.z)) & 1; /* TRT.c:850 */
                best.material = odd ? scene->ground.odd_material : scene->ground.even_material;
            }
        }
    }

    if (best.what == NONE) /* TRT.c:858-867 */
    {
        best_point = o;
        best_normal = d;
        if (want_surface)
        {
            int face;
            long idx;
            Vector dir;
            v3_to(&dir, d);
            Color texel = {0, 0, 0};
            if (trt_oracle_skybox_lookup(scene, &dir, &face, &idx))
                texel = scene->skybox.colors[face][idx];
            else
            { /* reference reads past the face here (undefined); the build defines it as the last texel */
                long last = (long)scene->skybox.dim * scene->skybox.dim - 1;
                texel = scene->skybox.colors[face][idx < 0 ? 0 : last];
            }
            memset(&best.material, 0, sizeof best.material); /* reflectivity = specularity = 0 (TRT.c:866) */
            best.material.color.x = texel.r / 255.0;
            best.material.color.y = texel.g / 255.0;
            best.material.color.z = texel.b / 255.0;
        }
    }
    else /* TRT.c:868-875: pull the hit point 1e-6 back toward the ray origin */
    {
        v3 back = scale(unit(sub(o, best_point)), TRT_NUDGE);
        best_point = add(best_point, back);
    }

    best.point = best_point;
    best.normal = unit(best_normal); /* TRT.c:878 */
    return best;
}

ObjectType trt_oracle_trace_ray(const Scene *scene, const Ray *ray, Point *intersection, Vector *normal, Material *material)
{
    surface s = closest_hit(scene, v3_ofp(&ray->origin), v3_of(&ray->direction), normal != NULL || material != NULL);
    if (intersection)
        v3_top(intersection, s.point);
    if (normal)
        v3_to(normal, s.normal);
    if (material)
        *material = s.material;
    return s.what;
}

/* TRT.c:894-963.  Diffuse only; n.l is capped at 1 but NOT floored at 0. */
static inline v3 lit_color(const Scene *scene, v3 at, v3 normal, v3 albedo, trt_oracle_stats *st)
{
    v3 out = {0.0, 0.0, 0.0};

    for (int i = 0; i < scene->num_directional_lights; i++) /* TRT.c:900-923 */
    {
        const DirectionalLight *l = &scene->directional_lights[i];
        v3 to_light = unit(scale(v3_of(&l->direction), -1.0));
        st->shadow_rays++;
        log_ray(at, to_light, 1);
        surface blocker = closest_hit(scene, at, to_light, 0);
        if (blocker.what == NONE)
        {
            v3 diffuse = scale(v3_of(&l->color), fmin(dot(normal, to_light), 1.0));
            out = add(out, mulc(diffuse, albedo));
        }
    }

    for (int i = 0; i < scene->num_point_lights; i++) /* TRT.c:926-957 */
    {
        const PointLight *l = &scene->point_lights[i];
        v3 to_light = sub(v3_ofp(&l->position), at);
        double light_d2 = dot(to_light, to_light);
        double strength = clampd(l->intensity / light_d2, 0.0, 1.0);
        to_light = unit(to_light);
        st->shadow_rays++;
        log_ray(at, to_light, 2);
        surface blocker = closest_hit(scene, at, to_light, 0);
        v3 to_blocker = sub(blocker.point, at); /* on a miss blocker.point == at, so this is 0 */
        double blocker_d2 = dot(to_blocker, to_blocker);
        if (blocker.what == NONE || light_d2 < blocker_d2)
        {
            v3 diffuse = scale(v3_of(&l->color), strength * fmin(dot(normal, to_light), 1.0));
            out = add(out, mulc(diffuse, albedo));
        }
    }

    out.x = clampd(out.x, 0.0, 1.0); /* TRT.c:960 */
    out.y = clampd(out.y, 0.0, 1.0);
    out.z = clampd(out.z, 0.0, 1.0);
    return out;
}

void trt_oracle_apply_lighting(const Scene *scene, const Point *intersection, const Vector *normal, Material *material,
                               trt_oracle_stats *stats)
{
    trt_oracle_stats local = {0, 0, 0, 0};
    v3 c = lit_color(scene, v3_ofp(intersection), v3_of(normal), v3_of(&material->color), stats ? stats : &local);
    v3_to(&material->color, c);
}

/* one pixel: TRT.c:977-1066 */
static inline v3 shade_pixel(const Scene *scene, int width, int height, int row, int column, int bounce_limit,
                             int rays_per_pixel, trt_oracle_stats *st)
{
    const Camera *cam = &scene->camera;
    v3 bx = v3_of(&cam->frame.basis.x), by = v3_of(&cam->frame.basis.y), bz = v3_of(&cam->frame.basis.z);
    v3 eye = v3_ofp(&cam->frame.origin);
    v3 mean = {0.0, 0.0, 0.0};

    for (int k = 0; k < rays_per_pixel; k++)
    {
        double pixel_w = cam->screen_width / width; /* TRT.c:981-982 */
        double pixel_h = cam->screen_height / height;

        double sx = (((double)column / (double)width) * cam->screen_width - cam->screen_width / 2.0); /* TRT.c:987 */
        double sy = -(((double)row / (double)height) * cam->screen_height - cam->screen_height / 2.0); /* TRT.c:988 */
        double sz = -cam->screen_distance;

        sx += trt_oracle_triangle_wave(2 * TRT_PI * k / rays_per_pixel) / 2 * pixel_w; /* TRT.c:992 */
        sy += trt_oracle_triangle_wave(TRT_PI * k / rays_per_pixel) / 2 * pixel_h;     /* TRT.c:993 */

        /* (0 + bx*sx) + by*sy + bz*sz, then minus the eye (sic), TRT.c:996-1005 */
        v3 dir = {0.0, 0.0, 0.0};
        dir = add(dir, scale(bx, sx));
        dir = add(dir, scale(by, sy));
        dir = add(dir, scale(bz, sz));
        dir = sub(dir, eye);
        dir = unit(dir);

        v3 org = eye;
        v3 sample = {0.0, 0.0, 0.0};
        int bounces = 0;
        double weight = 1.0;     /* color_contribution */
        double weight_sum = 0.0; /* color_contribution_total */
        int going = 1;
        st->samples++;

        while (going && bounces < bounce_limit && weight > 0.00001) /* TRT.c:1018 */
        {
            st->path_rays++;
            log_ray(org, dir, 0);
            surface s = closest_hit(scene, org, dir, 1);
            v3 color = v3_of(&s.material.color);
            if (s.what != NONE)
                color = lit_color(scene, s.point, s.normal, color, st);
            else
                st->sky_lookups++;

            weight_sum += weight; /* TRT.c:1034-1035 */
            color = scale(color, weight);

            if (s.what != NONE) /* TRT.c:1039-1048 */
            {
                weight *= s.material.reflectivity;
                bounces++;
            }
            else
            {
                weight = 0.0;
                going = 0;
            }

            sample = add(sample, color);      /* TRT.c:1051 */
            dir = unit(reflect(dir, s.normal)); /* TRT.c:1054-1055 */
            org = s.point;                    /* TRT.c:1056 */
        }

        sample = scale(sample, 1.0 / weight_sum); /* TRT.c:1061 */
        mean = add(mean, sample);                 /* TRT.c:1063 */
    }
    return scale(mean, 1.0 / rays_per_pixel); /* TRT.c:1065 */
}

void trt_oracle_render_rows(const Scene *scene, Vector *out, int width, int height, int row_begin, int row_end,
                            int bounce_limit, int rays_per_pixel, int num_threads, trt_oracle_stats *stats)
{
    unsigned long long n_path = 0, n_shadow = 0, n_sky = 0, n_samples = 0;
#ifdef _OPENMP
    if (num_threads < 1)
        num_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads) reduction(+ : n_path, n_shadow, n_sky, n_samples)
#endif
    for (int row = row_begin; row < row_end; row++)
    {
        trt_oracle_stats st = {0, 0, 0, 0};
        for (int column = 0; column < width; column++)
        {
            v3 c = shade_pixel(scene, width, height, row, column, bounce_limit, rays_per_pixel, &st);
            v3_to(&out[(size_t)(row - row_begin) * width + column], c);
        }
        n_path += st.path_rays;
        n_shadow += st.shadow_rays;
        n_sky += st.sky_lookups;
        n_samples += st.samples;
    }
    (void)num_threads;
    if (stats)
    {
        stats->path_rays = n_path;
        stats->shadow_rays = n_shadow;
        stats->sky_lookups = n_sky;
        stats->samples = n_samples;
    }
}

/* TRT.c:966-1069 */
void trt_oracle_project_scene(const Scene *scene, Screen *screen, int bounce_limit, int rays_per_pixel, int num_threads,
                              trt_oracle_stats *stats)
{
    trt_oracle_render_rows(scene, screen->pixels, screen->width, screen->height, 0, screen->height, bounce_limit,
                           rays_per_pixel, num_threads, stats);
}

/* ==================================================================================================================
 * EXTENSION, PARITY UNPINNED: refraction.  The reference has no refraction (Material = {color, reflectivity,
 * specularity}, TRT.c:114-119; BASELINE config 3 names "refractive materials" all the same).  What follows restates the
 * semantics the HIP kernel's REFRACT variant implements (terminalraytracer_amd/csrc/trt_rounds.hpp), operation for
 * operation, so that the GPU can be checked bit for bit against SOMETHING -- but that something is this file, not the
 * reference.  With every index of refraction 0 it reduces to shade_pixel above.
 *
 *   ior[i] > 0 makes sphere i a refractor (index of refraction relative to the outside).  A path ray that hits it from
 *   outside is shaded exactly as the reference shades a hit (lighting, weight *= reflectivity, one bounce) and continues
 *   along the refracted direction from a point nudged 1e-6 PAST the surface; inside, the sphere itself is intersected with
 *   the FAR root (the reference only ever uses the near root, TRT.c:657); leaving it adds no colour and no weight but costs
 *   one bounce (so that a ray trapped by total internal reflection ends); total internal reflection mirrors the ray.
 *   Shadow rays are the reference's: a refractor blocks light like any sphere.
 * ================================================================================================================== */
static inline int hit_sphere_far(v3 o, v3 d, const Sphere *s, v3 *p)
{
    v3 oc = {o.x - s->center.x, o.y - s->center.y, o.z - s->center.z};
    double a = dot(d, d);
    double b = 2.0 * dot(oc, d);
    double c = dot(oc, oc) - s->radius * s->radius;
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0)
        return 0;
    double t1 = (-b + sqrt(disc)) / (2.0 * a);
    if (!(t1 > 0.0))
        return 0;
    p->x = o.x + t1 * d.x;
    p->y = o.y + t1 * d.y;
    p->z = o.z + t1 * d.z;
    return 1;
}

typedef struct
{
    surface s;  /* as closest_hit returns it */
    int sphere; /* index of the sphere that was hit, -1 otherwise */
    v3 raw;     /* the hit point before the nudge */
    v3 back;    /* the nudge vector: unit(o - raw) * 1e-6 */
} surface_ext;

static inline surface_ext closest_hit_ext(const Scene *scene, v3 o, v3 d, int inside)
{
    surface_ext r;
    r.sphere = -1;
    r.raw = o;
    r.back = (v3){0.0, 0.0, 0.0};
    surface best;
    best.what = NONE;
    double best_d2 = INFINITY;
    v3 best_point = o, best_normal = d;
    memset(&best.material, 0, sizeof best.material);
    for (int i = 0; i < scene->num_spheres; i++)
    {
        const Sphere *s = &scene->spheres[i];
        v3 p;
        if (i == inside ? hit_sphere_far(o, d, s, &p) : hit_sphere(o, d, s, &p))
        {
            v3 back = sub(o, p);
            double d2 = dot(back, back);
            if (d2 < best_d2)
            {
                best.what = SPHERE;
                best_d2 = d2;
                best_point = p;
                best_normal = sub(p, v3_ofp(&s->center));
                best.material = s->material;
                r.sphere = i;
            }
        }
    }
    {
        v3 p;
        if (hit_plane(o, d, &scene->ground, &p))
        {
            v3 back = sub(o, p);
            double d2 = dot(back, back);
            if (d2 < best_d2)
            {
                best.what = GROUND;
                best_d2 = d2;
                best_point = p;
                best_normal = v3_of(&scene->ground.normal);
                int odd = (int)(floor(p.x) + floor(p.z)) & 1;
                best.material = odd ? scene->ground.odd_material : scene->ground.even_material;
                r.sphere = -1;
            }
        }
    }
    if (best.what == NONE)
    {
        surface sky = closest_hit(scene, o, d, 1); /* a miss: exactly the reference's sky sample */
        r.s = sky;
        return r;
    }
    r.raw = best_point;
    r.back = scale(unit(sub(o, best_point)), TRT_NUDGE);
    best.point = add(best_point, r.back);
    best.normal = unit(best_normal);
    r.s = best;
    return r;
}

static inline v3 shade_pixel_refractive(const Scene *scene, const double *ior, int width, int height, int row, int column, int bounce_limit,
                                        int rays_per_pixel, trt_oracle_stats *st)
{
    const Camera *cam = &scene->camera;
    v3 bx = v3_of(&cam->frame.basis.x), by = v3_of(&cam->frame.basis.y), bz = v3_of(&cam->frame.basis.z);
    v3 eye = v3_ofp(&cam->frame.origin);
    v3 mean = {0.0, 0.0, 0.0};
    for (int k = 0; k < rays_per_pixel; k++)
    {
        double pixel_w = cam->screen_width / width;
        double pixel_h = cam->screen_height / height;
        double sx = (((double)column / (double)width) * cam->screen_width - cam->screen_width / 2.0);
        double sy = -(((double)row / (double)height) * cam->screen_height - cam->screen_height / 2.0);
        double sz = -cam->screen_distance;
        sx += trt_oracle_triangle_wave(2 * TRT_PI * k / rays_per_pixel) / 2 * pixel_w;
        sy += trt_oracle_triangle_wave(TRT_PI * k / rays_per_pixel) / 2 * pixel_h;
        v3 dir = {0.0, 0.0, 0.0};
        dir = add(dir, scale(bx, sx));
        dir = add(dir, scale(by, sy));
        dir = add(dir, scale(bz, sz));
        dir = sub(dir, eye);
        dir = unit(dir);
        v3 org = eye;
        v3 sample = {0.0, 0.0, 0.0};
        int bounces = 0, inside = -1, going = 1;
        double weight = 1.0, weight_sum = 0.0;
        st->samples++;
        while (going && bounces < bounce_limit && weight > 0.00001)
        {
            st->path_rays++;
            surface_ext h = closest_hit_ext(scene, org, dir, inside);
            const int leaving = h.s.what == SPHERE && h.sphere == inside;
            if (!leaving)
            { /* the reference's bounce, TRT.c:1026-1051 */
                v3 color = v3_of(&h.s.material.color);
                if (h.s.what != NONE)
                    color = lit_color(scene, h.s.point, h.s.normal, color, st);
                else
                    st->sky_lookups++;
                weight_sum += weight;
                color = scale(color, weight);
                if (h.s.what != NONE)
                {
                    weight *= h.s.material.reflectivity;
                    bounces++;
                }
                else
                {
                    weight = 0.0;
                    going = 0;
                }
                sample = add(sample, color);
            }
            else
                bounces++; /* leaving a refractor: no colour, no weight */
            int bent = 0;
            if (h.s.what == SPHERE && ior[h.sphere] > 0.0)
            {
                const v3 nn = leaving ? scale(h.s.normal, -1.0) : h.s.normal; /* the normal facing the incoming ray */
                const double cosi = -dot(nn, dir);
                const double eta = leaving ? ior[h.sphere] : 1.0 / ior[h.sphere];
                const double kk = 1.0 - (eta * eta) * (1.0 - cosi * cosi);
                if (!(kk < 0.0))
                {
                    const double f = eta * cosi - sqrt(kk);
                    v3 t = {eta * dir.x + f * nn.x, eta * dir.y + f * nn.y, eta * dir.z + f * nn.z};
                    dir = unit(t);
                    org = sub(h.raw, h.back); /* 1e-6 past the surface */
                    inside = leaving ? -1 : h.sphere;
                    bent = 1;
                }
                else
                { /* total internal reflection: mirror about the facing normal, stay on this side */
                    dir = unit(reflect(dir, nn));
                    org = h.s.point;
                    bent = 1;
                }
            }
            if (!bent)
            {
                dir = unit(reflect(dir, h.s.normal)); /* TRT.c:1054-1056 */
                org = h.s.point;
            }
        }
        sample = scale(sample, 1.0 / weight_sum);
        mean = add(mean, sample);
    }
    return scale(mean, 1.0 / rays_per_pixel);
}

void trt_oracle_project_scene_refractive(const Scene *scene, const double *ior, Screen *screen, int bounce_limit, int rays_per_pixel,
                                         int num_threads, trt_oracle_stats *stats)
{
    unsigned long long n_path = 0, n_shadow = 0, n_sky = 0, n_samples = 0;
    const int width = screen->width, height = screen->height;
#ifdef _OPENMP
    if (num_threads < 1)
        num_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads) reduction(+ : n_path, n_shadow, n_sky, n_samples)
#endif
    for (int row = 0; row < height; row++)
    {
        trt_oracle_stats st = {0, 0, 0, 0};
        for (int column = 0; column < width; column++)
        {
            v3 c = shade_pixel_refractive(scene, ior, width, height, row, column, bounce_limit, rays_per_pixel, &st);
            v3_to(&screen->pixels[(size_t)row * width + column], c);
        }
        n_path += st.path_rays;
        n_shadow += st.shadow_rays;
        n_sky += st.sky_lookups;
        n_samples += st.samples;
    }
    if (stats)
    {
        stats->path_rays = n_path;
        stats->shadow_rays = n_shadow;
        stats->sky_lookups = n_sky;
        stats->samples = n_samples;
    }
}

void trt_oracle_rgb8(const Vector *pixels, size_t count, unsigned char *rgb)
{
    for (size_t i = 0; i < count; i++)
    {
        rgb[3 * i + 0] = (unsigned char)(int)(pixels[i].x * 255);
        rgb[3 * i + 1] = (unsigned char)(int)(pixels[i].y * 255);
        rgb[3 * i + 2] = (unsigned char)(int)(pixels[i].z * 255);
    }
}

unsigned long long trt_oracle_fnv1a64(const void *data, size_t bytes)
{
    const unsigned char *p = (const unsigned char *)data;
    unsigned long long h = 1469598103934665603ULL;
    for (size_t i = 0; i < bytes; i++)
    {
        h ^= p[i];
        h *= 1099511628211ULL;
    }
    return h;
}

void trt_oracle_div_sqrt(const double *a, const double *b, size_t n, double *quot, double *root)
{
    for (size_t i = 0; i < n; i++)
    {
        quot[i] = a[i] / b[i];
        root[i] = sqrt(a[i]);
    }
}
