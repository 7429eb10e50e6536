/*
 * vrt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See vrt_oracle.h.
 *
 * PARITY UNPINNED by the reference (no golden vectors exist; the GLSL cannot run here); pinned by
 * hand-computed known-answer tests only.  Every function cites the reference lines it restates.
 * Paths are relative to /root/reference.
 *
 * Compile with:  gcc -O2 -ffp-contract=off -fno-fast-math   (fp32 everywhere, no FMA contraction).
 *
 * Canonical resolutions of the shader's undefined corners (DESIGN.md "Spec"):
 *   A  RayHitInternal.mask uninitialised when the first sampled voxel is solid
 *      -> mask = box-entry axes (tminDir == tmin) when boxIntersection moved the origin,
 *         (0,0,0) otherwise; normalize(0) = 0.
 *   B  material/pos/normal uninitialised on a miss -> 0.
 *   C  getVoxel samples at pos/bounds with nearest filtering -> integer fetch voxel[pos].
 *   I  vec3(mask)*deltaDist with deltaDist = inf (axis-parallel ray) is 0*inf = NaN in the
 *      literal GLSL -> select semantics: side += mask ? delta : 0.  min/max = IEEE fminf/fmaxf.
 *   F  unwritten outColor.a / outPos.w / outNormal.w -> 0.
 *   atan/asin/exp/normalize: fixed polynomial / IEEE definitions below (GLSL only bounds their
 *      error), so that a second implementation of the same spec is bit-identical.
 */
#include "vrt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* math primitives                                                                             */
/* ------------------------------------------------------------------------------------------ */

#define VO_PI      3.14159265358979323846f
#define VO_PI_2    1.57079632679489661923f
#define VO_PI_4    0.78539816339744830962f

static inline float f_sign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }

static inline float dot3(const float a[3], const float b[3])
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

static inline float length3(const float a[3]) { return sqrtf(dot3(a, a)); }

/* GLSL normalize(); canonical: component / sqrt(dot), normalize(0) = 0 (rule A). */
static inline void normalize3(const float a[3], float out[3])
{
    float l = length3(a);
    if (l == 0.0f) { out[0] = out[1] = out[2] = 0.0f; return; }
    out[0] = a[0] / l; out[1] = a[1] / l; out[2] = a[2] / l;
}

/* atan(t), |t| <= 1 after the caller's min/max ratio: Cephes atanf reduction + degree-4 (in t^2)
 * polynomial. */
static inline float atan_unit(float t)
{
    float y0 = 0.0f;
    if (t > 0.4142135623730950f) { y0 = VO_PI_4; t = (t - 1.0f) / (t + 1.0f); }
    float z = t * t;
    float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z
               - 3.33329491539e-1f) * z * t + t;
    return p + y0;
}

float vo_atan2f(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    int swap = ay > ax;
    float t = swap ? ax / ay : ay / ax;
    float r = atan_unit(t);
    if (swap) r = VO_PI_2 - r;
    if (x < 0.0f) r = VO_PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

/* Cephes asinf. */
float vo_asinf(float x)
{
    float a = fabsf(x);
    int big = a > 0.5f;
    float z, s;
    if (a > 1.0f) a = 1.0f;                     /* clamp: GLSL asin undefined for |x| > 1 */
    if (big) { z = 0.5f * (1.0f - a); s = sqrtf(z); }
    else     { s = a; z = a * a; }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z
                + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * s + s;
    if (big) p = VO_PI_2 - (p + p);
    return (x < 0.0f) ? -p : p;
}

/* Cephes expf (exact 1 at +-0, 0 below -87, +inf above 88). */
float vo_expf(float x)
{
    if (x != x) return x;
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return INFINITY;
    float fx = floorf(x * 1.44269504088896341f + 0.5f);
    x = x - fx * 0.693359375f;
    x = x - fx * -2.12194440e-4f;
    float z = x * x;
    float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x
                 + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    int n = (int)fx;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)(n + 127) << 23;
    return p * s.f;
}

/* Vulkan float -> UNORM8 / SNORM8 conversion, canonical round-half-up of the scaled value. */
uint8_t vo_unorm8(float c)
{
    c = fminf(fmaxf(c, 0.0f), 1.0f);
    return (uint8_t)floorf(c * 255.0f + 0.5f);
}
int8_t vo_snorm8(float c)
{
    c = fminf(fmaxf(c, -1.0f), 1.0f);
    return (int8_t)floorf(c * 127.0f + 0.5f);
}

/* nearest / repeat texel index (texture_2d.cpp:158-163). */
static inline uint32_t wrap_texel(float u, uint32_t n)
{
    float fr = u - floorf(u);
    if (!(fr >= 0.0f)) return 0;
    int32_t i = (int32_t)floorf(fr * (float)n);
    if (i >= (int32_t)n) i = (int32_t)n - 1;
    return (uint32_t)i;
}

/* ------------------------------------------------------------------------------------------ */
/* per-pixel context                                                                           */
/* ------------------------------------------------------------------------------------------ */

typedef struct pix_ctx {
    const vo_scene*  sc;
    const vo_push*   pc;
    const vo_params* pr;
    int px, py;
    uint32_t fetches;   /* voxel fetches by all rays of this pixel */
    uint32_t rays;
} pix_ctx;

typedef struct ray_int {          /* RayHitInternal, voxel_volume.frag:33-41 */
    float    pos[3];
    float    side[3];
    float    delta[3];
    int      step[3];
    uint32_t material;
    int      mask[3];
    int      map[3];
    uint32_t fetches;
} ray_int;

typedef struct ray_hit {          /* RayHit, voxel_volume.frag:43-49 */
    uint32_t material;
    float    pos[3];
    float    normal[3];
    float    dir[3];
} ray_hit;

/* getVoxel, voxel_volume.frag:73-77 (rule C: integer fetch). */
static inline uint32_t get_voxel(const vo_scene* sc, const int p[3])
{
    size_t W = sc->dims[0], H = sc->dims[1];
    if (!sc->voxels) {                                        /* brick storage of the same texture */
        size_t nbx = W / 8, nby = H / 8;
        uint32_t b = sc->brick_grid[(size_t)(p[0] >> 3) + ((size_t)(p[1] >> 3) + (size_t)(p[2] >> 3) * nby) * nbx];
        if (b == 0) return 0;
        return sc->brick_pool[(size_t)(b - 1) * 512 + (size_t)(p[0] & 7) + (size_t)(p[1] & 7) * 8 + (size_t)(p[2] & 7) * 64];
    }
    return sc->voxels[(size_t)p[0] + (size_t)p[1] * W + (size_t)p[2] * W * H];
}

/* skyColor, voxel_volume.frag:98-105. */
static void sky_color(const vo_scene* sc, const float d[3], float out[3])
{
    float u = vo_atan2f(d[2], d[0]) * 0.1591f + 0.5f;
    float v = vo_asinf(-d[1]) * 0.3183f + 0.5f;
    uint32_t x = wrap_texel(u, sc->sky_w), y = wrap_texel(v, sc->sky_h);
    const float* t = sc->sky + ((size_t)y * sc->sky_w + x) * 4;
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
}

/* fragmentNoiseSeq, voxel_volume.frag:80-89. */
static void fragment_noise_seq(const pix_ctx* c, uint32_t num, float out[3])
{
    uint32_t offset = num * 32u + c->pc->frame % 32u;
    const float g = 1.22074408460575947536f;
    const float a0 = 1.0f / g, a1 = 1.0f / (g * g), a2 = 1.0f / ((g * g) * g);
    float pxf = ((float)c->px + 0.5f) / 512.0f + 0.5f;   /* gl_FragCoord.xy / NOISE_SIZE + 0.5 */
    float pyf = ((float)c->py + 0.5f) / 512.0f + 0.5f;
    uint32_t tx = wrap_texel(pxf, c->sc->noise_w), ty = wrap_texel(pyf, c->sc->noise_h);
    const uint8_t* t = c->sc->noise + ((size_t)ty * c->sc->noise_w + tx) * 4;
    float fo = (float)offset;
    float n0 = (float)t[0] / 255.0f + fo * a0;
    float n1 = (float)t[1] / 255.0f + fo * a1;
    float n2 = (float)t[2] / 255.0f + fo * a2;
    out[0] = n0 - floorf(n0);                            /* mod(x, 1.0) */
    out[1] = n1 - floorf(n1);
    out[2] = n2 - floorf(n2);
}

/* randomDir, voxel_volume.frag:92-95. */
static void random_dir(const pix_ctx* c, uint32_t num, float out[3])
{
    float n[3], v[3];
    fragment_noise_seq(c, num, n);
    v[0] = n[0] * 2.0f - 1.0f; v[1] = n[1] * 2.0f - 1.0f; v[2] = n[2] * 2.0f - 1.0f;
    normalize3(v, out);
}

/* boxIntersection, voxel_volume.frag:109-125.  entry[] = axes with tminDir == tmin (rule A). */
static int box_intersection(const vo_scene* sc, const float s[3], const float d[3],
                            float out[3], int entry[3])
{
    float tmn[3], tmx[3];
    for (int a = 0; a < 3; a++) {
        float inv = 1.0f / d[a];
        float t1 = (-s[a]) * inv;
        float t2 = ((float)sc->dims[a] - s[a]) * inv;
        tmn[a] = fminf(t1, t2);
        tmx[a] = fmaxf(t1, t2);
    }
    float tmin = fmaxf(tmn[0], fmaxf(tmn[1], tmn[2]));
    float tmax = fminf(tmx[0], fminf(tmx[1], tmx[2]));
    if (tmin >= 0.0f && tmax >= tmin) {
        float t = tmin + 0.1f;
        for (int a = 0; a < 3; a++) { out[a] = s[a] + t * d[a]; entry[a] = (tmn[a] == tmin); }
        return 1;
    }
    for (int a = 0; a < 3; a++) { out[a] = s[a]; entry[a] = 0; }
    return 0;
}

/* traceRayInt, voxel_volume.frag:127-174. */
static void trace_ray_int(const vo_scene* sc, const float start[3], const float dir[3],
                          uint32_t max_steps, ray_int* r)
{
    box_intersection(sc, start, dir, r->pos, r->mask);           /* :132, rule A mask init */
    float sg[3];
    for (int a = 0; a < 3; a++) {
        r->map[a]   = (int)floorf(r->pos[a]);                     /* :135 */
        r->delta[a] = fabsf(1.0f / dir[a]);                       /* :138 */
        sg[a]       = f_sign(dir[a]);
        r->step[a]  = (int)sg[a];                                 /* :141 */
        r->side[a]  = ((sg[a] * ((float)r->map[a] - r->pos[a]) + sg[a] * 0.5f) + 0.5f)
                      * r->delta[a];                              /* :144 */
    }
    r->material = 0;                                              /* rule B */
    r->fetches = 0;
    for (uint32_t i = 0; i < max_steps; i++) {                    /* :146 */
        if (r->map[0] < 0 || r->map[0] >= (int)sc->dims[0] ||
            r->map[1] < 0 || r->map[1] >= (int)sc->dims[1] ||
            r->map[2] < 0 || r->map[2] >= (int)sc->dims[2])
            break;                                                /* :149-154 */
        r->material = get_voxel(sc, r->map);                      /* :157 */
        r->fetches++;
        if (r->material != 0) break;                              /* :158 */
        int m0 = r->side[0] <= fminf(r->side[1], r->side[2]);     /* :164 */
        int m1 = r->side[1] <= fminf(r->side[2], r->side[0]);
        int m2 = r->side[2] <= fminf(r->side[0], r->side[1]);
        r->mask[0] = m0; r->mask[1] = m1; r->mask[2] = m2;
        if (m0) { r->side[0] = r->side[0] + r->delta[0]; r->map[0] += r->step[0]; }  /* :167,:170 */
        if (m1) { r->side[1] = r->side[1] + r->delta[1]; r->map[1] += r->step[1]; }
        if (m2) { r->side[2] = r->side[2] + r->delta[2]; r->map[2] += r->step[2]; }
    }
}

/* traceRay, voxel_volume.frag:176-196. */
static void trace_ray(pix_ctx* c, const float start[3], const float dir[3], uint32_t max_steps,
                      ray_hit* h, ray_int* ri_out)
{
    ray_int r;
    trace_ray_int(c->sc, start, dir, max_steps, &r);
    c->fetches += r.fetches; c->rays++;
    h->material = r.material;
    h->dir[0] = dir[0]; h->dir[1] = dir[1]; h->dir[2] = dir[2];
    if (r.material != 0) {
        float n[3], m[3];
        for (int a = 0; a < 3; a++) {
            n[a] = r.mask[a] ? (float)(-r.step[a]) : 0.0f;                 /* :188 */
            m[a] = r.mask[a] ? (r.side[a] - r.delta[a]) : 0.0f;            /* :191 */
        }
        normalize3(n, h->normal);
        float d = length3(m);
        for (int a = 0; a < 3; a++) h->pos[a] = r.pos[a] + d * dir[a];     /* :192 */
    } else {
        for (int a = 0; a < 3; a++) { h->pos[a] = 0.0f; h->normal[a] = 0.0f; }   /* rule B */
    }
    if (ri_out) *ri_out = r;
}

/* traceRayHit, voxel_volume.frag:198-202. */
static int trace_ray_hit(pix_ctx* c, const float start[3], const float dir[3], uint32_t max_steps)
{
    ray_int r;
    trace_ray_int(c->sc, start, dir, max_steps, &r);
    c->fetches += r.fetches; c->rays++;
    return r.material != 0;
}

/* calcAmbient, voxel_volume.frag:205-227. */
static void calc_ambient(pix_ctx* c, const ray_hit* hit, uint32_t depth, float out[3])
{
    const vo_params* pr = c->pr;
    float ambient = 0.0f;
    if (pr->ao_samples == 0) {
        ambient = 1.0f;
    } else {
        float sample_frac = 1.0f / (float)pr->ao_samples;
        for (uint32_t i = 0; i < pr->ao_samples; i++) {
            float rd[3], dir[3], o[3];
            random_dir(c, i + depth * pr->ao_samples, rd);
            for (int a = 0; a < 3; a++) dir[a] = hit->normal[a] + rd[a];
            for (int a = 0; a < 3; a++) o[a] = hit->pos[a] + dir[a] * 0.01f;
            if (trace_ray_hit(c, o, dir, pr->ao_steps))
                ambient += sample_frac;
        }
    }
    float sky[3];
    sky_color(c->sc, hit->normal, sky);
    float k = ambient * pr->ambient_intensity;
    out[0] = k * sky[0]; out[1] = k * sky[1]; out[2] = k * sky[2];
}

/* isShadowed, voxel_volume.frag:230-233. */
static int is_shadowed(pix_ctx* c, const ray_hit* hit)
{
    if (!c->pr->shadows) return 0;
    float o[3];
    for (int a = 0; a < 3; a++) o[a] = hit->pos[a] + hit->normal[a] * 0.01f;
    return trace_ray_hit(c, o, c->pr->light_dir, c->pr->max_steps);
}

/* color + colorHit, voxel_volume.frag:236-264. */
static void color_hit(pix_ctx* c, const ray_hit* hit, const float reflection[3], uint32_t depth,
                      float out[3])
{
    if (hit->material != 0) {
        const vo_params* pr = c->pr;
        const vo_material* mat = &c->sc->palette[hit->material];
        float ambient[3];
        calc_ambient(c, hit, depth, ambient);
        int shadowed = is_shadowed(c, hit);
        float diffuse[3] = {0.0f, 0.0f, 0.0f};
        if (!shadowed) {
            float diff = fmaxf(dot3(hit->normal, pr->light_dir), 0.0f);
            for (int a = 0; a < 3; a++) diffuse[a] = (diff * pr->light_color[a]) * pr->light_intensity;
        }
        float inv = (float)(depth + 1);
        for (int a = 0; a < 3; a++) {
            float specular = reflection[a] * mat->metallic;
            float col = ((diffuse[a] + specular) + ambient[a]) * mat->diffuse[a];
            out[a] = (col * 1.0f) / inv;
        }
    } else {
        sky_color(c->sc, hit->dir, out);
    }
}

/* colorMainRay, voxel_volume.frag:267-307. */
static void color_main_ray(pix_ctx* c, const ray_hit* hit, float out[3])
{
    const vo_params* pr = c->pr;
    const vo_material* mat = &c->sc->palette[hit->material];
    float reflection[3] = {0.0f, 0.0f, 0.0f};
    if (mat->metallic > 0.0f) {
        ray_hit bounces[VO_MAX_BOUNCES];
        ray_hit last = *hit;
        int last_idx = -1;
        uint32_t nb = pr->max_bounces > VO_MAX_BOUNCES ? VO_MAX_BOUNCES : pr->max_bounces;
        for (int i = 0; i < (int)nb; i++) {
            float k = 2.0f * dot3(last.normal, last.dir);             /* reflect(I,N) = I - 2 dot(N,I) N */
            float rdir[3], o[3];
            for (int a = 0; a < 3; a++) rdir[a] = last.dir[a] - k * last.normal[a];
            for (int a = 0; a < 3; a++) o[a] = last.pos[a] + last.normal[a] * 0.01f;
            ray_hit rh;
            trace_ray(c, o, rdir, pr->max_steps, &rh, NULL);
            bounces[i] = rh;
            last = rh;
            if (last.material == 0 || c->sc->palette[last.material].metallic <= 0.0f) {
                last_idx = i;
                break;
            }
        }
        for (int i = last_idx; i >= 0; i--) {
            float col[3];
            color_hit(c, &bounces[i], reflection, (uint32_t)i, col);
            for (int a = 0; a < 3; a++) reflection[a] += col[a];
        }
    }
    color_hit(c, hit, reflection, 0, out);
}

/* main() ray generation, voxel_volume.frag:312-322 with vScreenPos from screen_quad.vert:18-31. */
void vo_primary_ray(const vo_push* pc, int px, int py, float start[3], float dir[3])
{
    float W = (float)pc->screen_size[0], H = (float)pc->screen_size[1];
    float sx = (((float)px + 0.5f) / W) * 2.0f - 1.0f;
    float sy = (((float)py + 0.5f) / H) * 2.0f - 1.0f;
    float cd[3], v[3];
    normalize3(pc->cam_dir, cd);
    float jx = (pc->camera_jitter[0] / W) * -2.0f;
    float jy = (pc->camera_jitter[1] / H) * 2.0f;
    float jit[3] = {jx, jy, 0.0f};
    for (int a = 0; a < 3; a++) {
        float U = pc->cam_right[a];
        float V = (pc->cam_up[a] * H) / W;
        v[a] = ((cd[a] + sx * U) + sy * V) + jit[a];
    }
    normalize3(v, dir);
    start[0] = pc->cam_pos[0]; start[1] = pc->cam_pos[1]; start[2] = pc->cam_pos[2];
}

void vo_trace_ray(const vo_scene* sc, const float start[3], const float dir[3],
                  uint32_t max_steps, vo_hit* out)
{
    pix_ctx c; memset(&c, 0, sizeof c); c.sc = sc;
    ray_hit h; ray_int r;
    trace_ray(&c, start, dir, max_steps, &h, &r);
    out->material = h.material;
    out->mask = 0;
    for (int a = 0; a < 3; a++) {
        out->pos[a] = h.pos[a]; out->normal[a] = h.normal[a]; out->dir[a] = h.dir[a];
        out->voxel[a] = r.map[a]; out->p0[a] = r.pos[a];
        out->side[a] = r.side[a]; out->delta[a] = r.delta[a];
        if (r.mask[a]) out->mask |= 1u << a;
    }
    out->steps = r.fetches;
}

/* main(), voxel_volume.frag:309-346. */
static void render_pixel(const vo_scene* sc, const vo_push* pc, const vo_params* pr,
                         const vo_frame* f, int px, int py)
{
    pix_ctx c; c.sc = sc; c.pc = pc; c.pr = pr; c.px = px; c.py = py; c.fetches = 0; c.rays = 0;
    size_t i = (size_t)py * (size_t)pc->screen_size[0] + (size_t)px;
    float start[3], dir[3], col[3];
    vo_primary_ray(pc, px, py, start, dir);
    ray_hit h; ray_int r;
    trace_ray(&c, start, dir, pr->max_steps, &h, &r);
    uint32_t primary_fetches = r.fetches;
    float depth, mask, normal[3] = {0, 0, 0};
    if (h.material != 0) {
        color_main_ray(&c, &h, col);
        float dv[3] = {h.pos[0] - pc->cam_pos[0], h.pos[1] - pc->cam_pos[1], h.pos[2] - pc->cam_pos[2]};
        depth = length3(dv);
        mask = 0.9f;
        normal[0] = h.normal[0]; normal[1] = h.normal[1]; normal[2] = h.normal[2];
    } else {
        sky_color(sc, dir, col);
        depth = 0.0f;
        mask = 0.0f;
    }
    if (f->color_f) { f->color_f[i * 3 + 0] = col[0]; f->color_f[i * 3 + 1] = col[1]; f->color_f[i * 3 + 2] = col[2]; }
    if (f->color8) {
        f->color8[i * 4 + 0] = vo_unorm8(col[0]); f->color8[i * 4 + 1] = vo_unorm8(col[1]);
        f->color8[i * 4 + 2] = vo_unorm8(col[2]); f->color8[i * 4 + 3] = 0;
    }
    if (f->depth) f->depth[i] = depth;
    if (f->motion) { f->motion[i * 2] = 0.0f; f->motion[i * 2 + 1] = 0.0f; }
    if (f->mask8) f->mask8[i] = vo_unorm8(mask);
    if (f->position) {
        f->position[i * 4 + 0] = h.pos[0]; f->position[i * 4 + 1] = h.pos[1];
        f->position[i * 4 + 2] = h.pos[2]; f->position[i * 4 + 3] = 0.0f;
    }
    if (f->normal8) {
        f->normal8[i * 4 + 0] = vo_snorm8(normal[0]); f->normal8[i * 4 + 1] = vo_snorm8(normal[1]);
        f->normal8[i * 4 + 2] = vo_snorm8(normal[2]); f->normal8[i * 4 + 3] = 0;
    }
    int hit = h.material != 0;
    if (f->hit_id) f->hit_id[i] = (uint8_t)h.material;
    if (f->hit_voxel) for (int a = 0; a < 3; a++) f->hit_voxel[i * 3 + a] = hit ? (int16_t)r.map[a] : 0;
    if (f->hit_mask) f->hit_mask[i] = hit ? (uint8_t)(r.mask[0] | (r.mask[1] << 1) | (r.mask[2] << 2)) : 0;
    if (f->steps_primary) f->steps_primary[i] = primary_fetches;
    if (f->steps_total) f->steps_total[i] = c.fetches;
    if (f->rays_total) f->rays_total[i] = c.rays;
}

void vo_render_rows(const vo_scene* sc, const vo_push* pc, const vo_params* pr,
                    const vo_frame* out, int row0, int row1)
{
    int W = pc->screen_size[0];
    for (int y = row0; y < row1; y++)
        for (int x = 0; x < W; x++)
            render_pixel(sc, pc, pr, out, x, y);
}

typedef struct mt_job {
    const vo_scene* sc; const vo_push* pc; const vo_params* pr; const vo_frame* out;
    int t, n;
} mt_job;

static void* mt_main(void* p)
{
    mt_job* j = (mt_job*)p;
    int H = j->pc->screen_size[1];
    for (int y = j->t; y < H; y += j->n) vo_render_rows(j->sc, j->pc, j->pr, j->out, y, y + 1);
    return NULL;
}

void vo_render_mt(const vo_scene* sc, const vo_push* pc, const vo_params* pr,
                  const vo_frame* out, int nthreads)
{
    if (nthreads <= 1) { vo_render_rows(sc, pc, pr, out, 0, pc->screen_size[1]); return; }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
    mt_job* jobs = (mt_job*)malloc(sizeof(mt_job) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t].sc = sc; jobs[t].pc = pc; jobs[t].pr = pr; jobs[t].out = out; jobs[t].t = t; jobs[t].n = nthreads;
        pthread_create(&th[t], NULL, mt_main, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* ------------------------------------------------------------------------------------------ */
/* denoiser                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* denoiser_stage.cpp:143-154 */
void vo_denoise_pass_params(int pass, float phi_color0, float phi_normal0, float phi_pos0,
                            float step_width0, vo_denoise_params* out)
{
    float inv = 1.0f / (float)pass;           /* pass 0 -> +inf (rule E) */
    out->phi_color  = inv * phi_color0;
    out->phi_normal = inv * phi_normal0;
    out->phi_pos    = inv * phi_pos0;
    out->step_width = (float)pass * step_width0 + 1.0f;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

typedef struct guides { float c[4], n[4], p[4]; } guides;

static inline void texel_guides(const uint8_t* color, const int8_t* normal, const float* pos,
                                int W, int H, int x, int y, guides* g)
{
    x = clampi(x, 0, W - 1); y = clampi(y, 0, H - 1);         /* clamp-to-edge */
    size_t i = ((size_t)y * (size_t)W + (size_t)x) * 4;
    for (int k = 0; k < 4; k++) {
        g->c[k] = (float)color[i + k] / 255.0f;                /* UNORM8 decode */
        g->n[k] = fmaxf((float)normal[i + k] / 127.0f, -1.0f); /* SNORM8 decode */
        g->p[k] = pos[i + k];
    }
}

/* Sample the three guides at continuous pixel-space position (fx,fy) = uv*(W,H).
 * Integer-offset taps land on texel centres -> point sample (rule K); otherwise bilinear. */
static void sample_guides(const uint8_t* color, const int8_t* normal, const float* pos,
                          int W, int H, int px, int py, float ox, float oy, guides* g)
{
    if (ox == floorf(ox) && oy == floorf(oy)) {
        texel_guides(color, normal, pos, W, H, px + (int)ox, py + (int)oy, g);
        return;
    }
    float fx = ((float)px + 0.5f + ox) - 0.5f, fy = ((float)py + 0.5f + oy) - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    guides g00, g10, g01, g11;
    texel_guides(color, normal, pos, W, H, x0, y0, &g00);
    texel_guides(color, normal, pos, W, H, x0 + 1, y0, &g10);
    texel_guides(color, normal, pos, W, H, x0, y0 + 1, &g01);
    texel_guides(color, normal, pos, W, H, x0 + 1, y0 + 1, &g11);
    for (int k = 0; k < 4; k++) {
        float a, b;
        a = g00.c[k] + tx * (g10.c[k] - g00.c[k]); b = g01.c[k] + tx * (g11.c[k] - g01.c[k]); g->c[k] = a + ty * (b - a);
        a = g00.n[k] + tx * (g10.n[k] - g00.n[k]); b = g01.n[k] + tx * (g11.n[k] - g01.n[k]); g->n[k] = a + ty * (b - a);
        a = g00.p[k] + tx * (g10.p[k] - g00.p[k]); b = g01.p[k] + tx * (g11.p[k] - g01.p[k]); g->p[k] = a + ty * (b - a);
    }
}

static inline float dist2_4(const float a[4], const float b[4])
{
    float t0 = a[0] - b[0], t1 = a[1] - b[1], t2 = a[2] - b[2], t3 = a[3] - b[3];
    return ((t0 * t0 + t1 * t1) + t2 * t2) + t3 * t3;
}

/* denoiser.frag:38-73 */
void vo_denoise_pass(const uint8_t* color_in, const int8_t* normal, const float* position,
                     uint8_t* color_out, int W, int H, const vo_denoise_params* p, int mode,
                     int row0, int row1)
{
    /* glm::gauss(vec2(x,y), 0, vec2(2)) = exp(-(x^2+y^2)/8), denoiser_stage.cpp:52-59 */
    const float G0 = 1.0f, G1 = 0.8824969025845955f, G2 = 0.7788007830714049f;
    float kern[9], offx[9], offy[9];
    int ntaps;
    if (mode == VO_DENOISE_AS_SHIPPED) {
        /* std140 stride-16 view of the tightly packed arrays (rule D): kernel[0..2] = w[0],w[4],w[8],
         * offset[0..2] = o[0],o[2],o[4]; every other element reads out of bounds -> 0 weight. */
        ntaps = 3;
        kern[0] = G2; offx[0] = -1.0f; offy[0] = -1.0f;
        kern[1] = G0; offx[1] =  1.0f; offy[1] = -1.0f;
        kern[2] = G2; offx[2] =  0.0f; offy[2] =  0.0f;
    } else {
        ntaps = 9;
        for (int i = 0, y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++, i++) {
                offx[i] = (float)x; offy[i] = (float)y;
                int r2 = x * x + y * y;
                kern[i] = r2 == 0 ? G0 : (r2 == 1 ? G1 : G2);
            }
    }
    float sw = p->step_width;
    for (int py = row0; py < row1; py++) {
        for (int px = 0; px < W; px++) {
            guides s, o;
            texel_guides(color_in, normal, position, W, H, px, py, &s);       /* :43-45 */
            float sum[4] = {0, 0, 0, 0};
            float total = 0.0f;
            for (int i = 0; i < ntaps; i++) {
                sample_guides(color_in, normal, position, W, H, px, py, offx[i] * sw, offy[i] * sw, &o);
                float d2 = dist2_4(s.c, o.c);
                float cw = fminf(vo_expf((-d2) / p->phi_color), 1.0f);           /* :55 */
                d2 = fmaxf(dist2_4(s.n, o.n) / (sw * sw), 0.0f);
                float nw = fminf(vo_expf((-d2) / p->phi_normal), 1.0f);          /* :60 */
                d2 = dist2_4(s.p, o.p);
                float pw = fminf(vo_expf((-d2) / p->phi_pos), 1.0f);             /* :65 */
                float w = (cw * nw) * pw;
                for (int k = 0; k < 4; k++) sum[k] += (o.c[k] * w) * kern[i];    /* :68 */
                total += w * kern[i];                                            /* :69 */
            }
            size_t i4 = ((size_t)py * (size_t)W + (size_t)px) * 4;
            for (int k = 0; k < 4; k++) color_out[i4 + k] = vo_unorm8(sum[k] / total);   /* :72 */
        }
    }
}

/* denoiser_stage.cpp:204-257 */
int vo_denoise(const uint8_t* color_in, const int8_t* normal, const float* position,
               uint8_t* target0, uint8_t* target1, int W, int H, int iterations,
               float phi_color0, float phi_normal0, float phi_pos0, float step_width0, int mode)
{
    int last = -1;
    uint8_t* targets[2] = {target0, target1};
    for (int i = 0; i < iterations; i++) {
        int ping = i % 2;
        vo_denoise_params p;
        vo_denoise_pass_params(i, phi_color0, phi_normal0, phi_pos0, step_width0, &p);
        const uint8_t* in = (i == 0) ? color_in : targets[last];
        vo_denoise_pass(in, normal, position, targets[ping], W, H, &p, mode, 0, H);
        last = ping;
    }
    return last;
}

/* ------------------------------------------------------------------------------------------ */
/* blit + jitter (rows "next" of SURVEY 8(f))                                                  */
/* ------------------------------------------------------------------------------------------ */

/* texture(inputImage, uv) with a linear / clamp-to-edge sampler on an RGBA8_UNORM image */
static void sample_rgba8_linear(const uint8_t* img, int w, int h, float u, float v, float out[4])
{
    float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = clampi((int)x0f, 0, w - 1), x1 = clampi((int)x0f + 1, 0, w - 1);
    int y0 = clampi((int)y0f, 0, h - 1), y1 = clampi((int)y0f + 1, 0, h - 1);
    for (int k = 0; k < 4; k++) {
        float c00 = (float)img[((size_t)y0 * w + x0) * 4 + k] / 255.0f, c10 = (float)img[((size_t)y0 * w + x1) * 4 + k] / 255.0f;
        float c01 = (float)img[((size_t)y1 * w + x0) * 4 + k] / 255.0f, c11 = (float)img[((size_t)y1 * w + x1) * 4 + k] / 255.0f;
        float a = c00 + tx * (c10 - c00), b = c01 + tx * (c11 - c01);
        out[k] = a + ty * (b - a);
    }
}

/* blit.frag:14-22 */
void vo_blit(const uint8_t* src, int sw, int sh, uint8_t* dst, int tw, int th)
{
    float sx = (float)sw, sy = (float)sh, txs = (float)tw, tys = (float)th;
    float scale = fminf(sx / txs, sy / tys);
    float stx = txs * scale, sty = tys * scale;                    /* scaledTarget */
    for (int py = 0; py < th; py++)
        for (int px = 0; px < tw; px++) {
            float vx = ((float)px + 0.5f) / txs, vy = ((float)py + 0.5f) / tys;   /* vScreenPos */
            float tpx = vx * txs, tpy = vy * tys;                                   /* targetPos  */
            float spx = tpx * scale + (sx - stx) / 2.0f, spy = tpy * scale + (sy - sty) / 2.0f;
            float c[4];
            sample_rgba8_linear(src, sw, sh, spx / sx, spy / sy, c);
            for (int k = 0; k < 4; k++) dst[((size_t)py * tw + px) * 4 + k] = vo_unorm8(c[k]);
        }
}

int vo_jitter_phase_count(int render_width, int display_width)
{
    float r = (float)display_width / (float)render_width;
    return (int)(8.0f * (r * r));                                 /* powf(r, 2.0f) == r*r exactly */
}

static float halton(int index, int base)
{
    float f = 1.0f, result = 0.0f;
    for (int i = index; i > 0; i = i / base) {
        f /= (float)base;
        result += f * (float)(i % base);
    }
    return result;
}

void vo_jitter_offset(int index, int phase_count, float* jx, float* jy)
{
    int i = (index % phase_count) + 1;
    *jx = halton(i, 2) - 0.5f;
    *jy = halton(i, 3) - 0.5f;
}
