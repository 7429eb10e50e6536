/*
 * vrt_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A scalar, plain-C restatement of the reference's per-pixel hot path:
 *   /root/reference/shader/voxel_volume.frag:68-346   (ray gen, box clip, DDA, AO, shadow,
 *                                                     mirror bounces, sky, G-buffer write)
 *   /root/reference/shader/screen_quad.vert:18-31     (pixel -> vScreenPos)
 *   /root/reference/shader/denoiser.frag:38-73        (a-trous cross-bilateral filter)
 *   /root/reference/source/voxels/stages/denoiser_stage.cpp:37-61,143-154,204-257
 *                                                     (kernel weights/offsets, pass schedule)
 *
 * PARITY UNPINNED: the reference is GLSL-for-Vulkan, ships no golden vectors, no tests on this
 * path, and cannot be executed in the build container (no glslc / Vulkan / GPU).  This oracle is
 * therefore pinned only by hand-computed known-answer tests (tests/test_oracle_kat.py) and by the
 * canonical resolutions of the shader's undefined corners listed in SURVEY.md section 9.4 and
 * DESIGN.md.  The .vox ingestion side IS pinned: oracle/ref_vox builds the reference's own
 * thirdparty/opengametools/include/ogt_vox.h from where it lies under /root/reference.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The shipped library (voxel-raytracing_amd/csrc) never links, includes or calls it.
 */
#ifndef VRT_ORACLE_H
#define VRT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_MAX_BOUNCES 8

/* ScreenQuadPush, source/voxels/resource/screen_quad_push.hpp:5-15 (96 bytes, same offsets). */
typedef struct vo_push {
    float    cam_pos[4];
    float    cam_dir[4];
    float    cam_right[4];
    float    cam_up[4];
    uint32_t volume_bounds[3];
    uint32_t frame;
    int32_t  screen_size[2];
    float    camera_jitter[2];
} vo_push;

/* Material, source/voxels/resource/material.hpp:5-12 (32 bytes). */
typedef struct vo_material {
    float diffuse[4];
    float metallic;
    float pad[3];
} vo_material;

typedef struct vo_scene {
    const uint8_t*     voxels;      /* dense R8, index x + y*W + z*W*H  (voxel_scene.cpp:99) */
    uint32_t           dims[3];     /* W,H,D */
    const vo_material* palette;     /* 256 entries */
    const float*       sky;         /* RGBA32F, row-major, sky_w*sky_h*4 */
    uint32_t           sky_w, sky_h;
    const uint8_t*     noise;       /* RGBA8, noise_w*noise_h*4 */
    uint32_t           noise_w, noise_h;
    /* The same volume stored sparsely (voxels == NULL): 8^3 bricks.  brick_grid[bx + by*nbx + bz*nbx*nby] = 0 for an empty
     * brick, else 1 + index into brick_pool (512 bytes per brick, voxel (x,y,z) of the brick at x + 8y + 64z).  A storage
     * format only: getVoxel returns what the dense texture of the same content would (BASELINE configs[4]: a 2048^3
     * volume is 8 GiB dense). */
    const uint32_t*    brick_grid;
    const uint8_t*     brick_pool;
} vo_scene;

/* Parameters + Light UBOs (voxel_volume.frag:57-65) and the shader's compile-time constants
 * (voxel_volume.frag:68-69,219) promoted to runtime fields. */
typedef struct vo_params {
    uint32_t ao_samples;        /* default 4 */
    float    ambient_intensity; /* default 1 */
    float    light_dir[3];      /* default normalize(1,1,1) */
    float    light_intensity;   /* default 1 */
    float    light_color[4];    /* default 1,1,1,1 */
    uint32_t max_steps;         /* MAX_RAY_STEPS = 512 */
    uint32_t ao_steps;          /* 64 */
    uint32_t max_bounces;       /* MAX_REFLECTIONS = 5 (<= VO_MAX_BOUNCES) */
    uint32_t shadows;           /* 1 = reference behaviour; 0 = "primary rays only" mode */
} vo_params;

/* All planes optional (NULL = skip).  Row-major, W*H pixels, origin top-left. */
typedef struct vo_frame {
    float*    color_f;      /* 3 floats / px, shading result before UNORM8 quantisation      */
    uint8_t*  color8;       /* RGBA8_UNORM, alpha = 0   (geometry_stage.cpp:22)               */
    float*    depth;        /* R32F                      (geometry_stage.cpp:24)               */
    float*    motion;       /* RG32F = 0                 (geometry_stage.cpp:26)               */
    uint8_t*  mask8;        /* R8_UNORM 0.9 / 0          (geometry_stage.cpp:28)               */
    float*    position;     /* RGBA32F, w = 0            (geometry_stage.cpp:30)               */
    int8_t*   normal8;      /* RGBA8_SNORM, w = 0        (geometry_stage.cpp:32)               */
    uint8_t*  hit_id;       /* primary-ray material id (0 = miss)                             */
    int16_t*  hit_voxel;    /* 3 / px: mapPos at the hit (0,0,0 on miss)                      */
    uint8_t*  hit_mask;     /* bit0..2 = final DDA mask x,y,z (0 on miss)                     */
    uint32_t* steps_primary;/* executions of the voxel fetch (frag:157) by the primary ray    */
    uint32_t* steps_total;  /* ... by all rays of the pixel                                   */
    uint32_t* rays_total;   /* rays traced for the pixel (primary + AO + shadow + bounce)     */
} vo_frame;

/* Render rows [row0,row1) of the W x H frame (W,H from push->screen_size). */
void vo_render_rows(const vo_scene* sc, const vo_push* pc, const vo_params* pr,
                    const vo_frame* out, int row0, int row1);

/* Same, rows interleaved over nthreads pthreads (thread t takes rows t, t+n, ...). */
void vo_render_mt(const vo_scene* sc, const vo_push* pc, const vo_params* pr,
                  const vo_frame* out, int nthreads);

/* Trace a single ray (traceRay, frag:176-196) -- used by the known-answer tests. */
typedef struct vo_hit {
    uint32_t material;
    float    pos[3];
    float    normal[3];
    float    dir[3];
    int32_t  voxel[3];
    uint32_t mask;       /* bit0..2 */
    uint32_t steps;      /* voxel fetches */
    float    p0[3];      /* boxIntersection() result */
    float    side[3];    /* sideDist at loop exit */
    float    delta[3];
} vo_hit;
void vo_trace_ray(const vo_scene* sc, const float start[3], const float dir[3],
                  uint32_t max_steps, vo_hit* out);

/* Primary ray for pixel (px,py): voxel_volume.frag:312-322. */
void vo_primary_ray(const vo_push* pc, int px, int py, float start[3], float dir[3]);

/* Denoiser ---------------------------------------------------------------------------------- */
#define VO_DENOISE_CANONICAL 0   /* intended 9-tap filter (SURVEY 9.4-D canonical)                */
#define VO_DENOISE_AS_SHIPPED 1  /* std140 aliasing of the tightly packed UBOs, OOB reads = 0     */

typedef struct vo_denoise_params {  /* DenoiserParams, denoiser_stage.cpp:14-20 */
    float phi_color, phi_normal, phi_pos, step_width;
} vo_denoise_params;

/* Per-pass parameters as denoiser_stage.cpp:143-154 derives them. */
void vo_denoise_pass_params(int pass, float phi_color0, float phi_normal0, float phi_pos0,
                            float step_width0, vo_denoise_params* out);

/* One pass (denoiser.frag:38-73) over rows [row0,row1).  color_in/out RGBA8_UNORM, normal RGBA8_SNORM,
 * position RGBA32F; all W*H.  Guides are point-sampled for integer stepWidth, bilinear otherwise
 * (render_image.cpp:61-66: linear filter, clamp-to-edge). */
void vo_denoise_pass(const uint8_t* color_in, const int8_t* normal, const float* position,
                     uint8_t* color_out, int W, int H, const vo_denoise_params* p, int mode,
                     int row0, int row1);

/* Whole schedule (denoiser_stage.cpp:204-257): ping-pong `iterations` passes; returns the index
 * (0/1) of the scratch target that holds the final image (or -1 if iterations == 0). */
int vo_denoise(const uint8_t* color_in, const int8_t* normal, const float* position,
               uint8_t* target0, uint8_t* target1, int W, int H, int iterations,
               float phi_color0, float phi_normal0, float phi_pos0, float step_width0, int mode);

/* blit.frag:14-22 (BlitStage, source/voxels/stages/blit_stage.cpp:41-43): centre-crop / scale the source image into
 * the target with a linear, clamp-to-edge sampler (render_image.cpp:61-66).  RGBA8 in, RGBA8 out. */
void vo_blit(const uint8_t* src, int sw, int sh, uint8_t* dst, int tw, int th);

/* Jitter sequence of UpscalerStage::update (upscaler_stage.cpp:59-70) with FidelityFX-FSR2's published helpers
 * (ffxFsr2GetJitterPhaseCount = int(8 * (display/render)^2), ffxFsr2GetJitterOffset = Halton(2,3) - 0.5 in pixels). */
int  vo_jitter_phase_count(int render_width, int display_width);
void vo_jitter_offset(int index, int phase_count, float* jx, float* jy);

/* Math primitives exported for the accuracy tests. */
float vo_atan2f(float y, float x);
float vo_asinf(float x);
float vo_expf(float x);
uint8_t vo_unorm8(float c);
int8_t  vo_snorm8(float c);

#ifdef __cplusplus
}
#endif
#endif
