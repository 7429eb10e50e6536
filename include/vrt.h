/*
 * vrt.h -- C-ABI of the MI355X-native voxel ray tracer (libvrt_hip.so).
 *
 * The reference (ectucker1/voxel-raytracing) has no FFI layer; the boundary this library replaces is
 * the C++ object surface of source/voxels + source/engine.  Each entry point cites the reference
 * interface it stands in for (paths relative to the reference root).  Plain pointers and sizes only;
 * no exceptions cross the ABI: every call returns VRT_OK or an error code and vrt_last_error()
 * returns the thread-local message (the reference throws std::runtime_error, app.cpp:21-25).
 *
 * Threading: one vrt_ctx per host thread / per GPU; calls on one context are serialised on its HIP
 * stream (the reference is strictly single-threaded, engine.cpp:28-46).  Frames in flight
 * (MAX_FRAMES_IN_FLIGHT, engine.hpp:19): one context per frame slot; contexts of one device run
 * concurrently and may share a vrt_scene, which is read-only while rendering.  All image pointers in
 * vrt_frame are DEVICE pointers owned by the caller (or allocated with vrt_device_alloc).
 * There is no CPU fallback: without a HIP device every compute call fails with VRT_ERR_NO_DEVICE.
 */
#ifndef VRT_H
#define VRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRT_OK               0
#define VRT_ERR_INVALID      1   /* bad argument */
#define VRT_ERR_IO           2   /* "Failed to read voxel scene"            voxel_scene.cpp:42 */
#define VRT_ERR_PARSE        3   /* "Could not parse voxel scene"           voxel_scene.cpp:46 */
#define VRT_ERR_NO_INSTANCE  4   /* "Voxel scene does not contain an instance." voxel_scene.cpp:50 */
#define VRT_ERR_NO_DEVICE    5   /* no HIP device / HIP runtime error at context creation */
#define VRT_ERR_HIP          6   /* HIP runtime error (message has the HIP error string) */
#define VRT_ERR_UNSUPPORTED  7

#define VRT_MAX_BOUNCES 8

typedef struct vrt_ctx   vrt_ctx;    /* device + stream + workspace  (stands in for Engine, engine.hpp:71-76) */
typedef struct vrt_scene vrt_scene;  /* device-resident scene        (VoxelScene, voxel_scene.hpp:18-34)      */

/* Material, source/voxels/resource/material.hpp:5-12 (32 B, std140 compatible). */
typedef struct vrt_material {
    float diffuse[4];
    float metallic;
    float pad[3];
} vrt_material;

/* ScreenQuadPush, source/voxels/resource/screen_quad_push.hpp:5-15; filled at voxel_renderer.cpp:72-83.
 * 96 bytes, identical member offsets (camPos@0 camDir@16 camRight@32 camUp@48 volumeBounds@64 frame@76
 * screenSize@80 cameraJitter@88). */
typedef struct vrt_push {
    float    cam_pos[4];
    float    cam_dir[4];
    float    cam_right[4];
    float    cam_up[4];
    uint32_t volume_bounds[3];
    uint32_t frame;
    int32_t  screen_size[2];
    float    camera_jitter[2];
} vrt_push;

/* Traversal strategies.  All of them produce bit-identical hit records and G-buffers (the DDA state is
 * advanced with the same fp32 additions as voxel_volume.frag:164-170); they differ only in how much memory
 * work is skipped.  steps_* planes are exact for DENSE / BITMASK / DF and upper bounds for JUMP / DFJ. */
#define VRT_TRAVERSAL_AUTO    0   /* fastest available (currently DF) */
#define VRT_TRAVERSAL_DENSE   1   /* one R8 fetch per DDA iteration (voxel_volume.frag:157), fetched 4 iterations ahead */
#define VRT_TRAVERSAL_BITMASK 2   /* solid test on 4^3 occupancy words (L2) + 16^3 / 64^3 summaries staged in LDS */
#define VRT_TRAVERSAL_JUMP    3   /* BITMASK + exact closed-form jumps across empty pyramid cells */
#define VRT_TRAVERSAL_DF      4   /* octant clearance, its minimum over the wave agreed by a DPP reduction; ALU-only runs between look-ups */
#define VRT_TRAVERSAL_DFJ     5   /* DF with the long runs (clearance >= 12) done per lane in closed form (JUMP's integer form) */
/* (6, 7: internal -- the brick-scene march and the hand-written DF loop; both are what AUTO resolves to where they apply) */

#define VRT_FLAG_SPLIT_KERNELS 4u /* trace secondary rays in a second kernel (K2) over the compacted hit list instead of inside K1 */
#define VRT_FLAG_DEBUG_PLANES 1u  /* steps_total / rays_total receive traversal diagnostics instead (development aid) */
#define VRT_FLAG_MARCHED_COUNTS 16u /* the count planes (steps_primary, steps_total, rays_total) report the PRODUCT march's own work -- rays
                                     * end at open cells, untagged blocks are not traced, an any-hit ray decided at a look-up reports the
                                     * iterations it took -- instead of the reference loop's iterations; every other plane is unchanged.
                                     * bench.py's roofline figures count these.  The launch runs the counting twins of the look-up loops: the same
                                     * march as the product launch (threshold runs included: a threshold run reports the steps its axes took,
                                     * which is the iteration count unless two axes tie) */
#define VRT_FLAG_LOOKUP_COUNTS 32u  /* with VRT_FLAG_MARCHED_COUNTS: the count planes hold the BYTES the march asked for instead of its iterations --
                                     * one per clearance look-up of a live lane (three where the look-ups prefetch), one per voxel id read; brick
                                     * scenes: 8 per brick word, 1 per fine byte and id.  What bench.py reports as roofline.requested_bytes */

/* VolumeParameters (parameters.hpp:5-9) + Light (voxel_scene.hpp:10-15) as GeometryStage::record fills
 * them each frame (geometry_stage.cpp:135-145), plus the shader's compile-time constants
 * (voxel_volume.frag:68-69,219) promoted to runtime fields. */
typedef struct vrt_settings {
    uint32_t ao_samples;        /* occlusionSettings.numSamples, default 4   */
    float    ambient_intensity; /* occlusionSettings.intensity,  default 1   */
    float    light_dir[3];      /* lightSettings.direction, default normalize(1,1,1) */
    float    light_intensity;   /* default 1 */
    float    light_color[4];    /* default (1,1,1,1) */
    uint32_t max_steps;         /* MAX_RAY_STEPS   = 512 */
    uint32_t ao_steps;          /* AO ray step cap = 64  */
    uint32_t max_bounces;       /* MAX_REFLECTIONS = 5 (<= VRT_MAX_BOUNCES) */
    uint32_t shadows;           /* 1 = reference behaviour; 0 = "primary rays only" (no shadow ray) */
    uint32_t traversal;         /* VRT_TRAVERSAL_* */
    uint32_t flags;
} vrt_settings;

/* GeometryBuffer, source/voxels/stages/geometry_stage.hpp:19-27, target formats geometry_stage.cpp:22-33.
 * Row-major, origin top-left.  Any plane may be NULL (not written).  All DEVICE pointers.
 * In sharded mode (vrt_shard.nranks > 1) the planes are still full-frame W*H; a rank writes only
 * the rows it owns. */
typedef struct vrt_frame {
    uint8_t*  color8;        /* RGBA8_UNORM, alpha = 0                     */
    float*    depth;         /* R32F                                        */
    float*    motion;        /* RG32F (always 0, voxel_volume.frag:333)     */
    uint8_t*  mask8;         /* R8_UNORM: 0.9 on hit / 0                    */
    float*    position;      /* RGBA32F, w = 0                              */
    int8_t*   normal8;       /* RGBA8_SNORM, w = 0                          */
    /* debug / parity planes (no reference analogue) */
    float*    color_f;       /* 3 floats / px before UNORM8 quantisation    */
    uint8_t*  hit_id;        /* primary-ray material id, 0 = miss           */
    int16_t*  hit_voxel;     /* 3 / px: grid cell of the hit                */
    uint8_t*  hit_mask;      /* bit0..2 = final DDA mask x,y,z              */
    /* The two count planes report the iterations of the REFERENCE's loop (every ray walks to a hit, the wall or the end of its
     * budget).  The product march takes fewer -- rays end where nothing solid is left in their octant, blocks of pixels no
     * occupied cell projects onto are not traced -- so a launch with a count plane marches a second set of clearance fields
     * without those shortcuts (built on first use, as large as the first) and uses no tile tags: diagnostics, not for
     * production launches.  Every other plane is the same either way. */
    uint32_t* steps_primary; /* DDA iterations that sampled a voxel (frag:157), primary ray  */
    uint32_t* steps_total;   /* ... all rays of the pixel                   */
    uint32_t* rays_total;    /* rays traced for the pixel                   */
    /* multi-GPU (no reference analogue): the colour once more, in the layout vrt_pack_rows produces -- the rows this rank
     * owns, packed in increasing row order, vrt_shard_rows() rows in all (padding rows are not written).  Lets the
     * tracing kernel fill the send buffer of the collective itself instead of a copy kernel after it. */
    uint8_t*  color8_strips;
} vrt_frame;

/* Screen-tile sharding (no reference analogue; BASELINE north_star).  The frame is cut into
 * horizontal strips of strip_rows rows (multiple of 16); strip s belongs to rank s % nranks.
 * nranks = 1 (or a NULL vrt_shard*) renders everything. */
typedef struct vrt_shard {
    int32_t rank;
    int32_t nranks;
    int32_t strip_rows;
} vrt_shard;

/* ---- context --------------------------------------------------------------------------------- */
/* Engine::init (engine.cpp:14): selects HIP device `device`, creates the context stream. */
int  vrt_ctx_create(int device, vrt_ctx** out);
void vrt_ctx_destroy(vrt_ctx* ctx);                         /* Engine::destroy */
/* Adopt an externally owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) in place of the
 * context's own stream.  NULL is a valid handle: the HIP null (legacy default) stream, which is what
 * torch's default stream is. */
int  vrt_ctx_set_stream(vrt_ctx* ctx, void* hip_stream);
int  vrt_ctx_synchronize(vrt_ctx* ctx);                     /* device.waitIdle(), engine.cpp:351 */
/* Development switches of a context (no reference analogue; the reference's counterpart is recompiling a shader): each changes
 * speed only, never a result -- the tests render "the same frame without X" with them.  Name (default), who looks at it:
 *   every vrt_render_geometry* call:  "tile_tags" (1), "box_rect" (1), "fast_loop" (1), "thresh_runs" (1), "hit_table" (1), "sky_fast" (1),
 *                                     "no_bounce_kernel" (1), "ao_batch" (1: the AO rays of a wave from a pool in LDS every lane draws on),
 *                                     "packed_bounces" (1: the bounce chain as one word per hit on dense scenes; 2: on brick scenes too,
 *                                     where it measured slower; 0: the stack of hits), "tags_async" (0: the tile tags of a launch on a
 *                                     stream of their own measured slower),
 *                                     "xcd_regions" (0 -- until round 3 VRT_XCD_REGIONS was on by default; as a three-dimensional grid the
 *                                     form ran 30.0 or 33.6 us per bench frame from one process to the next, so it is opt-in now)
 *   vrt_denoise:                      "denoise_packed" (1), "denoise_verified" (1), "denoise_pair" (1: verified passes of the canonical taps
 *                                     with offsets 2 .. 5 compute every weight once, k_denoise_pair; 0: k_denoise_ver), "denoise_p0" (1: pass 0 through k_denoise_p0), "denoise_pair_wgs" (0; experiments: workgroups of a
 *                                     k_denoise_pair launch), "denoise_th16" (0: the tolerance kernel's 64 x 16 tiles
 *                                     are an experiment), and the tests' handles on the verified pass "denoise_guard_div8" (0), "denoise_count" (0)
 *   scene creation:                   "open_cells" (1), "df_prefetch" (1), "df_own" (1)
 * The environment seeds them ONCE, at vrt_ctx_create (VRT_TILE_TAGS=0, VRT_SKY_FAST=0, ...); nothing on the render path calls
 * getenv.  Values are non-negative integers (the switches: 0 / 1; a negative value is stored as 0).  Unknown name: VRT_ERR_INVALID. */
int  vrt_ctx_set_option(vrt_ctx* ctx, const char* name, int32_t value);
int  vrt_ctx_get_option(vrt_ctx* ctx, const char* name, int32_t* value);
const char* vrt_last_error(void);
/* Name of the device, compute units, and whether the library was built for its gfx arch. */
int  vrt_device_info(vrt_ctx* ctx, char* name, size_t name_len, int* compute_units);

/* Device memory helpers so that C/C++ callers need no HIP headers (Buffer, engine/resource/buffer.hpp). */
int  vrt_device_alloc(vrt_ctx* ctx, size_t bytes, void** out);
int  vrt_device_free(vrt_ctx* ctx, void* p);
int  vrt_memcpy_h2d(vrt_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int  vrt_memcpy_d2h(vrt_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int  vrt_memset(vrt_ctx* ctx, void* dst_dev, int value, size_t bytes);

/* ---- scene ----------------------------------------------------------------------------------- */
/* VoxelScene::VoxelScene(engine, filename, skybox) (voxel_scene.cpp:33-133): read the file, parse the
 * MagicaVoxel chunks (own reader, validated against ogt_vox.h:1179-1991), flatten all instances of
 * frame 0 into one dense R8 volume with the Y/Z swap (voxel_scene.cpp:53-105), build the palette
 * (pow 2.2 + MATL metal, :108-117), upload, and build the occupancy pyramid. */
int  vrt_scene_load_vox_file(vrt_ctx* ctx, const char* path, vrt_scene** out);
/* ogt_vox_read_scene(buffer,size) + the same flatten. */
int  vrt_scene_load_vox_mem(vrt_ctx* ctx, const void* buf, size_t n, vrt_scene** out);
/* Synthetic scenes: dense R8 volume, index x + y*W + z*W*H (Texture3D upload order, voxel_scene.cpp:99,122). */
int  vrt_scene_from_dense(vrt_ctx* ctx, const uint8_t* voxels, uint32_t W, uint32_t H, uint32_t D,
                          const vrt_material palette[256], vrt_scene** out);
/* The same volume handed over SPARSELY, in 8^3 bricks (no reference analogue: the reference's Texture3D is dense,
 * voxel_scene.cpp:77-78,122; BASELINE configs[4] is a 2048^3 volume -- 8 GiB dense, 9x that with the dense scene's clearance
 * fields).  grid[bx + by*nbx + bz*nbx*nby] = 0 for an empty brick, else 1 + its index in `pool`; pool holds n_bricks x 512
 * voxel ids, voxel (x,y,z) of a brick at x + 8y + 64z.  The scene is the W x H x D = 8nbx x 8nby x 8nbz volume those bricks
 * spell out, and renders bit for bit like vrt_scene_from_dense of the same content; device memory is per OCCUPIED brick
 * (4.5 KiB: ids + per-voxel clearance) plus 12 bytes per brick of the grid.  Only VRT_TRAVERSAL_AUTO applies to it. */
int  vrt_scene_from_bricks(vrt_ctx* ctx, const uint32_t* grid, uint32_t nbx, uint32_t nby, uint32_t nbz,
                           const uint8_t* pool, uint32_t n_bricks, const vrt_material palette[256], vrt_scene** out);
/* Drops what a scene holds only for diagnostics: the second set of clearance fields (without open cells) that the first launch
 * with a count plane (steps_primary / steps_total) builds -- as large as the first, 9 x the padded voxel bytes.  A later launch
 * with count planes builds it again.  Callers that never attach count planes never hold it.  Waits for the context's stream. */
int  vrt_scene_trim(vrt_ctx* ctx, vrt_scene* scene);
/* Device bytes a scene holds (volume, clearance, pyramid / brick structures, palette, sky, noise). */
int  vrt_scene_memory(const vrt_scene* sc, uint64_t* bytes);
/* Host-only half of the loader (no device needed): parse + flatten into malloc'd host memory.
 * *voxels must be released with vrt_host_free.  Used by tests and by the C++ host mirror. */
int  vrt_vox_flatten_host(const void* buf, size_t n, uint32_t dims[3], uint8_t** voxels,
                          vrt_material palette[256], uint32_t* num_instances, uint64_t* dropped);
void vrt_host_free(void* p);
/* Texture2D(skybox .hdr, RGBA32F) (voxel_scene.cpp:132; nearest/repeat sampler texture_2d.cpp:158-163):
 * raw float RGBA texels, row-major.  Default when never called: 1x1 white. */
int  vrt_scene_set_sky(vrt_ctx* ctx, vrt_scene* sc, const float* rgba, uint32_t w, uint32_t h);
/* Texture2D(blue_noise_rgba.png, RGBA8_UNORM) (voxel_renderer.cpp:22).  Default: 1x1 mid-grey. */
int  vrt_scene_set_blue_noise(vrt_ctx* ctx, vrt_scene* sc, const uint8_t* rgba8, uint32_t w, uint32_t h);
/* Texture2D(engine, filepath, 4, format) (source/engine/resource/texture_2d.cpp:22-44): decode the file on the host
 * (Radiance .hdr -> linear float RGBA like stbi_loadf, PNG -> RGBA8 like stbi_load) and upload it.  An 8-bit image given
 * as sky is converted c/255; a float image given as noise is rejected.  Failure: "Could not load image <path>". */
int  vrt_scene_set_sky_file(vrt_ctx* ctx, vrt_scene* sc, const char* path);
int  vrt_scene_set_blue_noise_file(vrt_ctx* ctx, vrt_scene* sc, const char* path);
/* The decoders alone (host only).  *pixels: float RGBA (is_hdr = 1) or RGBA8 (is_hdr = 0), release with vrt_host_free. */
int  vrt_image_load(const char* path, int* is_hdr, uint32_t* w, uint32_t* h, void** pixels);
/* Writers for rendered frames (host memory): 8-bit PNG / binary PPM from RGBA8, PFM from float RGB(A) with the given
 * stride in floats (3 for color_f, 4 for position). */
int  vrt_image_write_png(const char* path, const uint8_t* rgba8, uint32_t w, uint32_t h);
int  vrt_image_write_ppm(const char* path, const uint8_t* rgba8, uint32_t w, uint32_t h);
int  vrt_image_write_pfm(const char* path, const float* pixels, uint32_t w, uint32_t h, uint32_t stride_floats);
/* VoxelScene::width/height/depth (voxel_scene.hpp:21). */
int  vrt_scene_info(const vrt_scene* sc, uint32_t dims[3]);
/* Copy the dense volume / palette back to the host (tests, oracle comparison). */
int  vrt_scene_download(vrt_ctx* ctx, const vrt_scene* sc, uint8_t* voxels, vrt_material palette[256]);
void vrt_scene_free(vrt_ctx* ctx, vrt_scene* sc);            /* VoxelScene::destroy */

/* ---- settings -------------------------------------------------------------------------------- */
/* Defaults of VoxelRenderSettings (voxel_render_settings.hpp:21-42) + shader constants. */
void vrt_settings_default(vrt_settings* s);

/* ---- geometry stage -------------------------------------------------------------------------- */
/* GeometryStage::record (geometry_stage.cpp:106-153) == one full-screen run of voxel_volume.frag:
 * primary-ray DDA kernel + (when AO / shadow / reflection rays are enabled) the secondary-ray and
 * shading kernels.  Asynchronous on the context stream.
 * Limits: screen_size up to 32768 per side and below 2^28 pixels per frame (16384 x 16384 is refused with
 * VRT_ERR_UNSUPPORTED): a launch addresses the full-frame planes with 32-bit byte offsets -- 16 B per pixel in the
 * position plane -- and sharded launches (vrt_shard) index the same full-frame planes, so strips do not lift the limit;
 * a larger image is rendered as several frames whose camera planes are shifted. */
int  vrt_render_geometry(vrt_ctx* ctx, const vrt_scene* sc, const vrt_push* push,
                         const vrt_settings* settings, const vrt_frame* frame, const vrt_shard* shard);

/* The same for n frames of one scene, one screen size and one set of settings -- consecutive camera poses of an
 * animation (App::run's frame loop, source/app.cpp:18-27, rendered offline), or the frames of a multi-GPU batch --
 * in ONE launch (per 256 frames): the tiles of frame f+1 are dispatched while frame f drains, so the tail of a frame
 * (tens of microseconds during which most of the GPU idles) is paid once per launch instead of once per frame.
 * Up to 8 frames travel in the kernel arguments; larger batches go through a small table in device memory that is
 * uploaded on a stream of its own.  pushes[n], frames[n]; every frame needs its own output planes.  Results are
 * identical to n single calls. */
int  vrt_render_geometry_batch(vrt_ctx* ctx, const vrt_scene* sc, int32_t n, const vrt_push* pushes,
                               const vrt_settings* settings, const vrt_frame* frames, const vrt_shard* shard);

/* The same with one strip assignment PER FRAME (shards[n]; all with the same nranks and strip_rows): a rank of a
 * multi-GPU batch traces frame block b with the assignment of rank (rank + b) % nranks, so that rows which do not divide
 * evenly over the ranks still cost every rank the same per step -- in one launch.  No reference analogue. */
int  vrt_render_geometry_slots(vrt_ctx* ctx, const vrt_scene* sc, int32_t n, const vrt_push* pushes,
                               const vrt_settings* settings, const vrt_frame* frames, const vrt_shard* shards);

/* ---- denoiser stage -------------------------------------------------------------------------- */
#define VRT_DENOISE_CANONICAL  0  /* the intended 9-tap a-trous filter                               */
#define VRT_DENOISE_AS_SHIPPED 1  /* the std140-aliased 3-tap filter the shipped UBO upload produces  */
#define VRT_DENOISE_FAST       2  /* flag, OR-ed into either: the weighted passes (pass >= 1, integral stepWidth) evaluate the three
                                   * edge-stopping weights as ONE hardware exponential, exp2(-(dc2*kc + dn2*kn + dp2*kp)), and divide
                                   * by multiplying with the hardware reciprocal -- the shader's own freedom (GLSL exp / division are
                                   * not correctly rounded either).  Stated bound against the exact mode and the oracle: at most ONE
                                   * RGBA8 code per channel per pass (tests/test_gpu_denoise.py); the exact mode stays the default. */

/* DenoiserSettings, voxel_render_settings.hpp:21-29. */
typedef struct vrt_denoiser_settings {
    int32_t iterations;     /* default 2, max 10 (MAX_DENOISER_PASSES, denoiser_stage.hpp:9) */
    float   phi_color0;     /* 20.4 */
    float   phi_normal0;    /* 1e-2 */
    float   phi_pos0;       /* 1e-1 */
    float   step_width;     /* 2.0  */
    int32_t mode;           /* VRT_DENOISE_* */
} vrt_denoiser_settings;
void vrt_denoiser_settings_default(vrt_denoiser_settings* s);

/* DenoiserStage::record(cmd, flightFrame, color, normal, pos) (denoiser_stage.cpp:156-258): runs
 * `iterations` ping-pong passes of denoiser.frag:38-73 with the per-pass parameters of
 * denoiser_stage.cpp:143-154.  target0/target1 are the two RGBA8 ping-pong images (_colorTargets);
 * *result receives the one holding the final image (color_in itself when iterations == 0).
 * With a shard, only the rank's own rows of the final image are valid; the guide planes must already
 * hold `vrt_denoise_halo_rows()` valid rows either side of every owned strip. */
int  vrt_denoise(vrt_ctx* ctx, int32_t W, int32_t H, const vrt_denoiser_settings* ds,
                 const uint8_t* color_in, const int8_t* normal8, const float* position,
                 uint8_t* target0, uint8_t* target1, const vrt_shard* shard, const uint8_t** result);
/* Sum of the per-pass tap reach ceil(stepWidth_i) over all passes: rows a strip needs from its neighbours. */
int  vrt_denoise_halo_rows(const vrt_denoiser_settings* ds);
/* A pass with an integral tap offset (<= 5; whole frames and sharded passes alike) is computed in two steps: a cheap evaluation of
 * every pixel (hardware exponential and reciprocal, integer code distances) and a second, literal one -- the operations of
 * denoiser.frag:48-72 as the numeric spec fixes them -- of the pixels whose cheap value lies within `guard` RGBA8 codes of a
 * rounding boundary of the RGBA8 target (csrc/vrt_denoise_bound.h derives the guard).  The image is the literal evaluation's,
 * bit for bit.  vrt_denoise_guard reports the guard of a pass (+inf: the pass is computed literally throughout; needs no
 * device); vrt_debug_denoise_redone how many pixels of a pass the latest vrt_denoise call on the context evaluated twice
 * (counted only while the context option "denoise_count" is 1). */
int  vrt_denoise_guard(const vrt_denoiser_settings* ds, int32_t pass, float* guard);
int  vrt_debug_denoise_redone(vrt_ctx* ctx, int32_t pass, uint32_t* pixels);

/* ---- multi-GPU strip packing (feeds the RCCL gather; no reference analogue) -------------------- */
/* Number of rows rank owns. */
int  vrt_shard_rows(int32_t H, const vrt_shard* shard);
/* Pack the rows a rank owns (in increasing row order) of a full-frame plane with `bytes_per_px` into a
 * contiguous buffer, or scatter them back (root side, after the gather).  Device pointers. */
int  vrt_pack_rows(vrt_ctx* ctx, const void* full, void* packed, int32_t W, int32_t H,
                   int32_t bytes_per_px, const vrt_shard* shard);
int  vrt_unpack_rows(vrt_ctx* ctx, const void* packed, void* full, int32_t W, int32_t H,
                     int32_t bytes_per_px, const vrt_shard* shard);
/* The same for n images in one launch per 64 images (host arrays of n device pointers): the frames of a batch on a
 * rank (one shard for all), and at the root of the gather frames x source ranks (shards[n], one per image). */
int  vrt_pack_rows_batch(vrt_ctx* ctx, int32_t n, const void* const* full, void* const* packed, int32_t W, int32_t H,
                         int32_t bytes_per_px, const vrt_shard* shard);
int  vrt_unpack_rows_batch(vrt_ctx* ctx, int32_t n, const void* const* packed, void* const* full, int32_t W, int32_t H,
                           int32_t bytes_per_px, const vrt_shard* shards);
/* Pack / unpack the halo rows exchanged with the ring neighbours before a sharded denoise:
 * dir = -1: the first `halo` rows of every owned strip (sent to the rank owning the strip above),
 * dir = +1: the last `halo` rows of every owned strip (sent to the rank owning the strip below).
 * vrt_unpack_halo takes the SENDER's shard and the same dir: the rows are written back at their true
 * frame positions in the receiver's full-frame plane (i.e. just outside the receiver's own strips).
 * vrt_unpack_rows likewise takes the shard of the rank that packed the buffer. */
int  vrt_pack_halo(vrt_ctx* ctx, const void* full, void* packed, int32_t W, int32_t H,
                   int32_t bytes_per_px, const vrt_shard* shard, int32_t halo, int32_t dir);
int  vrt_unpack_halo(vrt_ctx* ctx, const void* packed, void* full, int32_t W, int32_t H,
                     int32_t bytes_per_px, const vrt_shard* shard, int32_t halo, int32_t dir);
size_t vrt_halo_bytes(int32_t W, int32_t H, int32_t bytes_per_px, const vrt_shard* shard, int32_t halo);
/* The halo rows of n images in one launch per 64 (the frames of a batch; shards[n]: each image's own strip assignment --
 * for unpack the SENDER's): one ring exchange per step then carries the colour, normal and position rows of every frame. */
int  vrt_pack_halo_batch(vrt_ctx* ctx, int32_t n, const void* const* full, void* const* packed, int32_t W, int32_t H,
                         int32_t bytes_per_px, const vrt_shard* shards, int32_t halo, int32_t dir);
int  vrt_unpack_halo_batch(vrt_ctx* ctx, int32_t n, const void* const* packed, void* const* full, int32_t W, int32_t H,
                           int32_t bytes_per_px, const vrt_shard* shards, int32_t halo, int32_t dir);

/* ---- the collective itself, for hosts without torch.distributed (no reference analogue; BASELINE north_star: "C++ host code
 * ... RCCL gather over xGMI") ---------------------------------------------------------------------------------------------
 * RCCL (librccl.so, found with dlopen at first use: the library the process already has -- torch's own copy in a Python
 * process -- or /opt/rocm's) behind plain C: one communicator rank per vrt_ctx.  Two ways to build the ranks:
 *   one process per GPU:        rank 0 calls vrt_comm_unique_id, hands the 128 bytes to the others out of band (a file, MPI,
 *                               a socket), every rank calls vrt_comm_init_rank;
 *   one process, several GPUs:  vrt_comm_init_all over the contexts (the C++ App: `vrt_app --devices 0,1,...`); calls that
 *                               belong to different ranks of one step are then wrapped in vrt_group_start / vrt_group_end.
 * vrt_gather_strips: every rank's `bytes` at `send` arrive at `recv + r * bytes` on `root` (the packed strips of vrt_pack_rows:
 * root unpacks them with vrt_unpack_rows and rank r's vrt_shard).  Enqueued on the context's stream; device pointers. */
typedef struct vrt_comm vrt_comm;
int  vrt_comm_unique_id(uint8_t id[128]);
int  vrt_comm_init_rank(vrt_ctx* ctx, int32_t nranks, int32_t rank, const uint8_t id[128], vrt_comm** out);
int  vrt_comm_init_all(int32_t n, vrt_ctx* const* ctxs, vrt_comm** out /* n handles */);
void vrt_comm_destroy(vrt_comm* comm);
int  vrt_group_start(void);
int  vrt_group_end(void);
int  vrt_gather_strips(vrt_ctx* ctx, vrt_comm* comm, int32_t root, const void* send, void* recv, size_t bytes);

/* ---- presentation / temporal helpers ----------------------------------- */
/* Replaces BlitStage::record + shader/blit.frag:14-22 (source/voxels/stages/blit_stage.cpp:41-75): the RGBA8 source
 * is centre-cropped to the target's aspect ratio and resampled with a linear, clamp-to-edge sampler
 * (render_image.cpp:61-66).  Also the plain upscale of a reduced-resolution render
 * (voxel_render_settings.cpp:3-13) that stands in for the FSR2 dispatch (upscaler_stage.cpp:72-161, out of scope).
 * Device pointers; runs on the context's stream. */
int  vrt_blit(vrt_ctx* ctx, const void* src_rgba8, int32_t src_w, int32_t src_h,
              void* dst_rgba8, int32_t dst_w, int32_t dst_h);

/* N-frame accumulation of jittered frames, the offline stand-in for FSR2's temporal pass: accum is W*H*4 uint32
 * (device), holding exact sums of the UNORM8 codes; reset != 0 starts a new sequence with this frame.
 * vrt_resolve writes the mean of `frames` accumulated frames, rounded half up: (2*sum + frames) / (2*frames). */
int  vrt_accumulate(vrt_ctx* ctx, const void* color_rgba8, void* accum_u32, int32_t W, int32_t H, int32_t reset);
int  vrt_resolve(vrt_ctx* ctx, const void* accum_u32, void* out_rgba8, int32_t W, int32_t H, uint32_t frames);

/* Replace ffxFsr2GetJitterPhaseCount / ffxFsr2GetJitterOffset as called by UpscalerStage::update
 * (source/voxels/stages/upscaler_stage.cpp:59-70): phase count int(8 * (display_width / render_width)^2);
 * offset = Halton(2,3)(index % phase_count + 1) - 0.5, in pixels; feeds vrt_push.camera_jitter. */
int32_t vrt_jitter_phase_count(int32_t render_width, int32_t display_width);
int  vrt_jitter_offset(int32_t index, int32_t phase_count, float* jitter_x, float* jitter_y);

/* ---- instrumentation ------------------------------------------------------------------------- */
/* Time of the most recent vrt_render_geometry primary-ray kernel / all its kernels, and of the most
 * recent vrt_denoise, in milliseconds (HIP events on the context stream; blocks until they complete). */
int  vrt_last_timings(vrt_ctx* ctx, float* primary_ms, float* geometry_ms, float* denoise_ms);
/* Diagnostic of the sky-texel fast path (csrc/vrt_sky.h; skyColor, voxel_volume.frag:98-105): for n unnormalised
 * directions (device floats, xyz per direction) the texel skyColor(normalize(v)) reads under the numeric spec and the
 * one the fast path decides, both computed by the GPU: out[4i] = spec x | y << 16, out[4i+1] = fast x | y << 16,
 * out[4i+2] = 1 if the fast path is sure of its texel (only then may the two be compared), out[4i+3] = float bits of the
 * fast coordinate u * sky_w.  tests/test_gpu_sky.py sweeps it over 10^8 directions. */
int  vrt_debug_sky_texels(vrt_ctx* ctx, const vrt_scene* scene, const float* dirs_dev, size_t n, uint32_t* out_dev);
/* Development builds only (make variant EXTRA=-DVRT_TRACE_COUNTERS; the product library answers zeros): look-ups of the brick
 * march since the last call, summed over the lanes of all rays -- all, those inside an occupied brick (the second, dependent
 * load), those that found a solid voxel, those that ended in the border or an open brick.  Waits for the stream; resets. */
int  vrt_debug_brick_counts(vrt_ctx* ctx, uint64_t out[4]);
/* Enable/disable per-call event recording (default on). */
int  vrt_ctx_set_timing(vrt_ctx* ctx, int enabled);

#ifdef __cplusplus
}
#endif
#endif /* VRT_H */
