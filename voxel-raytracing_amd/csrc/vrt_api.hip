// vrt_api.hip -- the C-ABI of libvrt_hip.so (include/vrt.h): contexts, scenes, the geometry and
// denoiser stages, strip packing.  Host code only; kernels live in vrt_device.hip.
//
// Call surface mirrored from the reference (paths relative to its root):
//   VoxelScene ctor            source/voxels/resource/voxel_scene.cpp:33-133
//   GeometryStage::record      source/voxels/stages/geometry_stage.cpp:106-153
//   DenoiserStage::record      source/voxels/stages/denoiser_stage.cpp:143-154,156-258
//   Engine::upload_submit      source/engine/engine.cpp:349-375 (blocking uploads)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "image_io.h"
#include "vox_reader.h"
#include "vrt_internal.h"
#include "vrt_denoise_bound.h"

using namespace vrt;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(VRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

} // namespace

// Development switches of a context: A/B experiments and the tests' "the same frame without X" runs.  Each changes speed only,
// never a result; the defaults are the initialisers below (include/vrt.h lists them: not every one is on).  Seeded ONCE, at vrt_ctx_create, from the environment (VRT_TILE_TAGS=0 ...); nothing
// on the render path reads the environment.  vrt_ctx_set_option changes them per context.
struct DevOptions {
    int tile_tags = 1;         // k_tile_tags ahead of K1
    int box_rect = 1;          // the frame's box rectangle
    int xcd_regions = 0;       // launches of >= 8 unsharded frames: one screen region per XCD and frame (off: as a three-dimensional
                               // grid the form runs 30 or 33.6 us per bench frame from one process to the next; rows dealt to the XCDs: 30.0)
    int fast_loop = 1;         // AUTO / DF through the hand-written look-up loop
    int no_bounce_kernel = 1;  // the megakernel without its bounce loop when nothing can bounce
    int packed_bounces = 1;    // the megakernel's bounce chain as one word per hit (no stack of hits in scratch)
    int tags_async = 0;        // the tile tags of a launch on a stream of their own while the context's stream is still busy with the launch before
                               // (off: measured SLOWER -- one frame per call, 1 / 2 / 3 contexts in flight: 53.8 / 31.0 / 26.9 us per frame with the
                               // tags on the context's stream, 58.5 / 34.1 / 52.2 us on their own; tools/exp_r4_inflight.py)
    int ao_batch = 1;          // the AO rays of a wave from a pool in LDS that every lane draws on (df_ao_pool_loop, brick_ao_pool)
    int sky_fast = 1;          // sky texel of waves that cannot hit anything by vrt_sky.h
    int thresh_runs = 1;       // primary rays through df_prim_loop (long runs by threshold)
    int hit_table = 1;         // launches without secondary rays take a hit's colour from the table of colorHit() over materials x normals
    int denoise_th16 = 0;      // the tolerance denoiser on 64 x 16 tiles
    int denoise_packed = 1;    // the exact weighted pass two taps at a time in packed fp32
    int denoise_pair = 1;      // verified passes of the canonical taps with an offset of 2 .. 5 through k_denoise_pair (every weight computed once)
    int denoise_p0 = 1;        // verified pass 0 of the canonical taps through k_denoise_p0 (a wave to itself: no LDS, no barrier)
    int denoise_pair_wgs = 0;  // (experiments) workgroups of a k_denoise_pair launch; 0: as many waves as k_denoise_ver's 1024 workgroups
    int denoise_verified = 1;  // weighted passes through k_denoise_ver (vrt_denoise_bound.h); 0: the exact kernels compute every pixel
    int denoise_guard_div8 = 0;// (tests) an eighth of the guard: how much room the bound leaves
    int denoise_count = 0;     // (tests) count the pixels a verified pass evaluates twice (vrt_debug_denoise_redone)
    int open_cells = 1;        // (scene build) open cells / open bricks in the clearance fields
    int df_prefetch = 1;       // (scene build) secondary rays' look-ups prefetch the neighbouring rows
    int df_own = 1;            // (scene build) AO rays spend their own clearance
};
struct OptName { const char* name; const char* env; int DevOptions::*field; };
static const OptName kOptNames[] = {
    {"tile_tags", "VRT_TILE_TAGS", &DevOptions::tile_tags}, {"box_rect", "VRT_BOX_RECT", &DevOptions::box_rect},
    {"xcd_regions", "VRT_XCD_REGIONS", &DevOptions::xcd_regions}, {"fast_loop", "VRT_FAST_LOOP", &DevOptions::fast_loop},
    {"no_bounce_kernel", "VRT_NO_BOUNCE_KERNEL", &DevOptions::no_bounce_kernel}, {"sky_fast", "VRT_SKY_FAST", &DevOptions::sky_fast},
    {"packed_bounces", "VRT_PACKED_BOUNCES", &DevOptions::packed_bounces}, {"ao_batch", "VRT_AO_BATCH", &DevOptions::ao_batch}, {"tags_async", "VRT_TAGS_ASYNC", &DevOptions::tags_async},
    {"hit_table", "VRT_HIT_TABLE", &DevOptions::hit_table},
    {"thresh_runs", "VRT_THRESH_RUNS", &DevOptions::thresh_runs}, {"denoise_th16", "VRT_DENOISE_TH", &DevOptions::denoise_th16}, {"denoise_packed", "VRT_DENOISE_PACKED", &DevOptions::denoise_packed},
    {"denoise_pair", "VRT_DENOISE_PAIR", &DevOptions::denoise_pair}, {"denoise_p0", "VRT_DENOISE_P0", &DevOptions::denoise_p0}, {"denoise_pair_wgs", "VRT_DENOISE_PAIR_WGS", &DevOptions::denoise_pair_wgs},
    {"denoise_verified", "VRT_DENOISE_VERIFIED", &DevOptions::denoise_verified}, {"denoise_guard_div8", "VRT_DENOISE_GUARD_DIV8", &DevOptions::denoise_guard_div8},
    {"denoise_count", "VRT_DENOISE_COUNT", &DevOptions::denoise_count},
    {"open_cells", "VRT_OPEN_CELLS", &DevOptions::open_cells}, {"df_prefetch", "VRT_DF_PREFETCH", &DevOptions::df_prefetch},
    {"df_own", "VRT_DF_OWN", &DevOptions::df_own},
};

struct vrt_ctx {
    int device = 0;
    DevOptions opt;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    bool timing = true;
    hipEvent_t ev_geo0 = nullptr, ev_prim1 = nullptr, ev_geo1 = nullptr, ev_den0 = nullptr, ev_den1 = nullptr;
    bool have_geo = false, have_den = false;
    uint4* records = nullptr;
    uint32_t* hit_list = nullptr;     // [records_px] + 1 counter word at the end
    size_t records_px = 0;
    int div_w = 0, div_h = 0, div_ok = 0;   // screen size last examined by screen_div_exact, and its verdict
    uint64_t checked_ptrs[2] = {0, 0};   // digests of the image pointers last verified to be device memory (geometry, denoiser)
    // frame-slot tables of launches with more than VRT_MAX_BATCH frames: a ring of device tables, each with a pinned host
    // image that is uploaded on a stream of its own (the copy runs while the previous launch is still tracing)
    static constexpr int kTabRing = 4;
    FrameSlot* tab_dev[kTabRing] = {nullptr, nullptr, nullptr, nullptr};
    FrameSlot* tab_host[kTabRing] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t tab_uploaded[kTabRing] = {nullptr, nullptr, nullptr, nullptr};   // on upload_stream: table i is in device memory
    hipEvent_t tab_consumed[kTabRing] = {nullptr, nullptr, nullptr, nullptr};   // on stream: the launch that read table i is done
    bool tab_busy[kTabRing] = {false, false, false, false};
    int tab_next = 0;
    hipStream_t upload_stream = nullptr;
    // tile tags (k_tile_tags): one word per 8x8-pixel block and frame, valid where == tile_gen.  Two buffers taken in turn: the tags
    // of launch N + 1 are made on a stream of their own while launch N still traces (they depend on the camera alone), and must not
    // land in the buffer launch N reads
    uint32_t* tile_tags[2] = {nullptr, nullptr};
    size_t tile_tags_words[2] = {0, 0};
    uint32_t tile_gen = 0;
    int tag_flip = 0;
    hipStream_t tag_stream = nullptr;
    hipEvent_t tag_done[2] = {nullptr, nullptr};      // on tag_stream: the tags in buffer b are complete
    hipEvent_t tag_read[2] = {nullptr, nullptr};      // on stream: the launch that read buffer b is done
    bool tag_read_valid[2] = {false, false};
    // colorHit() over materials x normals for launches without secondary rays (k_hit_colors), and what it was made from
    uint32_t* hit_colors = nullptr;
    uint64_t hit_scene_gen = 0;
    // the verified denoiser pass, diagnostics ("denoise_count"): pixels evaluated twice, one set of counters per pass
    uint32_t* den_counts = nullptr;    // [10 passes][VRT_DENOISE_SEGS]
    int den_last_passes = 0;           // passes of the latest vrt_denoise call that went through k_denoise_ver (bit i = pass i)
    vrt_settings hit_settings{};
};

struct vrt_scene {
    DevScene d{};
    uint8_t* vox = nullptr;
    uint64_t *occ1 = nullptr, *occ2 = nullptr, *occ3 = nullptr;
    uint8_t* df = nullptr;
    // the clearance fields once more without open cells (launch_open_cells): the march the count planes are rendered with,
    // built when a launch first asks for them
    uint8_t* df_counts = nullptr;
    bool metallic_voxels = true;       // some voxel of the scene has a material with metallic > 0 (only then can a ray bounce, frag:283)
    uint32_t* cells = nullptr;         // occupied 4^3 cells (k_tile_tags), x | y << 10 | z << 20
    uint32_t n_cells = 0;
    bool cells_ok = false;
    bool open_cells = false;
    size_t df_bytes = 0;
    size_t df_guard = 0;               // bytes of room in front of field 0 and behind the last byte of each set of fields: trace_df_fast counts
                                       // its offsets from (W+2)(H+2) bytes in front of field 0, and its prefetches reach one slice past the border
    std::mutex lazy;
    vrt_material* palette = nullptr;
    float* sky = nullptr;
    uint8_t* noise = nullptr;
    float* sky_normals = nullptr;
    uint32_t* sky8 = nullptr;
    uint32_t occ2_bytes = 0, occ3_bytes = 0;
    // brick scenes
    uint32_t* bgrid = nullptr; uint8_t* bcoarse = nullptr; uint8_t* bpool = nullptr; uint8_t* bfine = nullptr;
    uint64_t* bentry = nullptr;        // bgrid + bcoarse folded into one word per brick (what the march reads; the two are freed after the build)
    bool bricks = false;
    uint64_t bytes = 0;                // device memory held (volume structures + textures)
    uint64_t shade_gen = 0;            // changes whenever something a hit's colour depends on does (creation, vrt_scene_set_sky)
};
static std::atomic<uint64_t> g_shade_gen{0};

extern "C" {

const char* vrt_last_error(void) { return g_err.c_str(); }

int vrt_ctx_create(int device, vrt_ctx** out)
{
    if (!out) return fail(VRT_ERR_INVALID, "vrt_ctx_create: out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(VRT_ERR_NO_DEVICE, std::string("no HIP device available (") + hipGetErrorString(e) +
                                           "); this library has no CPU fallback");
    if (device < 0 || device >= n) return fail(VRT_ERR_INVALID, "vrt_ctx_create: device index out of range");
    HIPCHK(hipSetDevice(device));
    vrt_ctx* c = new vrt_ctx();
    c->device = device;
    for (const OptName& o : kOptNames) {                       // the one place the environment is read
        const char* e = getenv(o.env);
        if (e && e[0] >= '0' && e[0] <= '9') c->opt.*(o.field) = atoi(e);        // (switches: 0 / 1; a few take small numbers)
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(VRT_ERR_HIP, "hipStreamCreate failed"); }
    hipEventCreate(&c->ev_geo0); hipEventCreate(&c->ev_prim1); hipEventCreate(&c->ev_geo1);
    hipEventCreate(&c->ev_den0); hipEventCreate(&c->ev_den1);
    *out = c;
    return VRT_OK;
}

void vrt_ctx_destroy(vrt_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->records) hipFree(c->records);
    if (c->hit_list) hipFree(c->hit_list);
    if (c->tag_stream) { hipStreamSynchronize(c->tag_stream); hipStreamDestroy(c->tag_stream); }
    for (int b = 0; b < 2; b++) {
        if (c->tile_tags[b]) hipFree(c->tile_tags[b]);
        if (c->tag_done[b]) hipEventDestroy(c->tag_done[b]);
        if (c->tag_read[b]) hipEventDestroy(c->tag_read[b]);
    }
    if (c->hit_colors) hipFree(c->hit_colors);
    if (c->den_counts) hipFree(c->den_counts);
    if (c->upload_stream) { hipStreamSynchronize(c->upload_stream); hipStreamDestroy(c->upload_stream); }
    for (int i = 0; i < vrt_ctx::kTabRing; i++) {
        if (c->tab_dev[i]) hipFree(c->tab_dev[i]);
        if (c->tab_host[i]) hipHostFree(c->tab_host[i]);
        if (c->tab_uploaded[i]) hipEventDestroy(c->tab_uploaded[i]);
        if (c->tab_consumed[i]) hipEventDestroy(c->tab_consumed[i]);
    }
    hipEventDestroy(c->ev_geo0); hipEventDestroy(c->ev_prim1); hipEventDestroy(c->ev_geo1);
    hipEventDestroy(c->ev_den0); hipEventDestroy(c->ev_den1);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int vrt_ctx_set_stream(vrt_ctx* c, void* hip_stream)
{
    if (!c) return fail(VRT_ERR_INVALID, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;      // NULL is a valid handle: the HIP null (legacy default) stream
    c->own_stream = false;
    c->have_geo = c->have_den = false;
    c->checked_ptrs[0] = c->checked_ptrs[1] = 0;
    return VRT_OK;
}

int vrt_ctx_synchronize(vrt_ctx* c)
{
    if (!c) return fail(VRT_ERR_INVALID, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_ctx_set_option(vrt_ctx* c, const char* name, int32_t value)
{
    if (!c || !name) return fail(VRT_ERR_INVALID, "vrt_ctx_set_option: NULL argument");
    for (const OptName& o : kOptNames)
        if (strcmp(o.name, name) == 0) { c->opt.*(o.field) = value < 0 ? 0 : value; return VRT_OK; }
    return fail(VRT_ERR_INVALID, std::string("vrt_ctx_set_option: unknown option '") + name + "'");
}

int vrt_ctx_get_option(vrt_ctx* c, const char* name, int32_t* value)
{
    if (!c || !name || !value) return fail(VRT_ERR_INVALID, "vrt_ctx_get_option: NULL argument");
    for (const OptName& o : kOptNames)
        if (strcmp(o.name, name) == 0) { *value = c->opt.*(o.field); return VRT_OK; }
    return fail(VRT_ERR_INVALID, std::string("vrt_ctx_get_option: unknown option '") + name + "'");
}

int vrt_ctx_set_timing(vrt_ctx* c, int enabled)
{
    if (!c) return fail(VRT_ERR_INVALID, "ctx is NULL");
    c->timing = enabled != 0;
    return VRT_OK;
}

int vrt_device_info(vrt_ctx* c, char* name, size_t name_len, int* compute_units)
{
    if (!c) return fail(VRT_ERR_INVALID, "ctx is NULL");
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, c->device));
    if (name && name_len) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    if (compute_units) *compute_units = p.multiProcessorCount;
    return VRT_OK;
}

int vrt_device_alloc(vrt_ctx* c, size_t bytes, void** out)
{
    if (!c || !out) return fail(VRT_ERR_INVALID, "vrt_device_alloc: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMalloc(out, bytes ? bytes : 1));
    return VRT_OK;
}

int vrt_device_free(vrt_ctx* c, void* p)
{
    if (!c) return fail(VRT_ERR_INVALID, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (p) HIPCHK(hipFree(p));
    c->checked_ptrs[0] = c->checked_ptrs[1] = 0;     // a freed address may come back as something else: verify again
    return VRT_OK;
}

int vrt_memcpy_h2d(vrt_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return fail(VRT_ERR_INVALID, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_memcpy_d2h(vrt_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return fail(VRT_ERR_INVALID, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_memset(vrt_ctx* c, void* dst, int value, size_t bytes)
{
    if (!c) return fail(VRT_ERR_INVALID, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemsetAsync(dst, value, bytes, c->stream));
    return VRT_OK;
}

// ---- scene ---------------------------------------------------------------------------------------

void vrt_scene_free(vrt_ctx* c, vrt_scene* s)
{
    if (!s) return;
    if (c) { hipSetDevice(c->device); hipStreamSynchronize(c->stream); }
    if (s->vox) hipFree(s->vox);
    if (s->occ1) hipFree(s->occ1);
    if (s->occ2) hipFree(s->occ2);
    if (s->occ3) hipFree(s->occ3);
    if (s->df) hipFree(s->df - s->df_guard);
    if (s->df_counts) hipFree(s->df_counts - s->df_guard);
    if (s->cells) hipFree(s->cells);
    if (s->palette) hipFree(s->palette);
    if (s->sky) hipFree(s->sky);
    if (s->noise) hipFree(s->noise);
    if (s->sky_normals) hipFree(s->sky_normals);
    if (s->sky8) hipFree(s->sky8);
    if (s->bgrid) hipFree(s->bgrid);
    if (s->bcoarse) hipFree(s->bcoarse);
    if (s->bpool) hipFree(s->bpool);
    if (s->bfine) hipFree(s->bfine);
    if (s->bentry) hipFree(s->bentry);
    delete s;
}

int vrt_scene_set_sky(vrt_ctx* c, vrt_scene* s, const float* rgba, uint32_t w, uint32_t h)
{
    if (!c || !s || !rgba || !w || !h) return fail(VRT_ERR_INVALID, "vrt_scene_set_sky: bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    float* d = nullptr;
    size_t bytes = (size_t)w * h * 16;
    HIPCHK(hipMalloc((void**)&d, bytes));
    HIPCHK(hipMemcpy(d, rgba, bytes, hipMemcpyHostToDevice));
    uint32_t* d8 = nullptr;
    { hipError_t e8 = hipMalloc((void**)&d8, (size_t)w * h * 4); if (e8 != hipSuccess) { hipFree(d); return fail(VRT_ERR_HIP, std::string("hipMalloc (sky RGBA8): ") + hipGetErrorString(e8)); } }
    if (s->sky) hipFree(s->sky);
    if (s->sky8) hipFree(s->sky8);
    s->sky = d; s->d.sky = d; s->d.sky_w = w; s->d.sky_h = h;
    // the sky as the colour target stores a miss, and the constants of the texel fast path (vrt_sky.h)
    s->sky8 = d8; s->d.sky8 = d8; s->d.skyk = sky_fast_consts(w, h);
    s->shade_gen = ++g_shade_gen;
    HIPCHK(launch_sky_rgba8(d, d8, (size_t)w * h, c->stream));
    // skyColor of the normals a hit can have (calcAmbient's sky tint, frag:224): 64 x float4, by the shading code itself
    if (!s->sky_normals) HIPCHK(hipMalloc((void**)&s->sky_normals, 64 * 4 * sizeof(float)));
    s->d.sky_normals = s->sky_normals;
    HIPCHK(launch_sky_normals(s->d, s->sky_normals, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_scene_set_blue_noise(vrt_ctx* c, vrt_scene* s, const uint8_t* rgba8, uint32_t w, uint32_t h)
{
    if (!c || !s || !rgba8 || !w || !h) return fail(VRT_ERR_INVALID, "vrt_scene_set_blue_noise: bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint8_t* d = nullptr;
    size_t bytes = (size_t)w * h * 4;
    HIPCHK(hipMalloc((void**)&d, bytes));
    HIPCHK(hipMemcpy(d, rgba8, bytes, hipMemcpyHostToDevice));
    if (s->noise) hipFree(s->noise);
    s->noise = d; s->d.noise = d; s->d.noise_w = w; s->d.noise_h = h;
    return VRT_OK;
}

} // extern "C"

// does any of the n voxel ids have a metallic material?  (decides whether the megakernel needs its bounce stack)
static bool any_metallic(const uint8_t* ids, size_t n, const vrt_material palette[256])
{
    bool metal[256], any = false;
    for (int i = 0; i < 256; i++) { metal[i] = palette[i].metallic > 0.0f; any = any || (i != 0 && metal[i]); }
    if (!any) return false;
    for (size_t i = 0; i < n; i++)
        if (ids[i] != 0 && metal[ids[i]]) return true;
    return false;
}

// The clearance fields of a dense scene into dst (df_bytes: eight fields, or nine and the 0xFF byte in trace_df_fast's layout),
// from the voxels already on the device; open: with the open cells coded 0 (launch_open_cells).  Returns when they are built.
static hipError_t build_fields(vrt_ctx* c, const vrt_scene* s, uint8_t* dst, bool open)
{
    const VolumeView& d = s->d.vol;
    const size_t nvox = (size_t)d.W * (size_t)d.H * (size_t)d.D, ndf = df_field_bytes(d.W, d.H, d.D);
    hipError_t e = hipMemsetAsync(dst, 0, s->df_bytes, c->stream);
    if (e != hipSuccess) return e;
    if (d.df_fast) {
        if ((e = hipMemsetAsync(dst + 9 * ndf, 0xFF, 1, c->stream)) != hipSuccess) return e;
        if ((e = launch_pad_vox(s->vox, d.W, d.H, d.D, dst + 8 * ndf, c->stream)) != hipSuccess) return e;
    }
    uint8_t *tmp0 = nullptr, *tmp1 = nullptr;                   // ping-pong buffers of the 3-pass transforms
    if ((e = hipMalloc((void**)&tmp0, nvox)) != hipSuccess) return e;
    e = hipMalloc((void**)&tmp1, nvox);
    if (e == hipSuccess) e = launch_build_df(s->vox, d.W, d.H, d.D, dst, ndf, tmp0, tmp1, c->stream);
    if (e == hipSuccess && open) e = launch_open_cells(s->vox, d.W, d.H, d.D, dst, ndf, tmp0, tmp1, c->stream);
    const hipError_t sync = hipStreamSynchronize(c->stream);
    hipFree(tmp0); if (tmp1) hipFree(tmp1);
    return e != hipSuccess ? e : sync;
}

// The fields a launch that writes count planes marches through (no open cells: the iterations of the reference's loop, to the
// wall), built on first use.
static int fields_for_counts(vrt_ctx* c, const vrt_scene* cs, const uint8_t** out)
{
    vrt_scene* s = const_cast<vrt_scene*>(cs);
    std::lock_guard<std::mutex> lock(s->lazy);
    if (!s->open_cells) { *out = s->df; return VRT_OK; }
    if (!s->df_counts) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (uint64_t)s->df_bytes + 2ull * (uint64_t)s->d.vol.W * s->d.vol.H * s->d.vol.D > (uint64_t)free_b)
            return fail(VRT_ERR_UNSUPPORTED, "vrt_render_geometry: the count planes (steps_primary, steps_total) need a second set of clearance fields (" +
                        std::to_string((uint64_t)s->df_bytes) + " bytes), which does not fit the device memory that is free");
        uint8_t* p = nullptr;
        HIPCHK(hipMalloc((void**)&p, s->df_bytes + 2 * s->df_guard));
        hipError_t e = hipMemsetAsync(p, 0, s->df_bytes + 2 * s->df_guard, c->stream);
        if (e == hipSuccess) e = build_fields(c, s, p + s->df_guard, false);
        if (e != hipSuccess) { hipFree(p); return fail(VRT_ERR_HIP, std::string("building the count planes' clearance fields: ") + hipGetErrorString(e)); }
        s->df_counts = p + s->df_guard;
        s->bytes += s->df_bytes;
    }
    *out = s->df_counts;
    return VRT_OK;
}

extern "C" {

int vrt_scene_from_dense(vrt_ctx* c, const uint8_t* voxels, uint32_t W, uint32_t H, uint32_t D,
                         const vrt_material palette[256], vrt_scene** out)
{
    if (!c || !voxels || !palette || !out) return fail(VRT_ERR_INVALID, "vrt_scene_from_dense: NULL argument");
    if (!W || !H || !D || W > 4096 || H > 4096 || D > 4096)
        return fail(VRT_ERR_UNSUPPORTED, "vrt_scene_from_dense: each dimension must be in 1..4096");
    HIPCHK(hipSetDevice(c->device));
    vrt_scene* s = new vrt_scene();
    s->shade_gen = ++g_shade_gen;
    VolumeView& d = s->d.vol;
    d.W = (int)W; d.H = (int)H; d.D = (int)D;
    d.n1x = ceil_div(d.W, 4); d.n1y = ceil_div(d.H, 4); d.n1z = ceil_div(d.D, 4);
    d.n2x = ceil_div(d.n1x, 4); d.n2y = ceil_div(d.n1y, 4); d.n2z = ceil_div(d.n1z, 4);
    d.n3x = ceil_div(d.n2x, 4); d.n3y = ceil_div(d.n2y, 4); d.n3z = ceil_div(d.n2z, 4);
    size_t nvox = (size_t)W * H * D;
    size_t n1 = (size_t)d.n1x * d.n1y * d.n1z, n2 = (size_t)d.n2x * d.n2y * d.n2z, n3 = (size_t)d.n3x * d.n3y * d.n3z;
    size_t n2pad = (n2 + 1) & ~(size_t)1;          // 16-byte multiples for the uint4 LDS staging loop
    size_t n3pad = (n3 + 1) & ~(size_t)1;
    size_t ndf = df_field_bytes(d.W, d.H, d.D);    // one clearance field: x-fastest with a one-voxel border of zeros
    int rc = VRT_OK;
    {
        // the dense scene holds about 10x the voxel bytes (eight or nine clearance fields) and two more volumes while it is
        // built: say so up front instead of failing half way through the allocations
        const uint64_t need = (uint64_t)nvox * 3u + 9ull * ndf + (n1 + n2pad + n3pad) * 8ull + (64ull << 20);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > (uint64_t)free_b) {
            delete s;
            return fail(VRT_ERR_UNSUPPORTED, "vrt_scene_from_dense: a " + std::to_string(W) + "x" + std::to_string(H) + "x" + std::to_string(D) +
                        " dense scene needs " + std::to_string(need) + " bytes of device memory (" + std::to_string((uint64_t)free_b) +
                        " free); hand the volume over in bricks (vrt_scene_from_bricks)");
        }
    }
#define SCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(VRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); goto bad; } } while (0)
    s->metallic_voxels = any_metallic(voxels, nvox, palette);
    SCHK(hipMalloc((void**)&s->vox, nvox));
    SCHK(hipMalloc((void**)&s->occ1, n1 * 8));
    SCHK(hipMalloc((void**)&s->occ2, n2pad * 8));
    SCHK(hipMalloc((void**)&s->occ3, n3pad * 8));
    SCHK(hipMalloc((void**)&s->palette, 256 * sizeof(vrt_material)));
    // eight clearance fields, and -- while 32-bit offsets reach all of it -- a ninth field with the voxel ids in the same
    // layout plus one byte 0xFF behind it (trace_df_fast)
    {
        const bool fast = df_fast_layout_ok(d.W, d.H, d.D);
        const size_t bytes = fast ? 9 * ndf + 256 : 8 * ndf;
        s->df_guard = ((((size_t)W + 2u) * ((size_t)H + 2u)) * 2u + 511u) & ~(size_t)255u;
        {
            uint8_t* raw = nullptr;
            SCHK(hipMalloc((void**)&raw, bytes + 2 * s->df_guard));
            s->df = raw + s->df_guard;
            SCHK(hipMemsetAsync(raw, 0, bytes + 2 * s->df_guard, c->stream));
        }
        s->df_bytes = bytes;
        s->bytes = (uint64_t)nvox + bytes + (n1 + n2pad + n3pad) * 8ull + 256 * sizeof(vrt_material);
        d.df_fast = fast ? 1u : 0u;
    }
    SCHK(hipMemsetAsync(s->occ2, 0, n2pad * 8, c->stream));
    SCHK(hipMemsetAsync(s->occ3, 0, n3pad * 8, c->stream));
    SCHK(hipMemcpyAsync(s->vox, voxels, nvox, hipMemcpyHostToDevice, c->stream));
    SCHK(hipMemcpyAsync(s->palette, palette, 256 * sizeof(vrt_material), hipMemcpyHostToDevice, c->stream));
    SCHK(launch_build_pyramid(s->vox, d.W, d.H, d.D, s->occ1, s->occ2, s->occ3, c->stream));
    // the occupied 4^3 cells as a list (from the 16^3 summaries: one bit per cell), for the tile tags of a launch
    if (d.n1x <= 1024 && d.n1y <= 1024 && d.n1z <= 1024) {
        std::vector<uint64_t> h2(n2);
        SCHK(hipMemcpyAsync(h2.data(), s->occ2, n2 * 8, hipMemcpyDeviceToHost, c->stream));
        SCHK(hipStreamSynchronize(c->stream));
        std::vector<uint32_t> cells;
        for (size_t w = 0; w < n2; w++) {
            uint64_t bits = h2[w];
            if (!bits) continue;
            const uint32_t wx = (uint32_t)(w % (size_t)d.n2x), wy = (uint32_t)((w / (size_t)d.n2x) % (size_t)d.n2y), wz = (uint32_t)(w / ((size_t)d.n2x * d.n2y));
            for (uint32_t b = 0; b < 64; b++)
                if ((bits >> b) & 1ull) cells.push_back((wx * 4u + (b & 3u)) | ((wy * 4u + ((b >> 2) & 3u)) << 10) | ((wz * 4u + (b >> 4)) << 20));
        }
        if (cells.size() <= (4u << 20)) {                                   // (beyond that the tags cost more than they save)
            if (!cells.empty()) {
                SCHK(hipMalloc((void**)&s->cells, cells.size() * 4));
                SCHK(hipMemcpy(s->cells, cells.data(), cells.size() * 4, hipMemcpyHostToDevice));
                s->bytes += cells.size() * 4;
            }
            s->n_cells = (uint32_t)cells.size();
            s->cells_ok = true;
        }
    }
    {
        s->open_cells = c->opt.open_cells != 0;                           // development switch: 0 = fields without open cells
        SCHK(build_fields(c, s, s->df, s->open_cells));
    }
#undef SCHK
    d.vox = s->vox; d.occ1 = s->occ1; d.occ2 = s->occ2; d.occ3 = s->occ3; d.df = s->df; d.df_stride = ndf; s->d.palette = s->palette;
    {
        d.df_prefetch = (d.df_fast && c->opt.df_prefetch) ? 1u : 0u;       // development switch: 0 = no neighbour-row prefetch in the secondary rays' look-ups
        d.df_own = (d.df_fast && c->opt.df_own) ? 1u : 0u;                 // development switch: 0 = the AO rays through the wave-minimum loop too
    }
    s->occ2_bytes = (uint32_t)(n2pad * 8); s->occ3_bytes = (uint32_t)(n3pad * 8);
    {
        const float white[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        const uint8_t grey[4] = {128, 128, 128, 255};
        rc = vrt_scene_set_sky(c, s, white, 1, 1);
        if (rc == VRT_OK) rc = vrt_scene_set_blue_noise(c, s, grey, 1, 1);
        if (rc != VRT_OK) goto bad;
    }
    *out = s;
    return VRT_OK;
bad:
    vrt_scene_free(c, s);
    return rc;
}

int vrt_scene_from_bricks(vrt_ctx* c, const uint32_t* grid, uint32_t nbx, uint32_t nby, uint32_t nbz,
                          const uint8_t* pool, uint32_t n_bricks, const vrt_material palette[256], vrt_scene** out)
{
    if (!c || !grid || !palette || !out || (n_bricks && !pool)) return fail(VRT_ERR_INVALID, "vrt_scene_from_bricks: NULL argument");
    if (!nbx || !nby || !nbz || nbx > 512 || nby > 512 || nbz > 512)
        return fail(VRT_ERR_UNSUPPORTED, "vrt_scene_from_bricks: each dimension must be 1..512 bricks (8..4096 voxels)");
    if (n_bricks >= 0xFFFFFEu) return fail(VRT_ERR_UNSUPPORTED, "vrt_scene_from_bricks: at most 2^24 - 2 occupied bricks (the march's 24-bit brick pointer)");
    const size_t nb = (size_t)nbx * nby * nbz;
    // the brick a pool entry belongs to, as an index into the padded grid; every entry must be referenced exactly once
    std::vector<uint32_t> coord(n_bricks, 0xFFFFFFFFu);
    const size_t pbx = (size_t)nbx + 2, pby = (size_t)nby + 2, pbz = (size_t)nbz + 2, npad = pbx * pby * pbz;
    for (size_t i = 0; i < nb; i++) {
        const uint32_t g = grid[i];
        if (g == 0u) continue;
        if (g > n_bricks) return fail(VRT_ERR_INVALID, "vrt_scene_from_bricks: a grid entry points past the pool");
        if (coord[g - 1u] != 0xFFFFFFFFu) return fail(VRT_ERR_INVALID, "vrt_scene_from_bricks: two grid entries share a pool brick");
        const size_t x = i % nbx, y = (i / nbx) % nby, z = i / ((size_t)nbx * nby);
        coord[g - 1u] = (uint32_t)((x + 1) + ((y + 1) + (z + 1) * pby) * pbx);
    }
    for (uint32_t i = 0; i < n_bricks; i++)
        if (coord[i] == 0xFFFFFFFFu) return fail(VRT_ERR_INVALID, "vrt_scene_from_bricks: a pool brick is not referenced by the grid");
    HIPCHK(hipSetDevice(c->device));
    vrt_scene* s = new vrt_scene();
    s->shade_gen = ++g_shade_gen;
    s->bricks = true;
    VolumeView& d = s->d.vol;
    d.W = (int)(nbx * 8u); d.H = (int)(nby * 8u); d.D = (int)(nbz * 8u);
    d.pbx = (int)pbx; d.pby = (int)pby;
    const size_t cstride = df_field_bytes((int)nbx, (int)nby, (int)nbz);          // one padded coarse field (= npad rounded up to 256 B)
    const size_t pool_bytes = (size_t)n_bricks * 512u, fine_bytes = pool_bytes * 8u;
    uint32_t *grid_dev = nullptr, *coord_dev = nullptr;
    uint8_t *occ = nullptr, *tmp0 = nullptr, *tmp1 = nullptr;
    int rc = VRT_OK;
#define SCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(VRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); goto bad; } } while (0)
    s->metallic_voxels = any_metallic(pool, pool_bytes, palette);
    SCHK(hipMalloc((void**)&s->bgrid, npad * 4));
    SCHK(hipMalloc((void**)&s->bcoarse, 8 * cstride));
    SCHK(hipMalloc((void**)&s->bpool, pool_bytes ? pool_bytes : 1));
    SCHK(hipMalloc((void**)&s->bfine, fine_bytes ? fine_bytes : 1));
    SCHK(hipMalloc((void**)&s->palette, 256 * sizeof(vrt_material)));
    SCHK(hipMalloc((void**)&grid_dev, nb * 4));
    SCHK(hipMalloc((void**)&coord_dev, (size_t)(n_bricks ? n_bricks : 1) * 4));
    SCHK(hipMalloc((void**)&occ, nb));
    SCHK(hipMalloc((void**)&tmp0, nb));
    SCHK(hipMalloc((void**)&tmp1, nb));
    s->bytes = npad * 4ull + 8ull * cstride + pool_bytes + fine_bytes + 256 * sizeof(vrt_material);
    SCHK(hipMemsetAsync(s->bgrid, 0xFF, npad * 4, c->stream));                  // border: 0xFFFFFFFF = outside the volume
    SCHK(hipMemsetAsync(s->bcoarse, 0, 8 * cstride, c->stream));
    SCHK(hipMemcpyAsync(grid_dev, grid, nb * 4, hipMemcpyHostToDevice, c->stream));
    if (n_bricks) {
        SCHK(hipMemcpyAsync(coord_dev, coord.data(), (size_t)n_bricks * 4, hipMemcpyHostToDevice, c->stream));
        SCHK(hipMemcpyAsync(s->bpool, pool, pool_bytes, hipMemcpyHostToDevice, c->stream));
    }
    SCHK(hipMemcpyAsync(s->palette, palette, 256 * sizeof(vrt_material), hipMemcpyHostToDevice, c->stream));
    SCHK(launch_brick_grid(grid_dev, (int)nbx, (int)nby, (int)nbz, s->bgrid, occ, c->stream));
    // brick-level clearance: the dense scene's transform over the occupancy of the bricks, capped at 16 bricks
    SCHK(launch_build_df(occ, (int)nbx, (int)nby, (int)nbz, s->bcoarse, cstride, tmp0, tmp1, c->stream, 16));
    // open bricks (bit 7): no occupied brick left between here and the volume's corner in the octant's direction
    {
        s->open_cells = c->opt.open_cells != 0;                           // development switch: 0 = no open bricks
        if (s->open_cells) SCHK(launch_open_cells(occ, (int)nbx, (int)nby, (int)nbz, s->bcoarse, cstride, tmp0, tmp1, c->stream, 0x80));
        d.brick_open = s->open_cells ? 1u : 0u;
        d.df_own = c->opt.df_own ? 1u : 0u;                                // development switch: 0 = the AO rays through the wave-minimum loop too
    }
    // the occupied bricks as a list of 8^3 cells, for the tile tags of a launch
    if (n_bricks <= (4u << 20)) {
        std::vector<uint32_t> cells(n_bricks);
        for (uint32_t i = 0; i < n_bricks; i++) {
            const uint32_t pc = coord[i];
            cells[i] = (uint32_t)(pc % pbx - 1) | ((uint32_t)((pc / pbx) % pby - 1) << 10) | ((uint32_t)(pc / (pbx * pby) - 1) << 20);
        }
        if (n_bricks) {
            SCHK(hipMalloc((void**)&s->cells, (size_t)n_bricks * 4));
            SCHK(hipMemcpy(s->cells, cells.data(), (size_t)n_bricks * 4, hipMemcpyHostToDevice));
            s->bytes += (uint64_t)n_bricks * 4;
        }
        s->n_cells = n_bricks; s->cells_ok = true;
    }
    SCHK(launch_brick_fine(s->bgrid, (int)pbx, (int)pby, coord_dev, n_bricks, s->bpool, s->bfine, c->stream));
    // what the march reads: pointer, open bits and the eight coarse clearances of a brick in ONE 8-byte word; the pointer grid and
    // the coarse fields were only needed to build it (and the fine bytes)
    SCHK(hipMalloc((void**)&s->bentry, npad * 8));
    SCHK(launch_brick_pack(s->bgrid, s->bcoarse, cstride, npad, s->bentry, c->stream));
    SCHK(hipStreamSynchronize(c->stream));
    hipFree(s->bgrid); hipFree(s->bcoarse); s->bgrid = nullptr; s->bcoarse = nullptr;
    s->bytes = npad * 8ull + pool_bytes + fine_bytes + 256 * sizeof(vrt_material) + (uint64_t)n_bricks * 4;
#undef SCHK
    hipFree(grid_dev); hipFree(coord_dev); hipFree(occ); hipFree(tmp0); hipFree(tmp1);
    grid_dev = coord_dev = nullptr; occ = tmp0 = tmp1 = nullptr;
    d.bgrid = nullptr; d.bcoarse = nullptr; d.bcoarse_stride = cstride; d.bpool = s->bpool; d.bfine = s->bfine; d.bentry = s->bentry;
    s->d.palette = s->palette;
    {
        const float white[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        const uint8_t grey[4] = {128, 128, 128, 255};
        rc = vrt_scene_set_sky(c, s, white, 1, 1);
        if (rc == VRT_OK) rc = vrt_scene_set_blue_noise(c, s, grey, 1, 1);
        if (rc != VRT_OK) goto bad;
    }
    *out = s;
    return VRT_OK;
bad:
    if (grid_dev) hipFree(grid_dev);
    if (coord_dev) hipFree(coord_dev);
    if (occ) hipFree(occ);
    if (tmp0) hipFree(tmp0);
    if (tmp1) hipFree(tmp1);
    vrt_scene_free(c, s);
    return rc;
}

int vrt_scene_trim(vrt_ctx* c, vrt_scene* s)
{
    if (!c || !s) return fail(VRT_ERR_INVALID, "vrt_scene_trim: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    // a scene may be shared by several contexts (frames in flight: one stream each); a count-plane launch enqueued on ANY of them
    // may still be reading the fields freed below, so this waits for the whole device, not for the calling context's stream
    HIPCHK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(s->lazy);
    if (s->df_counts) {
        hipFree(s->df_counts - s->df_guard);
        s->df_counts = nullptr;
        s->bytes -= s->df_bytes;
    }
    return VRT_OK;
}

int vrt_scene_memory(const vrt_scene* s, uint64_t* bytes)
{
    if (!s || !bytes) return fail(VRT_ERR_INVALID, "vrt_scene_memory: NULL argument");
    *bytes = s->bytes + (uint64_t)s->d.sky_w * s->d.sky_h * 20u + (uint64_t)s->d.noise_w * s->d.noise_h * 4u + 64u * 16u;
    return VRT_OK;
}

int vrt_vox_flatten_host(const void* buf, size_t n, uint32_t dims[3], uint8_t** voxels,
                         vrt_material palette[256], uint32_t* num_instances, uint64_t* dropped)
{
    if (!buf || !dims || !voxels || !palette) return fail(VRT_ERR_INVALID, "vrt_vox_flatten_host: NULL argument");
    FlatScene fs; std::string err;
    int rc = vox_flatten((const uint8_t*)buf, n, fs, err);
    if (rc != VRT_OK) return fail(rc, err);
    dims[0] = fs.dims[0]; dims[1] = fs.dims[1]; dims[2] = fs.dims[2];
    uint8_t* v = (uint8_t*)malloc(fs.voxels.size() ? fs.voxels.size() : 1);
    if (!v) return fail(VRT_ERR_INVALID, "out of host memory");
    memcpy(v, fs.voxels.data(), fs.voxels.size());
    *voxels = v;
    memcpy(palette, fs.palette, sizeof(fs.palette));
    if (num_instances) *num_instances = fs.num_instances;
    if (dropped) *dropped = fs.dropped;
    return VRT_OK;
}

void vrt_host_free(void* p) { free(p); }

int vrt_scene_load_vox_mem(vrt_ctx* c, const void* buf, size_t n, vrt_scene** out)
{
    if (!c || !buf || !out) return fail(VRT_ERR_INVALID, "vrt_scene_load_vox_mem: NULL argument");
    FlatScene fs; std::string err;
    int rc = vox_flatten((const uint8_t*)buf, n, fs, err);
    if (rc != VRT_OK) return fail(rc, err);
    return vrt_scene_from_dense(c, fs.voxels.data(), fs.dims[0], fs.dims[1], fs.dims[2], fs.palette, out);
}

int vrt_scene_load_vox_file(vrt_ctx* c, const char* path, vrt_scene** out)
{
    if (!c || !path || !out) return fail(VRT_ERR_INVALID, "vrt_scene_load_vox_file: NULL argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(VRT_ERR_IO, "Failed to read voxel scene");
    std::vector<uint8_t> buf;
    if (fseek(f, 0, SEEK_END) == 0) {
        long sz = ftell(f);
        if (sz > 0) { buf.resize((size_t)sz); rewind(f); if (fread(buf.data(), 1, buf.size(), f) != buf.size()) buf.clear(); }
    }
    fclose(f);
    if (buf.empty()) return fail(VRT_ERR_IO, "Failed to read voxel scene");
    return vrt_scene_load_vox_mem(c, buf.data(), buf.size(), out);
}

int vrt_image_load(const char* path, int* is_hdr, uint32_t* w, uint32_t* h, void** pixels)
{
    if (!path || !w || !h || !pixels) return fail(VRT_ERR_INVALID, "vrt_image_load: NULL argument");
    LoadedImage img; std::string err;
    int rc = image_load(path, img, err);
    if (rc != VRT_OK) return fail(rc, err);
    size_t bytes = img.is_hdr ? img.f32.size() * sizeof(float) : img.u8.size();
    void* p = malloc(bytes ? bytes : 1);
    if (!p) return fail(VRT_ERR_INVALID, "out of host memory");
    memcpy(p, img.is_hdr ? (const void*)img.f32.data() : (const void*)img.u8.data(), bytes);
    *pixels = p; *w = img.w; *h = img.h;
    if (is_hdr) *is_hdr = img.is_hdr ? 1 : 0;
    return VRT_OK;
}

int vrt_scene_set_sky_file(vrt_ctx* c, vrt_scene* s, const char* path)
{
    if (!c || !s || !path) return fail(VRT_ERR_INVALID, "vrt_scene_set_sky_file: NULL argument");
    LoadedImage img; std::string err;
    int rc = image_load(path, img, err);
    if (rc != VRT_OK) return fail(rc, err);
    if (!img.is_hdr) {                                        // 8-bit image as sky: c/255 per channel
        img.f32.resize(img.u8.size());
        for (size_t i = 0; i < img.u8.size(); i++) img.f32[i] = (float)img.u8[i] / 255.0f;
    }
    return vrt_scene_set_sky(c, s, img.f32.data(), img.w, img.h);
}

int vrt_scene_set_blue_noise_file(vrt_ctx* c, vrt_scene* s, const char* path)
{
    if (!c || !s || !path) return fail(VRT_ERR_INVALID, "vrt_scene_set_blue_noise_file: NULL argument");
    LoadedImage img; std::string err;
    int rc = image_load(path, img, err);
    if (rc != VRT_OK) return fail(rc, err);
    if (img.is_hdr) return fail(VRT_ERR_UNSUPPORTED, "vrt_scene_set_blue_noise_file: the noise texture is RGBA8_UNORM, got a float image");
    return vrt_scene_set_blue_noise(c, s, img.u8.data(), img.w, img.h);
}

int vrt_image_write_png(const char* path, const uint8_t* rgba8, uint32_t w, uint32_t h)
{
    if (!path || !rgba8 || !w || !h) return fail(VRT_ERR_INVALID, "vrt_image_write_png: bad argument");
    std::string err; int rc = image_write_png(path, rgba8, w, h, err);
    return rc == VRT_OK ? rc : fail(rc, err);
}

int vrt_image_write_ppm(const char* path, const uint8_t* rgba8, uint32_t w, uint32_t h)
{
    if (!path || !rgba8 || !w || !h) return fail(VRT_ERR_INVALID, "vrt_image_write_ppm: bad argument");
    std::string err; int rc = image_write_ppm(path, rgba8, w, h, err);
    return rc == VRT_OK ? rc : fail(rc, err);
}

int vrt_image_write_pfm(const char* path, const float* pixels, uint32_t w, uint32_t h, uint32_t stride_floats)
{
    if (!path || !pixels || !w || !h || stride_floats < 3) return fail(VRT_ERR_INVALID, "vrt_image_write_pfm: bad argument");
    std::string err; int rc = image_write_pfm(path, pixels, w, h, stride_floats, err);
    return rc == VRT_OK ? rc : fail(rc, err);
}

int vrt_scene_info(const vrt_scene* s, uint32_t dims[3])
{
    if (!s || !dims) return fail(VRT_ERR_INVALID, "vrt_scene_info: NULL argument");
    dims[0] = (uint32_t)s->d.vol.W; dims[1] = (uint32_t)s->d.vol.H; dims[2] = (uint32_t)s->d.vol.D;
    return VRT_OK;
}

int vrt_scene_download(vrt_ctx* c, const vrt_scene* s, uint8_t* voxels, vrt_material palette[256])
{
    if (!c || !s) return fail(VRT_ERR_INVALID, "vrt_scene_download: NULL argument");
    if (s->bricks && voxels) return fail(VRT_ERR_UNSUPPORTED, "vrt_scene_download: a brick scene has no dense volume to copy back");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (voxels) HIPCHK(hipMemcpy(voxels, s->vox, (size_t)s->d.vol.W * s->d.vol.H * s->d.vol.D, hipMemcpyDeviceToHost));
    if (palette) HIPCHK(hipMemcpy(palette, s->palette, 256 * sizeof(vrt_material), hipMemcpyDeviceToHost));
    return VRT_OK;
}

// ---- settings ------------------------------------------------------------------------------------

void vrt_settings_default(vrt_settings* s)
{
    if (!s) return;
    memset(s, 0, sizeof *s);
    s->ao_samples = 4;                       // voxel_render_settings.hpp:33
    s->ambient_intensity = 1.0f;             // :34
    const float inv = 0.57735026918962576f;  // glm::normalize(vec3(1)), :39
    s->light_dir[0] = s->light_dir[1] = s->light_dir[2] = inv;
    s->light_intensity = 1.0f;               // :41
    s->light_color[0] = s->light_color[1] = s->light_color[2] = s->light_color[3] = 1.0f;  // :40
    s->max_steps = 512;                      // voxel_volume.frag:68
    s->ao_steps = 64;                        // voxel_volume.frag:219
    s->max_bounces = 5;                      // voxel_volume.frag:69
    s->shadows = 1;
    s->traversal = VRT_TRAVERSAL_AUTO;
    s->flags = 0;
}

void vrt_denoiser_settings_default(vrt_denoiser_settings* s)
{
    if (!s) return;
    s->iterations = 2; s->phi_color0 = 20.4f; s->phi_normal0 = 1e-2f; s->phi_pos0 = 1e-1f; s->step_width = 2.0f;   // :21-29
    s->mode = VRT_DENOISE_CANONICAL;
}

} // extern "C"

// ---- shard helpers ---------------------------------------------------------------------------------

namespace {

int make_shard(const vrt_shard* sh, int H, ShardMap& m, int* max_local_strips)
{
    if (!sh || sh->nranks <= 1) {
        m.rank = 0; m.nranks = 1; m.strip_rows = ceil_div(H, 16) * 16; m.n_local_strips = 1;
        m.tiles_per_strip = m.strip_rows / 16;
        if (max_local_strips) *max_local_strips = 1;
        return VRT_OK;
    }
    if (sh->rank < 0 || sh->rank >= sh->nranks || sh->strip_rows <= 0 || sh->strip_rows % 16 != 0)
        return fail(VRT_ERR_INVALID, "vrt_shard: need 0 <= rank < nranks and strip_rows a positive multiple of 16");
    int nstrips = ceil_div(H, sh->strip_rows);
    m.rank = sh->rank; m.nranks = sh->nranks; m.strip_rows = sh->strip_rows;
    m.n_local_strips = nstrips > sh->rank ? ceil_div(nstrips - sh->rank, sh->nranks) : 0;
    m.tiles_per_strip = sh->strip_rows / 16;
    if (max_local_strips) *max_local_strips = ceil_div(nstrips, sh->nranks);
    return VRT_OK;
}


// Image planes must be device memory: a host pointer handed to a kernel is a GPU fault, not an error code.  The pointer
// set of a render loop repeats from call to call, so the (comparatively slow) attribute query runs only when it changes.
static int check_device_ptrs(vrt_ctx* c, int slot, const void* const* ptrs, int n, const char* what)
{
    uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)n;
    for (int i = 0; i < n; i++) { h ^= (uint64_t)(uintptr_t)ptrs[i]; h *= 0x100000001b3ull; }
    if (h == c->checked_ptrs[slot]) return VRT_OK;
    for (int i = 0; i < n; i++) {
        if (!ptrs[i]) continue;
        hipPointerAttribute_t a;
        hipError_t e = hipPointerGetAttributes(&a, ptrs[i]);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(VRT_ERR_INVALID, std::string(what) + ": image pointer is not device memory (host pointer?)"); }
        if (a.type != hipMemoryTypeDevice && a.type != hipMemoryTypeManaged)
            return fail(VRT_ERR_INVALID, std::string(what) + ": image pointer is not device memory");
    }
    c->checked_ptrs[slot] = h;
    return VRT_OK;
}

} // namespace

extern "C" {

int vrt_shard_rows(int32_t H, const vrt_shard* sh)
{
    ShardMap m; int mx;
    if (make_shard(sh, H, m, &mx) != VRT_OK) return -1;
    if (m.nranks == 1) return H;
    return mx * m.strip_rows;     // packed row count, identical on every rank (short ranks zero-pad)
}

// ---- geometry stage --------------------------------------------------------------------------------

// The next table of the ring, ready to be filled on the host (its previous launch, kTabRing launches ago, has finished).
static int next_table(vrt_ctx* c, int* idx)
{
    const int i = c->tab_next;
    c->tab_next = (c->tab_next + 1) % vrt_ctx::kTabRing;
    if (!c->upload_stream) HIPCHK(hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking));
    if (!c->tab_dev[i]) {
        HIPCHK(hipMalloc((void**)&c->tab_dev[i], sizeof(FrameSlot) * VRT_MAX_TABLE));
        HIPCHK(hipHostMalloc((void**)&c->tab_host[i], sizeof(FrameSlot) * VRT_MAX_TABLE, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&c->tab_uploaded[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->tab_consumed[i], hipEventDisableTiming));
    }
    if (c->tab_busy[i]) { HIPCHK(hipEventSynchronize(c->tab_consumed[i])); c->tab_busy[i] = false; }
    *idx = i;
    return VRT_OK;
}

// One launch of K1 over n frames: n <= VRT_MAX_BATCH slots travel in the kernel arguments, more (<= VRT_MAX_TABLE) in a
// table in device memory.  shards: one strip assignment for all frames (per_frame == 0) or one per frame, all with the
// same nranks and strip_rows.  K2 follows in split mode, which renders one frame at a time.
static int render_frames(vrt_ctx* c, const vrt_scene* s, int n, const vrt_push* pushes, const vrt_settings* st,
                         const vrt_frame* frames, const vrt_shard* shards, int per_frame)
{
    const vrt_push* push = &pushes[0];
    int W = push->screen_size[0], H = push->screen_size[1];
    if (W <= 0 || H <= 0 || W > 32768 || H > 32768) return fail(VRT_ERR_INVALID, "vrt_render_geometry: bad screen_size");
    if ((uint64_t)W * (uint64_t)H >= (1ull << 28))              // K1 addresses a plane with 32-bit byte offsets (16 B per pixel at most)
        return fail(VRT_ERR_UNSUPPORTED, "vrt_render_geometry: frames of 2^28 pixels and more are not supported (a launch indexes full-frame planes with "
                                         "32-bit byte offsets, sharded or not: render such an image as several frames with shifted camera planes; include/vrt.h)");
    for (int f = 0; f < n; f++) {
        const vrt_push& q = pushes[f];
        if (q.screen_size[0] != W || q.screen_size[1] != H)
            return fail(VRT_ERR_INVALID, "vrt_render_geometry_batch: all frames of a batch must have the same screen_size");
        if (q.volume_bounds[0] != (uint32_t)s->d.vol.W || q.volume_bounds[1] != (uint32_t)s->d.vol.H || q.volume_bounds[2] != (uint32_t)s->d.vol.D)
            return fail(VRT_ERR_INVALID, "vrt_render_geometry: push.volume_bounds must equal the scene dimensions (voxel_renderer.cpp:74)");
    }
    if (st->max_bounces > VRT_MAX_BOUNCES) return fail(VRT_ERR_INVALID, "vrt_render_geometry: max_bounces > VRT_MAX_BOUNCES");
    if (st->traversal > VRT_TRAVERSAL_DFJ) return fail(VRT_ERR_INVALID, "vrt_render_geometry: unknown traversal");
    if (s->bricks && st->traversal != VRT_TRAVERSAL_AUTO)
        return fail(VRT_ERR_UNSUPPORTED, "vrt_render_geometry: a brick scene (vrt_scene_from_bricks) renders with VRT_TRAVERSAL_AUTO only");
    if (st->traversal == VRT_TRAVERSAL_DENSE && (uint64_t)s->d.vol.W * (uint64_t)s->d.vol.H * (uint64_t)s->d.vol.D > 0xFFFFFFFFull)
        return fail(VRT_ERR_UNSUPPORTED, "vrt_render_geometry: VRT_TRAVERSAL_DENSE indexes voxels in 32 bits (volumes below 4 GiB)");
    HIPCHK(hipSetDevice(c->device));
    {
        std::vector<const void*> ptrs((size_t)14 * (size_t)n);
        for (int f = 0; f < n; f++) {
            const vrt_frame* frame = &frames[f];
            const void* one[14] = {frame->color8, frame->depth, frame->motion, frame->mask8, frame->position, frame->normal8, frame->color_f,
                                   frame->hit_id, frame->hit_voxel, frame->hit_mask, frame->steps_primary, frame->steps_total, frame->rays_total,
                                   frame->color8_strips};
            memcpy(&ptrs[(size_t)14 * f], one, sizeof one);
        }
        int prc = check_device_ptrs(c, 0, ptrs.data(), 14 * n, "vrt_render_geometry");
        if (prc != VRT_OK) return prc;
    }

    GeomParams p;
    memset(&p, 0, sizeof p);
    p.sc = s->d; p.st = *st;
    if (s->bricks) p.st.traversal = VRT_TRAVERSAL_BRICK;
    // a launch that writes count planes reports the iterations of the reference's loop: it marches through the fields without
    // open cells (with the development flags the planes hold the product march's own counters instead)
    bool counts = false;
    p.sc.vol.count_marched = (st->flags & VRT_FLAG_MARCHED_COUNTS) ? 1u : 0u;
    p.sc.vol.count_lookups = ((st->flags & VRT_FLAG_MARCHED_COUNTS) && (st->flags & VRT_FLAG_LOOKUP_COUNTS)) ? 1u : 0u;
    if (!(st->flags & (VRT_FLAG_DEBUG_PLANES | VRT_FLAG_MARCHED_COUNTS | 2u)))
        for (int f = 0; f < n && !counts; f++) counts = frames[f].steps_primary != nullptr || frames[f].steps_total != nullptr;
    if (s->bricks && counts) p.sc.vol.brick_open = 0u;
    if (!s->bricks && s->open_cells) {
        if (counts) {
            const uint8_t* fields = nullptr;
            int frc = fields_for_counts(c, s, &fields);
            if (frc != VRT_OK) return frc;
            p.sc.vol.df = fields;
        }
    }
    p.n_frames = n; p.W = W; p.H = H;
    if (c->div_w != W || c->div_h != H) { c->div_ok = (screen_div_exact(W) && screen_div_exact(H)) ? 1 : 0; c->div_w = W; c->div_h = H; }
    p.rcp_w = 1.0f / (float)W; p.rcp_h = 1.0f / (float)H; p.fast_screen_div = c->div_ok;
    int max_strips = 1;
    int rc = make_shard(shards, H, p.sh, &max_strips);
    if (rc != VRT_OK) return rc;
    int local_strips = p.sh.n_local_strips;
    if (per_frame && shards) {
        for (int f = 1; f < n; f++) {
            ShardMap m;
            rc = make_shard(&shards[f], H, m, nullptr);
            if (rc != VRT_OK) return rc;
            if (m.nranks != p.sh.nranks || m.strip_rows != p.sh.strip_rows)
                return fail(VRT_ERR_INVALID, "vrt_render_geometry_slots: the frames of a launch must share nranks and strip_rows");
        }
        local_strips = p.sh.nranks > 1 ? max_strips : p.sh.n_local_strips;    // the grid covers the longest assignment
    }
    FrameSlot* slots = p.slot;
    int tab = -1;
    const bool box_off = c->opt.box_rect == 0;                         // development switch: every wave tests the box
    if (n > VRT_MAX_BATCH) {
        rc = next_table(c, &tab);
        if (rc != VRT_OK) return rc;
        slots = c->tab_host[tab];
        p.table = c->tab_dev[tab];
    }
    for (int f = 0; f < n; f++) {
        slots[f].pc = pushes[f]; slots[f].fr = frames[f]; slots[f].rg = raygen_consts(pushes[f]);
        slots[f].shard_rank = (per_frame && shards) ? shards[f].rank : p.sh.rank;
        box_rect(pushes[f], slots[f].box);
        if (box_off) { slots[f].box[0] = slots[f].box[2] = 0; slots[f].box[1] = slots[f].box[3] = 255; }
    }
    // LDS-staged traversals amortise the staging over a 16x16 tile (4 waves); the others run one 8x8 wave per workgroup,
    // which frees a wave slot the moment a wave finishes instead of when its whole tile does
    {
        uint32_t t = st->traversal;
        bool lds_mode = (t == VRT_TRAVERSAL_BITMASK || t == VRT_TRAVERSAL_JUMP);
        p.tile_w = p.tile_h = (lds_mode || (st->flags & 8u)) ? 16 : 8;
    }
    p.tiles_x = ceil_div(W, p.tile_w);
    p.tiles_y_local = local_strips * (p.sh.strip_rows / p.tile_h);
    p.total_tiles = p.tiles_x * p.tiles_y_local;
    p.chunk = p.tiles_x * ceil_div(p.tiles_y_local, 8);      // workgroups per XCD slot (tile rows are dealt round-robin)
    p.wgs_per_frame = (uint32_t)p.chunk * 8u;
    p.xcd_turn = (n > 1 && p.tiles_y_local > 0 && ceil_div(p.tiles_y_local, 8) * 8 * 100 > p.tiles_y_local * 103) ? 1 : 0;
    p.wgs_per_frame_rcp = p.wgs_per_frame ? (uint32_t)(0x100000000ull / (uint64_t)p.wgs_per_frame) : 0u;
    // launches of 8 or more unsharded frames: one screen region per XCD and frame, rotating (block_to_tile, xcd_turn == 2)
    {
        const bool want = c->opt.xcd_regions != 0;                       // development switch
        if (want && n >= 8 && p.sh.nranks == 1 && p.tile_h == 8 && p.tiles_x >= 4 && p.tiles_y_local >= 8) {
            const uint32_t rw = ((uint32_t)p.tiles_x + 1u) / 2u, rh = ((uint32_t)p.tiles_y_local + 3u) / 4u;
            p.xcd_turn = 2;
            p.wgs_per_frame = rw * rh;
            p.wgs_per_frame_rcp = (uint32_t)(0x100000000ull / (uint64_t)p.wgs_per_frame);
            p.tiles_y_rcp = (uint32_t)(0x100000000ull / (uint64_t)rw);
        }
    }
    p.tps = (uint32_t)(p.sh.strip_rows / p.tile_h);
    p.tiles_x_rcp = (uint32_t)(0x100000000ull / (uint64_t)p.tiles_x);
    if (p.xcd_turn != 2) p.tiles_y_rcp = p.tiles_y_local ? (uint32_t)(0x100000000ull / (uint64_t)p.tiles_y_local) : 0u;
    p.tps_rcp = (uint32_t)(0x100000000ull / (uint64_t)p.tps);
    // 1: nothing but primary rays; 2: megakernel (default); 0: split K1 -> records -> K2 (VRT_FLAG_SPLIT_KERNELS)
    p.fused_shade = (st->ao_samples == 0 && st->shadows == 0 && (st->max_bounces == 0 || !s->metallic_voxels)) ? 1 : ((st->flags & VRT_FLAG_SPLIT_KERNELS) ? 0 : 2);
    p.no_bounce = ((st->max_bounces == 0 || !s->metallic_voxels) && c->opt.no_bounce_kernel) ? 1 : 0;
    p.sc.vol.ao_batch = (c->opt.ao_batch && p.sc.vol.df_own) ? 1u : 0u;
    // (16 bits of a chain word count the AO rays that hit; brick scenes keep the stack of hits: the packed chain measured 6 % slower there --
    // 3.46 against 3.25 ms on config 5 -- and 1 % faster on the Mandelbulb; context option packed_bounces = 2 forces it everywhere)
    p.packed_chain = (c->opt.packed_bounces && st->ao_samples <= 0xFFFFu && (!s->bricks || c->opt.packed_bounces >= 2)) ? 1 : 0;
    // default traversal and budgets the recovery of positions from sideDist is exact for: the hand-written look-up loop
    // (vrt_traverse.h trace_df_fast) for every ray of the frame
    {
        const bool want = c->opt.fast_loop != 0;                          // development switch
        const bool df = st->traversal == VRT_TRAVERSAL_AUTO || st->traversal == VRT_TRAVERSAL_DF;
        const bool sec = p.fused_shade != 1;
        bool ok = want && df && s->d.vol.df_fast && st->max_steps >= 1 && st->max_steps <= 1024 && p.tile_h == 8 &&
                  !(st->flags & (VRT_FLAG_DEBUG_PLANES | 2u)) && (!sec || st->ao_samples == 0 || (st->ao_steps >= 1 && st->ao_steps <= 1024));
        for (int f = 0; f < n && ok; f++) ok = frames[f].hit_voxel == nullptr;      // the fast loop keeps no mapPos: no hit_voxel plane
        p.fast_loop = ok ? ((counts || (st->flags & VRT_FLAG_MARCHED_COUNTS)) ? 2 : 1) : 0;      // (2: the loops' counting twins -- every launch that fills iteration-count planes)
        if (s->bricks && (counts || (st->flags & (VRT_FLAG_MARCHED_COUNTS | VRT_FLAG_DEBUG_PLANES | 2u)))) p.fast_loop = 2;      // (bricks: the march with its counters)
        // ... and the primary rays' long runs by threshold (df_prim_loop): launches that report no iteration counts (the loop keeps
        // none), axis step counts the position recovery is exact for, a budget worth not counting
        const int dmax = s->d.vol.W > s->d.vol.H ? (s->d.vol.W > s->d.vol.D ? s->d.vol.W : s->d.vol.D) : (s->d.vol.H > s->d.vol.D ? s->d.vol.H : s->d.vol.D);
        p.sc.vol.df_thresh = (ok && c->opt.thresh_runs && !counts && dmax <= 1022 && st->max_steps >= 32) ? 1u : 0u;
        // (brick scenes: the generic loop's form of the same, brick_march_thresh; its positions come from per-run differences)
        if (s->bricks) p.sc.vol.df_thresh = (c->opt.thresh_runs && !counts && !(st->flags & (VRT_FLAG_DEBUG_PLANES | 2u)) && st->max_steps >= 32) ? 1u : 0u;
    }
    // the sky texel of waves that cannot hit anything by vrt_sky.h: launches whose frames hold the reference's targets only
    // (a diagnostic plane wants values the short path does not make), pixel offsets that fit 32 bits, a sky the bound admits
    {
        bool ok = c->opt.sky_fast != 0 && s->d.skyk.w != 0u && !(st->flags & (VRT_FLAG_DEBUG_PLANES | 2u));
        for (int f = 0; f < n && ok; f++)
            ok = !frames[f].color_f && !frames[f].hit_voxel && !frames[f].hit_mask && !frames[f].steps_primary && !frames[f].steps_total && !frames[f].rays_total;
        p.sky_fast = ok ? 1 : 0;
    }
    p.occ2_bytes = s->occ2_bytes; p.occ3_bytes = s->occ3_bytes;
    p.occ_in_lds = ((size_t)s->occ2_bytes + s->occ3_bytes <= 65536) ? 1 : 0;
    if (!p.fused_shade) {
        size_t px = (size_t)W * (size_t)H;
        if (c->records_px < px) {
            HIPCHK(hipStreamSynchronize(c->stream));
            if (c->records) hipFree(c->records);
            if (c->hit_list) hipFree(c->hit_list);
            c->records = nullptr; c->hit_list = nullptr; c->records_px = 0;
            HIPCHK(hipMalloc((void**)&c->records, px * sizeof(uint4)));
            HIPCHK(hipMalloc((void**)&c->hit_list, (px + 1) * sizeof(uint32_t)));
            c->records_px = px;
        }
        p.records = c->records;
        p.hit_list = c->hit_list;
        p.hit_count = c->hit_list + c->records_px;
        HIPCHK(hipMemsetAsync(p.hit_count, 0, sizeof(uint32_t), c->stream));
    }
    if (p.total_tiles == 0) return VRT_OK;
    {
        TileMap& m = p.map;
        bool six = p.sky_fast != 0;                                    // (sky_fast: no diagnostic plane, no color_f)
        for (int f = 0; f < n && six; f++)
            six = frames[f].color8 && frames[f].depth && frames[f].motion && frames[f].mask8 && frames[f].position && frames[f].normal8 &&
                  !frames[f].hit_id && !frames[f].color8_strips;
        m.flags = (st->flags & 0xFFFFu) | (p.sky_fast ? VRT_MAPFLAG_SKY_FAST : 0u) | (six ? VRT_MAPFLAG_SIX : 0u); m.n_frames = p.n_frames; m.xcd_turn = p.xcd_turn;
        m.wgs_per_frame = p.wgs_per_frame; m.wgs_per_frame_rcp = p.wgs_per_frame_rcp;
        m.tiles_x = p.tiles_x; m.tiles_x_rcp = p.tiles_x_rcp; m.tiles_y_local = p.tiles_y_local; m.tiles_y_rcp = p.tiles_y_rcp;
        m.tps = p.tps; m.tps_rcp = p.tps_rcp; m.tile = p.tile_h; m.nranks = p.sh.nranks; m.strip_rows = p.sh.strip_rows;
        m.W = p.W; m.H = p.H;
    }
    if (tab >= 0) {
        HIPCHK(hipMemcpyAsync(c->tab_dev[tab], c->tab_host[tab], sizeof(FrameSlot) * (size_t)n, hipMemcpyHostToDevice, c->upload_stream));
        HIPCHK(hipEventRecord(c->tab_uploaded[tab], c->upload_stream));
        HIPCHK(hipStreamWaitEvent(c->stream, c->tab_uploaded[tab], 0));
    }
    if (c->timing) HIPCHK(hipEventRecord(c->ev_geo0, c->stream));
    // primary rays only, the reference's targets only: the hits' colours from the table (made anew when the settings or the
    // scene's sky have changed since it was made)
    if (p.fused_shade == 1 && c->opt.hit_table && p.sky_fast) {
        if (!c->hit_colors) { HIPCHK(hipMalloc((void**)&c->hit_colors, 256 * 64 * sizeof(uint32_t))); c->hit_scene_gen = 0; }
        if (c->hit_scene_gen != s->shade_gen || memcmp(&c->hit_settings, st, sizeof *st) != 0) {
            HIPCHK(launch_hit_colors(p, c->hit_colors, c->stream));
            c->hit_scene_gen = s->shade_gen; c->hit_settings = *st;
        }
        p.hit_colors = c->hit_colors;
    }
    // tile tags: dense scenes, frames with a box rectangle, launches that do not report the reference's iteration counts
    int tag_buf = -1;
    {
        bool want = c->opt.tile_tags != 0 && s->cells_ok && !counts && W <= 8128 && H <= 8128;
        bool any = false;
        for (int f = 0; f < n && want && !any; f++) any = !(slots[f].box[0] == 0 && slots[f].box[1] == 255 && slots[f].box[2] == 0 && slots[f].box[3] == 255);
        const bool tags = want && any;
        {
            // (in the launch's local rows of 8x8 blocks: vrt_device.hip k_tile_tags)
            p.tags_x = tags ? (uint32_t)(p.tiles_x * (p.tile_w / 8)) : 0u; p.tags_y = tags ? (uint32_t)(p.tiles_y_local * (p.tile_h / 8)) : 0u;
            p.tags_per_frame = p.tags_x * p.tags_y + 1u;
            // (without tags: one word per frame that says "trace every block")
            const size_t words = (size_t)p.tags_per_frame * (size_t)n;
            const int b = c->tag_flip; c->tag_flip ^= 1;
            tag_buf = b;
            if (c->tile_tags_words[b] < words) {
                HIPCHK(hipStreamSynchronize(c->stream));
                if (c->tag_stream) HIPCHK(hipStreamSynchronize(c->tag_stream));
                if (c->tile_tags[b]) hipFree(c->tile_tags[b]);
                c->tile_tags[b] = nullptr; c->tile_tags_words[b] = 0;
                HIPCHK(hipMalloc((void**)&c->tile_tags[b], words * sizeof(uint32_t)));
                HIPCHK(hipMemsetAsync(c->tile_tags[b], 0, words * sizeof(uint32_t), c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                c->tile_tags_words[b] = words;
                c->tag_read_valid[b] = false;
            }
            if (++c->tile_gen == 0u) {                                    // (wrapped: stale tags could match again)
                HIPCHK(hipStreamSynchronize(c->stream));
                if (c->tag_stream) HIPCHK(hipStreamSynchronize(c->tag_stream));
                for (int q = 0; q < 2; q++)
                    if (c->tile_tags[q]) HIPCHK(hipMemsetAsync(c->tile_tags[q], 0, c->tile_tags_words[q] * sizeof(uint32_t), c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                c->tile_gen = 1u;
            }
            p.tile_tags = c->tile_tags[b]; p.tile_gen = c->tile_gen;
            if (tags) {
                p.cells = s->cells; p.n_cells = s->n_cells; p.cell_size = s->bricks ? 8u : 4u;
                // The tags depend on the cameras alone, not on anything the context's stream is still computing: while that stream is
                // busy (the caller runs ahead of the device: a frame loop, a batch loop) they are made on a stream of their own, under
                // the previous launch's tail, and the launch waits for an event instead of a kernel; on an idle device the second
                // stream would only add the event's latency.  (context option tags_async = 0: always on the context's stream)
                const bool async = c->opt.tags_async != 0 && hipStreamQuery(c->stream) == hipErrorNotReady;
                if (async) {
                    if (!c->tag_stream) {
                        HIPCHK(hipStreamCreateWithFlags(&c->tag_stream, hipStreamNonBlocking));
                        for (int q = 0; q < 2; q++) {
                            HIPCHK(hipEventCreateWithFlags(&c->tag_done[q], hipEventDisableTiming));
                            HIPCHK(hipEventCreateWithFlags(&c->tag_read[q], hipEventDisableTiming));
                        }
                    }
                    if (c->tag_read_valid[b]) HIPCHK(hipStreamWaitEvent(c->tag_stream, c->tag_read[b], 0));     // the launch that last read this buffer
                    if (tab >= 0) HIPCHK(hipStreamWaitEvent(c->tag_stream, c->tab_uploaded[tab], 0));            // the slots the tag kernel reads
                    HIPCHK(launch_tile_tags(p, c->tag_stream));
                    HIPCHK(hipEventRecord(c->tag_done[b], c->tag_stream));
                    HIPCHK(hipStreamWaitEvent(c->stream, c->tag_done[b], 0));
                } else {
                    HIPCHK(launch_tile_tags(p, c->stream));
                }
            } else {
                // every frame's one word = tile_gen
                std::vector<uint32_t> ones((size_t)n, c->tile_gen);
                HIPCHK(hipMemcpyAsync(c->tile_tags[b], ones.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
            }
        }
    }
    HIPCHK(launch_primary(p, c->stream));
    if (tag_buf >= 0 && c->tag_stream) { HIPCHK(hipEventRecord(c->tag_read[tag_buf], c->stream)); c->tag_read_valid[tag_buf] = true; }
    if (tab >= 0) { HIPCHK(hipEventRecord(c->tab_consumed[tab], c->stream)); c->tab_busy[tab] = true; }
    if (c->timing) HIPCHK(hipEventRecord(c->ev_prim1, c->stream));
    if (!p.fused_shade) HIPCHK(launch_shade(p, c->stream));
    if (c->timing) { HIPCHK(hipEventRecord(c->ev_geo1, c->stream)); c->have_geo = true; }
    return VRT_OK;
}

static int render_many(vrt_ctx* c, const vrt_scene* s, int32_t n, const vrt_push* pushes, const vrt_settings* st,
                       const vrt_frame* frames, const vrt_shard* shards, int per_frame)
{
    // the split form keeps one frame's hit records: one frame per launch there
    const bool split = (st->flags & VRT_FLAG_SPLIT_KERNELS) && !(st->ao_samples == 0 && st->shadows == 0 && st->max_bounces == 0);
    const int per_launch = split ? 1 : VRT_MAX_TABLE;
    for (int f0 = 0; f0 < n; f0 += per_launch) {
        int m = n - f0 < per_launch ? n - f0 : per_launch;
        int rc = render_frames(c, s, m, pushes + f0, st, frames + f0, (per_frame && shards) ? shards + f0 : shards, per_frame);
        if (rc != VRT_OK) return rc;
    }
    return VRT_OK;
}

int vrt_render_geometry(vrt_ctx* c, const vrt_scene* s, const vrt_push* push, const vrt_settings* st,
                        const vrt_frame* frame, const vrt_shard* shard)
{
    if (!c || !s || !push || !st || !frame) return fail(VRT_ERR_INVALID, "vrt_render_geometry: NULL argument");
    return render_frames(c, s, 1, push, st, frame, shard, 0);
}

int vrt_render_geometry_batch(vrt_ctx* c, const vrt_scene* s, int32_t n, const vrt_push* pushes, const vrt_settings* st,
                              const vrt_frame* frames, const vrt_shard* shard)
{
    if (!c || !s || !pushes || !st || !frames) return fail(VRT_ERR_INVALID, "vrt_render_geometry_batch: NULL argument");
    if (n < 0) return fail(VRT_ERR_INVALID, "vrt_render_geometry_batch: n < 0");
    return render_many(c, s, n, pushes, st, frames, shard, 0);
}

int vrt_render_geometry_slots(vrt_ctx* c, const vrt_scene* s, int32_t n, const vrt_push* pushes, const vrt_settings* st,
                              const vrt_frame* frames, const vrt_shard* shards)
{
    if (!c || !s || !pushes || !st || !frames || !shards) return fail(VRT_ERR_INVALID, "vrt_render_geometry_slots: NULL argument");
    if (n < 0) return fail(VRT_ERR_INVALID, "vrt_render_geometry_slots: n < 0");
    return render_many(c, s, n, pushes, st, frames, shards, 1);
}

// ---- denoiser stage --------------------------------------------------------------------------------

static int tap_reach(const vrt_denoiser_settings* ds, int pass)
{
    float sw = (float)pass * ds->step_width + 1.0f;         // denoiser_stage.cpp:151
    int r = (int)sw; if ((float)r < sw) r++;
    return r;
}

int vrt_denoise_halo_rows(const vrt_denoiser_settings* ds)
{
    if (!ds) return 0;
    int h = 0;
    for (int i = 0; i < ds->iterations; i++) h += tap_reach(ds, i);
    return h;
}

int vrt_denoise(vrt_ctx* c, int32_t W, int32_t H, const vrt_denoiser_settings* ds,
                const uint8_t* color_in, const int8_t* normal8, const float* position,
                uint8_t* target0, uint8_t* target1, const vrt_shard* shard, const uint8_t** result)
{
    if (!c || !ds || !color_in || !normal8 || !position || !result) return fail(VRT_ERR_INVALID, "vrt_denoise: NULL argument");
    if (ds->iterations < 0 || ds->iterations > 10) return fail(VRT_ERR_INVALID, "vrt_denoise: iterations must be in 0..10 (MAX_DENOISER_PASSES)");
    if (ds->mode < 0 || ds->mode > 3) return fail(VRT_ERR_INVALID, "vrt_denoise: mode must be VRT_DENOISE_CANONICAL or _AS_SHIPPED, optionally | VRT_DENOISE_FAST");
    if (ds->iterations > 0 && !target0) return fail(VRT_ERR_INVALID, "vrt_denoise: target0 is NULL");
    if (ds->iterations > 1 && !target1) return fail(VRT_ERR_INVALID, "vrt_denoise: target1 is NULL");
    if (!(ds->phi_color0 > 0.0f) || !(ds->phi_normal0 > 0.0f) || !(ds->phi_pos0 > 0.0f))
        return fail(VRT_ERR_INVALID, "vrt_denoise: phi parameters must be > 0 (SURVEY 9.4-E)");
    if (!(ds->step_width >= 0.0f)) return fail(VRT_ERR_INVALID, "vrt_denoise: step_width must be >= 0");
    if (W <= 0 || H <= 0) return fail(VRT_ERR_INVALID, "vrt_denoise: bad size");
    HIPCHK(hipSetDevice(c->device));
    {
        const void* ptrs[5] = {color_in, normal8, position, target0, target1};
        int prc = check_device_ptrs(c, 1, ptrs, 5, "vrt_denoise");
        if (prc != VRT_OK) return prc;
    }
    DenoiseParams p;
    memset(&p, 0, sizeof p);
    int rc = make_shard(shard, H, p.sh, nullptr);
    if (rc != VRT_OK) return rc;
    p.normal = normal8; p.position = position; p.W = W; p.H = H; p.mode = ds->mode;
    p.tile16 = c->opt.denoise_th16; p.no_packed = c->opt.denoise_packed ? 0 : 1;
    p.no_pair = c->opt.denoise_pair ? 0 : 1; p.no_p0 = c->opt.denoise_p0 ? 0 : 1; p.pair_wgs = c->opt.denoise_pair_wgs;
    uint8_t* targets[2] = {target0, target1};
    const uint8_t* last = color_in;
    // which passes take the verified form (an integral tap offset, a guard worth having)
    double guards[10];
    bool any_verified = false;
    c->den_last_passes = 0;
    for (int i = 0; i < ds->iterations; i++) {
        guards[i] = INFINITY;
        if (!c->opt.denoise_verified || (size_t)W * (size_t)H >= (1u << 28)) continue;
        if (i == 0) { guards[0] = denoise_guard_pass0(); any_verified = true; continue; }      // pass 0: a plain blur, tap offset 1
        const float inv = 1.0f / (float)i;                     // the pass' parameters as the loop below makes them
        guards[i] = denoise_guard((double)(inv * ds->phi_color0), (double)(inv * ds->phi_normal0), (double)(inv * ds->phi_pos0),
                                  (double)((float)i * ds->step_width + 1.0f), (ds->mode & 1) == VRT_DENOISE_AS_SHIPPED);
        if (guards[i] <= kDenGuardMax) any_verified = true;
    }
    const bool counting = any_verified && !(ds->mode & VRT_DENOISE_FAST) && c->opt.denoise_count;
    if (counting) {
        if (!c->den_counts) HIPCHK(hipMalloc((void**)&c->den_counts, 10 * VRT_DENOISE_SEGS * sizeof(uint32_t)));
        HIPCHK(hipMemsetAsync(c->den_counts, 0, (size_t)ds->iterations * VRT_DENOISE_SEGS * sizeof(uint32_t), c->stream));
    }
    if (c->timing) HIPCHK(hipEventRecord(c->ev_den0, c->stream));
    for (int i = 0; i < ds->iterations; i++) {                 // denoiser_stage.cpp:204-255
        int ping = i % 2;
        float inv = 1.0f / (float)i;                           // pass 0: +inf (denoiser_stage.cpp:148-150)
        p.phi_color = inv * ds->phi_color0;
        p.phi_normal = inv * ds->phi_normal0;
        p.phi_pos = inv * ds->phi_pos0;
        p.step_width = (float)i * ds->step_width + 1.0f;
        {
            const float log2e = 1.44269504088896341f;
            p.kc = log2e / p.phi_color; p.kp = log2e / p.phi_pos; p.kn = log2e / (p.phi_normal * (p.step_width * p.step_width));   // pass 0: all 0
            {
                const float sw2 = p.step_width * p.step_width;
                auto ok = [](float v) { return v >= 0x1p-20f && v <= 0x1p20f; };
                p.packed_ok = (ok(p.phi_color) && ok(p.phi_normal) && ok(p.phi_pos) && ok(sw2)) ? 1 : 0;
                p.rc = 1.0f / p.phi_color; p.rn = 1.0f / p.phi_normal; p.rp = 1.0f / p.phi_pos; p.rs = 1.0f / sw2;
            }
        }
        p.verified = 0;
        if (guards[i] <= kDenGuardMax) {
            const double log2e = 1.4426950408889634;
            const double sw = (double)p.step_width;
            p.vkc = (float)(log2e / ((double)p.phi_color * 255.0 * 255.0));
            p.vkn = (float)(log2e / ((double)p.phi_normal * sw * sw * 127.0 * 127.0));
            p.vkp = (float)(log2e / (double)p.phi_pos);
            const double g = c->opt.denoise_guard_div8 ? guards[i] * 0.125 : guards[i];
            p.guard = std::nextafterf((float)g, 1.0f);
            p.fix_counts = counting ? c->den_counts + (size_t)i * VRT_DENOISE_SEGS : nullptr;
            p.verified = 1;
            if (counting) c->den_last_passes |= 1 << i;
        }
        p.color_in = last; p.color_out = targets[ping];
        int ext = 0;
        if (p.sh.nranks > 1) for (int j = i + 1; j < ds->iterations; j++) ext += tap_reach(ds, j);
        p.extend = ext;
        if (p.sh.n_local_strips > 0) HIPCHK(launch_denoise_pass(p, c->stream));
        last = targets[ping];
    }
    if (c->timing) { HIPCHK(hipEventRecord(c->ev_den1, c->stream)); c->have_den = true; }
    *result = last;
    return VRT_OK;
}

int vrt_denoise_guard(const vrt_denoiser_settings* ds, int32_t pass, float* guard)
{
    if (!ds || !guard) return fail(VRT_ERR_INVALID, "vrt_denoise_guard: NULL argument");
    if (pass < 0 || pass > 9) return fail(VRT_ERR_INVALID, "vrt_denoise_guard: pass must be 0..9");
    if (pass == 0) { *guard = (float)denoise_guard_pass0(); return VRT_OK; }
    const float inv = 1.0f / (float)pass;
    *guard = (float)denoise_guard((double)(inv * ds->phi_color0), (double)(inv * ds->phi_normal0), (double)(inv * ds->phi_pos0),
                                  (double)((float)pass * ds->step_width + 1.0f), (ds->mode & 1) == VRT_DENOISE_AS_SHIPPED);
    return VRT_OK;
}

int vrt_debug_denoise_redone(vrt_ctx* c, int32_t pass, uint32_t* pixels)
{
    if (!c || !pixels) return fail(VRT_ERR_INVALID, "vrt_debug_denoise_redone: NULL argument");
    if (pass < 0 || pass > 9) return fail(VRT_ERR_INVALID, "vrt_debug_denoise_redone: pass must be 0..9");
    *pixels = 0;
    if (!(c->den_last_passes & (1 << pass)) || !c->den_counts) return VRT_OK;      // the pass did not take the verified form
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint32_t counts[VRT_DENOISE_SEGS];
    HIPCHK(hipMemcpy(counts, c->den_counts + (size_t)pass * VRT_DENOISE_SEGS, sizeof counts, hipMemcpyDeviceToHost));
    uint64_t n = 0;
    for (uint32_t v : counts) n += v;
    *pixels = (uint32_t)n;
    return VRT_OK;
}

// ---- strip packing ---------------------------------------------------------------------------------

static int rows_call(vrt_ctx* c, const void* src, void* dst, int W, int H, int bpp, const vrt_shard* sh,
                     int halo, int dir, int unpack)
{
    if (!c || !src || !dst) return fail(VRT_ERR_INVALID, "strip copy: NULL argument");
    if (W <= 0 || H <= 0 || bpp <= 0 || halo < 0) return fail(VRT_ERR_INVALID, "strip copy: bad size");
    HIPCHK(hipSetDevice(c->device));
    RowsParams p;
    memset(&p, 0, sizeof p);
    int mx = 1;
    int rc = make_shard(sh, H, p.sh, &mx);
    if (rc != VRT_OK) return rc;
    if (halo > p.sh.strip_rows) return fail(VRT_ERR_INVALID, "strip copy: halo larger than strip_rows");
    p.src = (const uint8_t*)src; p.dst = (uint8_t*)dst; p.W = W; p.H = H; p.bpp = bpp;
    p.halo = halo; p.dir = dir; p.unpack = unpack;
    int rows = halo ? mx * halo : (p.sh.nranks == 1 ? H : mx * p.sh.strip_rows);
    HIPCHK(launch_rows(p, rows, c->stream));
    return VRT_OK;
}

int vrt_pack_rows(vrt_ctx* c, const void* full, void* packed, int32_t W, int32_t H, int32_t bpp, const vrt_shard* sh)
{ return rows_call(c, full, packed, W, H, bpp, sh, 0, 0, 0); }

int vrt_unpack_rows(vrt_ctx* c, const void* packed, void* full, int32_t W, int32_t H, int32_t bpp, const vrt_shard* sh)
{ return rows_call(c, packed, full, W, H, bpp, sh, 0, 0, 1); }

int vrt_pack_halo(vrt_ctx* c, const void* full, void* packed, int32_t W, int32_t H, int32_t bpp,
                  const vrt_shard* sh, int32_t halo, int32_t dir)
{
    if (halo <= 0 || (dir != -1 && dir != 1)) return fail(VRT_ERR_INVALID, "vrt_pack_halo: halo > 0 and dir = +-1 required");
    return rows_call(c, full, packed, W, H, bpp, sh, halo, dir, 0);
}

int vrt_unpack_halo(vrt_ctx* c, const void* packed, void* full, int32_t W, int32_t H, int32_t bpp,
                    const vrt_shard* sh, int32_t halo, int32_t dir)
{
    if (halo <= 0 || (dir != -1 && dir != 1)) return fail(VRT_ERR_INVALID, "vrt_unpack_halo: halo > 0 and dir = +-1 required");
    return rows_call(c, packed, full, W, H, bpp, sh, halo, dir, 1);
}

// n images per launch (chunks of VRT_ROWS_BATCH).  shards: one map for all images (per_image == 0) or one per image.
static int rows_batch_call(vrt_ctx* c, int n, const void* const* src, void* const* dst, int W, int H, int bpp,
                           const vrt_shard* shards, int per_image, int unpack, int halo = 0, int dir = 0)
{
    if (!c || !src || !dst) return fail(VRT_ERR_INVALID, "strip copy (batch): NULL argument");
    if (n < 0 || W <= 0 || H <= 0 || bpp <= 0) return fail(VRT_ERR_INVALID, "strip copy (batch): bad size");
    HIPCHK(hipSetDevice(c->device));
    for (int i0 = 0; i0 < n; i0 += VRT_ROWS_BATCH) {
        const int m = n - i0 < VRT_ROWS_BATCH ? n - i0 : VRT_ROWS_BATCH;
        RowsBatchParams p;
        memset(&p, 0, sizeof p);
        p.W = W; p.H = H; p.bpp = bpp; p.unpack = unpack; p.halo = halo; p.dir = dir;
        int rows = 0;
        for (int k = 0; k < m; k++) {
            if (!src[i0 + k] || !dst[i0 + k]) return fail(VRT_ERR_INVALID, "strip copy (batch): NULL image pointer");
            int mx = 1;
            int rc = make_shard(per_image ? &shards[i0 + k] : shards, H, p.sh[k], &mx);
            if (rc != VRT_OK) return rc;
            p.src[k] = (const uint8_t*)src[i0 + k]; p.dst[k] = (uint8_t*)dst[i0 + k];
            if (halo > p.sh[k].strip_rows) return fail(VRT_ERR_INVALID, "strip copy (batch): halo larger than strip_rows");
            int r = halo ? mx * halo : (p.sh[k].nranks == 1 ? H : mx * p.sh[k].strip_rows);
            rows = r > rows ? r : rows;
        }
        HIPCHK(launch_rows_batch(p, rows, m, c->stream));
    }
    return VRT_OK;
}

int vrt_pack_rows_batch(vrt_ctx* c, int32_t n, const void* const* full, void* const* packed, int32_t W, int32_t H, int32_t bpp,
                        const vrt_shard* shard)
{ return rows_batch_call(c, n, full, packed, W, H, bpp, shard, 0, 0); }

int vrt_unpack_rows_batch(vrt_ctx* c, int32_t n, const void* const* packed, void* const* full, int32_t W, int32_t H, int32_t bpp,
                          const vrt_shard* shards)
{
    if (!shards) return fail(VRT_ERR_INVALID, "vrt_unpack_rows_batch: one vrt_shard per image is required");
    return rows_batch_call(c, n, packed, full, W, H, bpp, shards, 1, 1);
}

int vrt_pack_halo_batch(vrt_ctx* c, int32_t n, const void* const* full, void* const* packed, int32_t W, int32_t H, int32_t bpp,
                        const vrt_shard* shards, int32_t halo, int32_t dir)
{
    if (!shards) return fail(VRT_ERR_INVALID, "vrt_pack_halo_batch: one vrt_shard per image is required");
    if (halo <= 0 || (dir != -1 && dir != 1)) return fail(VRT_ERR_INVALID, "vrt_pack_halo_batch: halo > 0 and dir = +-1 required");
    return rows_batch_call(c, n, full, packed, W, H, bpp, shards, 1, 0, halo, dir);
}

int vrt_unpack_halo_batch(vrt_ctx* c, int32_t n, const void* const* packed, void* const* full, int32_t W, int32_t H, int32_t bpp,
                          const vrt_shard* shards, int32_t halo, int32_t dir)
{
    if (!shards) return fail(VRT_ERR_INVALID, "vrt_unpack_halo_batch: one vrt_shard per image is required");
    if (halo <= 0 || (dir != -1 && dir != 1)) return fail(VRT_ERR_INVALID, "vrt_unpack_halo_batch: halo > 0 and dir = +-1 required");
    return rows_batch_call(c, n, packed, full, W, H, bpp, shards, 1, 1, halo, dir);
}

size_t vrt_halo_bytes(int32_t W, int32_t H, int32_t bpp, const vrt_shard* sh, int32_t halo)
{
    ShardMap m; int mx = 1;
    if (make_shard(sh, H, m, &mx) != VRT_OK) return 0;
    return (size_t)mx * (size_t)halo * (size_t)W * (size_t)bpp;
}

// ---- presentation / temporal helpers ---------------------------------------------------------------

int vrt_blit(vrt_ctx* c, const void* src_rgba8, int32_t sw, int32_t sh, void* dst_rgba8, int32_t tw, int32_t th)
{
    if (!c || !src_rgba8 || !dst_rgba8) return fail(VRT_ERR_INVALID, "vrt_blit: NULL argument");
    if (sw <= 0 || sh <= 0 || tw <= 0 || th <= 0) return fail(VRT_ERR_INVALID, "vrt_blit: bad size");
    if (src_rgba8 == dst_rgba8) return fail(VRT_ERR_INVALID, "vrt_blit: source and target must differ");
    HIPCHK(hipSetDevice(c->device));
    BlitParams p;
    p.src = (const uint8_t*)src_rgba8; p.dst = (uint8_t*)dst_rgba8; p.sw = sw; p.sh = sh; p.tw = tw; p.th = th;
    HIPCHK(launch_blit(p, c->stream));
    return VRT_OK;
}

int vrt_accumulate(vrt_ctx* c, const void* color_rgba8, void* accum_u32, int32_t W, int32_t H, int32_t reset)
{
    if (!c || !color_rgba8 || !accum_u32) return fail(VRT_ERR_INVALID, "vrt_accumulate: NULL argument");
    if (W <= 0 || H <= 0) return fail(VRT_ERR_INVALID, "vrt_accumulate: bad size");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(launch_accumulate(color_rgba8, accum_u32, (size_t)W * (size_t)H, reset != 0, c->stream));
    return VRT_OK;
}

int vrt_resolve(vrt_ctx* c, const void* accum_u32, void* out_rgba8, int32_t W, int32_t H, uint32_t frames)
{
    if (!c || !accum_u32 || !out_rgba8) return fail(VRT_ERR_INVALID, "vrt_resolve: NULL argument");
    if (W <= 0 || H <= 0) return fail(VRT_ERR_INVALID, "vrt_resolve: bad size");
    if (frames == 0 || frames > (1u << 22)) return fail(VRT_ERR_INVALID, "vrt_resolve: frames must be in 1..2^22");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(launch_resolve(accum_u32, out_rgba8, (size_t)W * (size_t)H, frames, c->stream));
    return VRT_OK;
}

// ffxFsr2GetJitterPhaseCount / ffxFsr2GetJitterOffset as documented in FidelityFX-FSR2's ffx_fsr2.h (the prebuilt
// library the reference links is absent from its tree): phase count int(8 * (display/render)^2), offset
// Halton(2,3)(index % phases + 1) - 0.5 in pixel units.  Host-only arithmetic (two floats per frame).
int32_t vrt_jitter_phase_count(int32_t render_width, int32_t display_width)
{
    if (render_width <= 0 || display_width <= 0) return 0;
    const float ratio = (float)display_width / (float)render_width;
    return (int32_t)(8.0f * (ratio * ratio));
}

static float radical_inverse(int32_t index, int32_t base)
{
    float digit = 1.0f, acc = 0.0f;
    while (index > 0) {
        digit /= (float)base;
        acc += digit * (float)(index % base);
        index /= base;
    }
    return acc;
}

int vrt_jitter_offset(int32_t index, int32_t phase_count, float* jitter_x, float* jitter_y)
{
    if (!jitter_x || !jitter_y) return fail(VRT_ERR_INVALID, "vrt_jitter_offset: NULL argument");
    if (phase_count <= 0 || index < 0) return fail(VRT_ERR_INVALID, "vrt_jitter_offset: index >= 0 and phase_count > 0 required");
    const int32_t k = index % phase_count + 1;
    *jitter_x = radical_inverse(k, 2) - 0.5f;
    *jitter_y = radical_inverse(k, 3) - 0.5f;
    return VRT_OK;
}

// ---- RCCL behind the C-ABI ---------------------------------------------------------------------------

} // extern "C"

#include <dlfcn.h>
#include <rccl/rccl.h>

struct vrt_comm { ncclComm_t comm = nullptr; int rank = 0, nranks = 1; };

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// librccl of the process: a copy that is already loaded wins (dlopen by soname returns it), else /opt/rocm's
Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;                                // (contexts on several threads may ask at once)
    std::call_once(once, [] {
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (r.lib) {
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name))
            SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommInitAll, "ncclCommInitAll");
            SYM(CommDestroy, "ncclCommDestroy"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
            SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv) {
                dlclose(r.lib); r.lib = nullptr;
            }
        }
    });
    return r.lib ? &r : nullptr;
}

int nccl_fail(const char* what, ncclResult_t e)
{
    Rccl* r = rccl();
    return fail(VRT_ERR_HIP, std::string(what) + ": " + ((r && r->GetErrorString) ? r->GetErrorString(e) : "RCCL error"));
}

#define RCCL_OR_FAIL(r) Rccl* r = rccl(); if (!r) return fail(VRT_ERR_UNSUPPORTED, "librccl.so could not be loaded")

} // namespace

extern "C" {

int vrt_comm_unique_id(uint8_t id[128])
{
    if (!id) return fail(VRT_ERR_INVALID, "vrt_comm_unique_id: NULL argument");
    RCCL_OR_FAIL(r);
    ncclUniqueId u;
    ncclResult_t e = r->GetUniqueId(&u);
    if (e != ncclSuccess) return nccl_fail("ncclGetUniqueId", e);
    static_assert(sizeof u == 128, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, 128);
    return VRT_OK;
}

int vrt_comm_init_rank(vrt_ctx* c, int32_t nranks, int32_t rank, const uint8_t id[128], vrt_comm** out)
{
    if (!c || !id || !out) return fail(VRT_ERR_INVALID, "vrt_comm_init_rank: NULL argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(VRT_ERR_INVALID, "vrt_comm_init_rank: need 0 <= rank < nranks");
    RCCL_OR_FAIL(r);
    HIPCHK(hipSetDevice(c->device));
    ncclUniqueId u; memcpy(&u, id, 128);
    vrt_comm* k = new vrt_comm(); k->rank = rank; k->nranks = nranks;
    ncclResult_t e = r->CommInitRank(&k->comm, nranks, u, rank);
    if (e != ncclSuccess) { delete k; return nccl_fail("ncclCommInitRank", e); }
    *out = k;
    return VRT_OK;
}

int vrt_comm_init_all(int32_t n, vrt_ctx* const* ctxs, vrt_comm** out)
{
    if (!ctxs || !out || n < 1) return fail(VRT_ERR_INVALID, "vrt_comm_init_all: bad argument");
    RCCL_OR_FAIL(r);
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; i++) { if (!ctxs[i]) return fail(VRT_ERR_INVALID, "vrt_comm_init_all: NULL context"); devs[(size_t)i] = ctxs[i]->device; }
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++)
        if (devs[(size_t)i] == devs[(size_t)j]) return fail(VRT_ERR_INVALID, "vrt_comm_init_all: two ranks on one device (RCCL wants one GPU per rank)");
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    ncclResult_t e = r->CommInitAll(comms.data(), n, devs.data());
    if (e != ncclSuccess) return nccl_fail("ncclCommInitAll", e);
    for (int i = 0; i < n; i++) { vrt_comm* k = new vrt_comm(); k->comm = comms[(size_t)i]; k->rank = i; k->nranks = n; out[i] = k; }
    return VRT_OK;
}

void vrt_comm_destroy(vrt_comm* k)
{
    if (!k) return;
    Rccl* r = rccl();
    if (r && k->comm) r->CommDestroy(k->comm);
    delete k;
}

int vrt_group_start(void) { RCCL_OR_FAIL(r); ncclResult_t e = r->GroupStart(); return e == ncclSuccess ? VRT_OK : nccl_fail("ncclGroupStart", e); }
int vrt_group_end(void)   { RCCL_OR_FAIL(r); ncclResult_t e = r->GroupEnd();   return e == ncclSuccess ? VRT_OK : nccl_fail("ncclGroupEnd", e); }

int vrt_gather_strips(vrt_ctx* c, vrt_comm* k, int32_t root, const void* send, void* recv, size_t bytes)
{
    if (!c || !k || !send) return fail(VRT_ERR_INVALID, "vrt_gather_strips: NULL argument");
    if (root < 0 || root >= k->nranks) return fail(VRT_ERR_INVALID, "vrt_gather_strips: root out of range");
    if (k->rank == root && !recv) return fail(VRT_ERR_INVALID, "vrt_gather_strips: the root needs a receive buffer");
    RCCL_OR_FAIL(r);
    HIPCHK(hipSetDevice(c->device));
    // a gather as grouped point-to-point operations: every rank's one send travels its own xGMI link to the root, whose
    // nranks receives proceed in parallel
    ncclResult_t e = r->GroupStart();
    if (e != ncclSuccess) return nccl_fail("ncclGroupStart", e);
    if (k->rank == root)
        for (int src = 0; src < k->nranks && e == ncclSuccess; src++)
            e = r->Recv((uint8_t*)recv + (size_t)src * bytes, bytes, ncclUint8, src, k->comm, c->stream);
    if (e == ncclSuccess) e = r->Send(send, bytes, ncclUint8, root, k->comm, c->stream);
    ncclResult_t e2 = r->GroupEnd();
    if (e != ncclSuccess) return nccl_fail("ncclSend / ncclRecv", e);
    if (e2 != ncclSuccess) return nccl_fail("ncclGroupEnd", e2);
    return VRT_OK;
}

// ---- instrumentation -------------------------------------------------------------------------------

int vrt_debug_sky_texels(vrt_ctx* c, const vrt_scene* s, const float* dirs_dev, size_t n, uint32_t* out_dev)
{
    if (!c || !s || !dirs_dev || !out_dev) return fail(VRT_ERR_INVALID, "vrt_debug_sky_texels: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(launch_debug_sky(s->d, dirs_dev, n, out_dev, c->stream));
    return VRT_OK;
}

int vrt_debug_brick_counts(vrt_ctx* c, uint64_t out[4])
{
    if (!c || !out) return fail(VRT_ERR_INVALID, "vrt_debug_brick_counts: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned long long v[4];
    HIPCHK(debug_brick_counts(v));
    for (int i = 0; i < 4; i++) out[i] = v[i];
    return VRT_OK;
}

int vrt_last_timings(vrt_ctx* c, float* primary_ms, float* geometry_ms, float* denoise_ms)
{
    if (!c) return fail(VRT_ERR_INVALID, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (primary_ms) *primary_ms = -1.0f;
    if (geometry_ms) *geometry_ms = -1.0f;
    if (denoise_ms) *denoise_ms = -1.0f;
    if (c->have_geo) {
        HIPCHK(hipEventSynchronize(c->ev_geo1));
        if (primary_ms) HIPCHK(hipEventElapsedTime(primary_ms, c->ev_geo0, c->ev_prim1));
        if (geometry_ms) HIPCHK(hipEventElapsedTime(geometry_ms, c->ev_geo0, c->ev_geo1));
    }
    if (c->have_den) {
        HIPCHK(hipEventSynchronize(c->ev_den1));
        if (denoise_ms) HIPCHK(hipEventElapsedTime(denoise_ms, c->ev_den0, c->ev_den1));
    }
    return VRT_OK;
}

} // extern "C"
