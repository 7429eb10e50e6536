// voxels.hpp -- C++17 host side above the C-ABI (include/vrt.h), mirroring the reference's object surface for
// the hot path: same class names, member names, argument meaning and error behaviour (std::runtime_error with
// the reference's messages).  Header-only; link with libvrt_hip.so.  No Vulkan, no HIP headers needed.
//
//   VoxelRenderSettings (+Fsr/Denoiser/AmbientOcclusion/LightSettings)  source/voxels/voxel_render_settings.hpp:6-59
//   CameraController                                                   source/voxels/resource/camera_controller.cpp:15-44
//   VoxelScene                                                         source/voxels/resource/voxel_scene.hpp:18-34
//   GeometryBuffer / GeometryStage                                     source/voxels/stages/geometry_stage.hpp:19-53
//   DenoiserStage                                                      source/voxels/stages/denoiser_stage.hpp:22-47
//   VoxelRenderer                                                      source/voxels/voxel_renderer.hpp:17-38
//   Engine                                                             source/engine/engine.hpp:71-76 (device + stream only)
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vrt.h"

namespace vrt_host {

inline void check(int rc)
{
    if (rc != VRT_OK) throw std::runtime_error(vrt_last_error());
}

struct vec3 { float x = 0, y = 0, z = 0; };
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline vec3 normalize(vec3 a) { float l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return {a.x / l, a.y / l, a.z / l}; }

// ---- settings (voxel_render_settings.hpp) -------------------------------------------------------------
enum class FsrScaling : uint32_t { NONE = 10, QUALITY = 15, BALANCED = 17, PERFORMANCE = 20, ULTRA_PERFORMANCE = 30 };
struct FsrSettings { bool enable = true; FsrScaling scaling = FsrScaling::BALANCED; };
struct DenoiserSettings {
    bool enable = true; int iterations = 2; float phiColor0 = 20.4f; float phiNormal0 = 1E-2f; float phiPos0 = 1E-1f;
    float stepWidth = 2.0f; int mode = VRT_DENOISE_CANONICAL;
};
struct AmbientOcclusionSettings { int numSamples = 4; float intensity = 1.0f; };
struct LightSettings { vec3 direction = normalize({1.0f, 1.0f, 1.0f}); std::array<float, 4> color{1, 1, 1, 1}; float intensity = 1.0f; };
struct TraceSettings {   // voxel_volume.frag:68-69,219 promoted to runtime knobs
    uint32_t maxRaySteps = 512, aoSteps = 64, maxReflections = 5; bool shadows = true; uint32_t traversal = VRT_TRAVERSAL_AUTO;
};

class VoxelRenderSettings {
public:
    std::array<uint32_t, 2> targetResolution{1920, 1080};
    FsrSettings fsrSetttings{};            // (sic)
    DenoiserSettings denoiserSettings{};
    AmbientOcclusionSettings occlusionSettings{};
    LightSettings lightSettings{};
    TraceSettings traceSettings{};
    std::string voxPath = "../resource/treehouse.vox";
    std::string skyboxPath = "../resource/rustig_koppie.hdr";

    std::array<uint32_t, 2> renderResolution() const     // voxel_render_settings.cpp:3-13
    {
        if (!fsrSetttings.enable) return targetResolution;
        auto scale = [&](uint32_t dim) { return static_cast<uint32_t>((10.0f / static_cast<uint32_t>(fsrSetttings.scaling)) * dim); };
        return {scale(targetResolution[0]), scale(targetResolution[1])};
    }
    vrt_settings toC() const                             // geometry_stage.cpp:135-145
    {
        vrt_settings s{};
        s.ao_samples = (uint32_t)occlusionSettings.numSamples; s.ambient_intensity = occlusionSettings.intensity;
        s.light_dir[0] = lightSettings.direction.x; s.light_dir[1] = lightSettings.direction.y; s.light_dir[2] = lightSettings.direction.z;
        s.light_intensity = lightSettings.intensity;
        for (int i = 0; i < 4; i++) s.light_color[i] = lightSettings.color[i];
        s.max_steps = traceSettings.maxRaySteps; s.ao_steps = traceSettings.aoSteps; s.max_bounces = traceSettings.maxReflections;
        s.shadows = traceSettings.shadows ? 1u : 0u; s.traversal = traceSettings.traversal; s.flags = 0;
        return s;
    }
    vrt_denoiser_settings denoiserToC() const
    {
        return {denoiserSettings.iterations, denoiserSettings.phiColor0, denoiserSettings.phiNormal0, denoiserSettings.phiPos0,
                denoiserSettings.stepWidth, denoiserSettings.mode};
    }
};

// ---- camera (camera_controller.cpp) -------------------------------------------------------------------
class CameraController {
public:
    vec3 position, direction, right, up, normalDir;
    float yaw, pitch, focalLength;
    explicit CameraController(vec3 position = {8, 8, -50}, float yaw = 90.0f, float pitch = 0.0f,
                              float focalLength = static_cast<float>(1 / std::tan((55.0 / 2) * 3.14159265358979323846 / 180)))
        : position(position), yaw(yaw), pitch(pitch), focalLength(focalLength) { updateDirectionVectors(); }
    void updateDirectionVectors()                         // :15-28
    {
        const vec3 worldUp{0.0f, -1.0f, 0.0f};
        const float ry = yaw * 0.01745329251994329576923690768489f, rp = pitch * 0.01745329251994329576923690768489f;
        normalDir = normalize({std::cos(ry) * std::cos(rp), std::sin(rp), std::sin(ry) * std::cos(rp)});
        right = normalize(cross(normalDir, worldUp));
        up = normalize(cross(right, normalDir));
        direction = normalDir * focalLength;
    }
    void update(float delta, float forward = 0.0f, float strafe = 0.0f)   // :30-44, keys replaced by scripted axes
    {
        const float cameraSpeed = 50.0f;
        position = position + normalDir * (cameraSpeed * delta * forward) + right * (cameraSpeed * delta * strafe);
        updateDirectionVectors();
    }
    void mouse(float offsetX, float offsetY)                              // :46-68 with the right button held
    {
        yaw -= offsetX;
        pitch = std::min(std::max(pitch - offsetY, -90.0f), 90.0f);
        updateDirectionVectors();
    }
};

// ---- engine -------------------------------------------------------------------------------------------
class Engine {
public:
    explicit Engine(int device = 0) { check(vrt_ctx_create(device, &ctx)); }
    ~Engine() { vrt_ctx_destroy(ctx); }
    Engine(const Engine&) = delete; Engine& operator=(const Engine&) = delete;
    void waitIdle() { check(vrt_ctx_synchronize(ctx)); }
    vrt_ctx* ctx = nullptr;
};

template <class T> class DeviceBuffer {                   // Buffer / RenderImage storage (engine/resource/buffer.hpp)
public:
    DeviceBuffer(const std::shared_ptr<Engine>& e, size_t count) : engine(e), count(count)
    {
        void* p = nullptr; check(vrt_device_alloc(e->ctx, count * sizeof(T), &p)); ptr = static_cast<T*>(p);
        check(vrt_memset(e->ctx, ptr, 0, count * sizeof(T)));
    }
    ~DeviceBuffer() { vrt_device_free(engine->ctx, ptr); }
    DeviceBuffer(const DeviceBuffer&) = delete; DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    std::vector<T> download() const { std::vector<T> h(count); check(vrt_memcpy_d2h(engine->ctx, h.data(), ptr, count * sizeof(T))); return h; }
private:
    std::shared_ptr<Engine> engine;
public:
    size_t count;
    T* ptr = nullptr;
};

// ---- scene --------------------------------------------------------------------------------------------
class VoxelScene {
public:
    uint32_t width = 0, height = 0, depth = 0;
    VoxelScene(const std::shared_ptr<Engine>& engine, const std::string& filename, const std::string& skyboxFilename = "")
        : engine(engine)                                                                              // voxel_scene.cpp:33
    {
        check(vrt_scene_load_vox_file(engine->ctx, filename.c_str(), &handle)); dims();
        if (!skyboxFilename.empty()) setSkybox(skyboxFilename);                                       // :132
    }
    VoxelScene(const std::shared_ptr<Engine>& engine, const uint8_t* voxels, uint32_t W, uint32_t H, uint32_t D, const vrt_material* palette)
        : engine(engine) { check(vrt_scene_from_dense(engine->ctx, voxels, W, H, D, palette, &handle)); dims(); }
    ~VoxelScene() { vrt_scene_free(engine->ctx, handle); }
    VoxelScene(const VoxelScene&) = delete; VoxelScene& operator=(const VoxelScene&) = delete;
    void setSkybox(const float* rgba, uint32_t w, uint32_t h) { check(vrt_scene_set_sky(engine->ctx, handle, rgba, w, h)); }
    void setSkybox(const std::string& path) { check(vrt_scene_set_sky_file(engine->ctx, handle, path.c_str())); }          // Texture2D(.hdr)
    void setBlueNoise(const std::string& path) { check(vrt_scene_set_blue_noise_file(engine->ctx, handle, path.c_str())); }   // Texture2D(.png)
    void setBlueNoise(const uint8_t* rgba8, uint32_t w, uint32_t h) { check(vrt_scene_set_blue_noise(engine->ctx, handle, rgba8, w, h)); }
    vrt_scene* handle = nullptr;
private:
    void dims() { uint32_t d[3]; check(vrt_scene_info(handle, d)); width = d[0]; height = d[1]; depth = d[2]; }
    std::shared_ptr<Engine> engine;
};

// ---- stages -------------------------------------------------------------------------------------------
struct GeometryBuffer {                                   // geometry_stage.hpp:19-27
    std::shared_ptr<DeviceBuffer<uint8_t>> color, mask; std::shared_ptr<DeviceBuffer<float>> depth, motion, position;
    std::shared_ptr<DeviceBuffer<int8_t>> normal; uint32_t width = 0, height = 0;
};

class GeometryStage {
public:
    GeometryStage(const std::shared_ptr<Engine>& engine, const std::shared_ptr<VoxelRenderSettings>& settings, const std::shared_ptr<VoxelScene>& scene)
        : engine(engine), settings(settings), scene(scene) {}
    GeometryBuffer record(const vrt_push& push, const vrt_shard* shard = nullptr)     // geometry_stage.cpp:106
    {
        auto res = settings->renderResolution();
        if (gb.width != res[0] || gb.height != res[1]) {                              // RENDER_RESIZE recreation (:19-45)
            size_t n = (size_t)res[0] * res[1];
            gb.color = std::make_shared<DeviceBuffer<uint8_t>>(engine, n * 4); gb.mask = std::make_shared<DeviceBuffer<uint8_t>>(engine, n);
            gb.depth = std::make_shared<DeviceBuffer<float>>(engine, n); gb.motion = std::make_shared<DeviceBuffer<float>>(engine, n * 2);
            gb.position = std::make_shared<DeviceBuffer<float>>(engine, n * 4); gb.normal = std::make_shared<DeviceBuffer<int8_t>>(engine, n * 4);
            gb.width = res[0]; gb.height = res[1];
        }
        vrt_frame f{}; f.color8 = gb.color->ptr; f.depth = gb.depth->ptr; f.motion = gb.motion->ptr; f.mask8 = gb.mask->ptr;
        f.position = gb.position->ptr; f.normal8 = gb.normal->ptr;
        vrt_settings st = settings->toC();
        check(vrt_render_geometry(engine->ctx, scene->handle, &push, &st, &f, shard));
        return gb;
    }
    // n camera poses in one launch per 8 frames (vrt_render_geometry_batch); every frame gets its own planes, kept in `batch`
    const std::vector<GeometryBuffer>& recordBatch(const std::vector<vrt_push>& pushes, const vrt_shard* shard = nullptr)
    {
        auto res = settings->renderResolution();
        size_t n = (size_t)res[0] * res[1];
        while (batch.size() < pushes.size()) {
            GeometryBuffer g;
            g.color = std::make_shared<DeviceBuffer<uint8_t>>(engine, n * 4); g.mask = std::make_shared<DeviceBuffer<uint8_t>>(engine, n);
            g.depth = std::make_shared<DeviceBuffer<float>>(engine, n); g.motion = std::make_shared<DeviceBuffer<float>>(engine, n * 2);
            g.position = std::make_shared<DeviceBuffer<float>>(engine, n * 4); g.normal = std::make_shared<DeviceBuffer<int8_t>>(engine, n * 4);
            g.width = res[0]; g.height = res[1];
            batch.push_back(g);
        }
        std::vector<vrt_frame> frames(pushes.size());
        for (size_t k = 0; k < pushes.size(); k++) {
            vrt_frame f{}; const GeometryBuffer& g = batch[k];
            f.color8 = g.color->ptr; f.depth = g.depth->ptr; f.motion = g.motion->ptr; f.mask8 = g.mask->ptr; f.position = g.position->ptr; f.normal8 = g.normal->ptr;
            frames[k] = f;
        }
        vrt_settings st = settings->toC();
        check(vrt_render_geometry_batch(engine->ctx, scene->handle, (int32_t)pushes.size(), pushes.data(), &st, frames.data(), shard));
        return batch;
    }
private:
    std::shared_ptr<Engine> engine; std::shared_ptr<VoxelRenderSettings> settings; std::shared_ptr<VoxelScene> scene; GeometryBuffer gb;
    std::vector<GeometryBuffer> batch;
};

class DenoiserStage {
public:
    DenoiserStage(const std::shared_ptr<Engine>& engine, const std::shared_ptr<VoxelRenderSettings>& settings) : engine(engine), settings(settings) {}
    // returns the device pointer of the image holding the result (one of the ping-pong targets, or colorInput)
    const uint8_t* record(const GeometryBuffer& g, const vrt_shard* shard = nullptr)  // denoiser_stage.cpp:156
    {
        size_t n = (size_t)g.width * g.height * 4;
        if (!targets[0] || targets[0]->count != n) { targets[0] = std::make_shared<DeviceBuffer<uint8_t>>(engine, n); targets[1] = std::make_shared<DeviceBuffer<uint8_t>>(engine, n); }
        vrt_denoiser_settings ds = settings->denoiserToC();
        const uint8_t* result = nullptr;
        check(vrt_denoise(engine->ctx, (int32_t)g.width, (int32_t)g.height, &ds, g.color->ptr, g.normal->ptr, g.position->ptr,
                          targets[0]->ptr, targets[1]->ptr, shard, &result));
        return result;
    }
private:
    std::shared_ptr<Engine> engine; std::shared_ptr<VoxelRenderSettings> settings; std::shared_ptr<DeviceBuffer<uint8_t>> targets[2];
};

// UpscalerStage (upscaler_stage.cpp): update() is the reference's jitter / frame sequence (:59-70); record() replaces the
// FSR2 dispatch (:72-161, prebuilt third party, out of scope) by exact N-frame accumulation + bilinear upscale.
class UpscalerStage {
public:
    float jitterX = 0, jitterY = 0; int frameCount = 0; uint32_t accumulated = 0;
    UpscalerStage(const std::shared_ptr<Engine>& engine, const std::shared_ptr<VoxelRenderSettings>& settings) : engine(engine), settings(settings) {}
    int32_t phaseCount() const { return vrt_jitter_phase_count((int32_t)settings->renderResolution()[0], (int32_t)settings->targetResolution[0]); }
    void update(float delta)                                                           // :59-70
    {
        _deltaMsec = delta * 1000;
        const int32_t jitterPhaseCount = phaseCount();
        check(vrt_jitter_offset(frameCount % jitterPhaseCount, jitterPhaseCount, &jitterX, &jitterY));
        frameCount++;
        if (frameCount > jitterPhaseCount) frameCount = 0;
    }
    void reset() { accumulated = 0; }
    const uint8_t* record(const uint8_t* color, uint32_t w, uint32_t h)
    {
        size_t n = (size_t)w * h * 4;
        if (!accum || accum->count != n) { accum = std::make_shared<DeviceBuffer<uint32_t>>(engine, n); resolved = std::make_shared<DeviceBuffer<uint8_t>>(engine, n); accumulated = 0; }
        size_t tn = (size_t)settings->targetResolution[0] * settings->targetResolution[1] * 4;
        if (!target || target->count != tn) target = std::make_shared<DeviceBuffer<uint8_t>>(engine, tn);
        check(vrt_accumulate(engine->ctx, color, accum->ptr, (int32_t)w, (int32_t)h, accumulated == 0));
        accumulated++;
        check(vrt_resolve(engine->ctx, accum->ptr, resolved->ptr, (int32_t)w, (int32_t)h, accumulated));
        check(vrt_blit(engine->ctx, resolved->ptr, (int32_t)w, (int32_t)h, target->ptr, (int32_t)settings->targetResolution[0], (int32_t)settings->targetResolution[1]));
        return target->ptr;
    }
private:
    std::shared_ptr<Engine> engine; std::shared_ptr<VoxelRenderSettings> settings; float _deltaMsec = 0;
    std::shared_ptr<DeviceBuffer<uint32_t>> accum; std::shared_ptr<DeviceBuffer<uint8_t>> resolved, target;
};

// BlitStage::record + shader/blit.frag (blit_stage.cpp:41-75): centre-cropped bilinear copy to a window-sized target.
class BlitStage {
public:
    BlitStage(const std::shared_ptr<Engine>& engine, const std::shared_ptr<VoxelRenderSettings>& settings) : engine(engine), settings(settings) {}
    const uint8_t* record(const uint8_t* source, uint32_t sw, uint32_t sh, uint32_t tw, uint32_t th)
    {
        size_t n = (size_t)tw * th * 4;
        if (!target || target->count != n) target = std::make_shared<DeviceBuffer<uint8_t>>(engine, n);
        check(vrt_blit(engine->ctx, source, (int32_t)sw, (int32_t)sh, target->ptr, (int32_t)tw, (int32_t)th));
        return target->ptr;
    }
private:
    std::shared_ptr<Engine> engine; std::shared_ptr<VoxelRenderSettings> settings; std::shared_ptr<DeviceBuffer<uint8_t>> target;
};

// ---- renderer (voxel_renderer.cpp) ----------------------------------------------------------------------
class VoxelRenderer {
public:
    VoxelRenderer(const std::shared_ptr<Engine>& engine, const std::shared_ptr<VoxelRenderSettings>& settings, const std::shared_ptr<VoxelScene>& scene)
        : engine(engine), _settings(settings), _scene(scene), _camera(std::make_unique<CameraController>()),
          _geometryStage(std::make_unique<GeometryStage>(engine, settings, scene)), _denoiserStage(std::make_unique<DenoiserStage>(engine, settings)),
          _upscalerStage(std::make_unique<UpscalerStage>(engine, settings)), _blitStage(std::make_unique<BlitStage>(engine, settings)) {}
    CameraController& camera() { return *_camera; }
    UpscalerStage& upscaler() { return *_upscalerStage; }
    void update(float delta, float forward = 0.0f, float strafe = 0.0f)               // :33-39
    { _time += delta; _camera->update(delta, forward, strafe); _upscalerStage->update(delta); }
    vrt_push pushConstants() const                                                     // :72-83
    {
        vrt_push p{}; auto res = _settings->renderResolution();
        p.screen_size[0] = (int32_t)res[0]; p.screen_size[1] = (int32_t)res[1];
        p.volume_bounds[0] = _scene->width; p.volume_bounds[1] = _scene->height; p.volume_bounds[2] = _scene->depth;
        const CameraController& c = *_camera;
        p.cam_pos[0] = c.position.x; p.cam_pos[1] = c.position.y; p.cam_pos[2] = c.position.z; p.cam_pos[3] = 1;
        p.cam_dir[0] = c.direction.x; p.cam_dir[1] = c.direction.y; p.cam_dir[2] = c.direction.z;
        p.cam_up[0] = c.up.x; p.cam_up[1] = c.up.y; p.cam_up[2] = c.up.z;
        p.cam_right[0] = c.right.x; p.cam_right[1] = c.right.y; p.cam_right[2] = c.right.z;
        p.frame = (uint32_t)_upscalerStage->frameCount;                                // :80-82
        p.camera_jitter[0] = _upscalerStage->jitterX; p.camera_jitter[1] = _upscalerStage->jitterY;
        return p;
    }
    // recordCommands (:55-94); returns the RGBA8 image (host copy) and its size in outW / outH.
    // temporal: take the FSR branch (:86-87) through the accumulation stand-in; windowW/H != 0: append the blit (:89).
    std::vector<uint8_t> render(uint32_t* outW = nullptr, uint32_t* outH = nullptr)
    {
        vrt_push push = pushConstants();
        GeometryBuffer g = _geometryStage->record(push);
        const uint8_t* img = _settings->denoiserSettings.enable ? _denoiserStage->record(g) : g.color->ptr;
        uint32_t w = g.width, h = g.height;
        if (temporal && _settings->fsrSetttings.enable) { img = _upscalerStage->record(img, w, h); w = _settings->targetResolution[0]; h = _settings->targetResolution[1]; }
        if (windowW && windowH) { img = _blitStage->record(img, w, h, windowW, windowH); w = windowW; h = windowH; }
        std::vector<uint8_t> host((size_t)w * h * 4);
        check(vrt_memcpy_d2h(engine->ctx, host.data(), img, host.size()));
        gBuffer = g;
        if (outW) *outW = w;
        if (outH) *outH = h;
        return host;
    }
    bool temporal = false; uint32_t windowW = 0, windowH = 0; GeometryBuffer gBuffer;
private:
    std::shared_ptr<Engine> engine; std::shared_ptr<VoxelRenderSettings> _settings; std::shared_ptr<VoxelScene> _scene;
    std::unique_ptr<CameraController> _camera; std::unique_ptr<GeometryStage> _geometryStage; std::unique_ptr<DenoiserStage> _denoiserStage;
    std::unique_ptr<UpscalerStage> _upscalerStage; std::unique_ptr<BlitStage> _blitStage;
    float _time = 0;
};

// ---- one process, several GPUs (no reference analogue; BASELINE north_star) -----------------------------------------------
// The frame cut into 16-row strips, strip s traced by device s % N (every device holds the whole scene), the three planes the
// rest of the frame graph reads -- colour, normal, position -- gathered to device 0 through RCCL (vrt_gather_strips, one
// grouped gather per plane), assembled there, and the denoiser run on the assembled frame.  (The Python host also shards
// the denoiser, with a ring exchange of halo rows: voxel-raytracing_amd/distributed.py.)
class ShardedRenderer {
public:
    ShardedRenderer(const std::vector<std::shared_ptr<Engine>>& engines, const std::shared_ptr<VoxelRenderSettings>& settings,
                    const std::vector<std::shared_ptr<VoxelScene>>& scenes, int stripRows = 16)
        : engines(engines), settings(settings), scenes(scenes), stripRows(stripRows), comms(engines.size(), nullptr)
    {
        if (engines.empty() || engines.size() != scenes.size()) throw std::runtime_error("ShardedRenderer: one scene per engine");
        std::vector<vrt_ctx*> ctxs;
        for (auto& e : engines) ctxs.push_back(e->ctx);
        check(vrt_comm_init_all((int32_t)ctxs.size(), ctxs.data(), comms.data()));
        for (size_t i = 0; i < engines.size(); i++) stages.push_back(std::make_unique<GeometryStage>(engines[i], settings, scenes[i]));
        denoiser = std::make_unique<DenoiserStage>(engines[0], settings);
    }
    ~ShardedRenderer() { for (vrt_comm* c : comms) vrt_comm_destroy(c); }
    ShardedRenderer(const ShardedRenderer&) = delete; ShardedRenderer& operator=(const ShardedRenderer&) = delete;

    std::vector<uint8_t> render(const vrt_push& push, uint32_t* outW = nullptr, uint32_t* outH = nullptr)
    {
        const int n = (int)engines.size();
        auto res = settings->renderResolution();
        const int32_t W = (int32_t)res[0], H = (int32_t)res[1];
        std::vector<GeometryBuffer> gbs;
        std::vector<vrt_shard> shards((size_t)n);
        for (int r = 0; r < n; r++) {                                   // every device traces its strips (asynchronously)
            shards[(size_t)r] = vrt_shard{r, n, stripRows};
            gbs.push_back(stages[(size_t)r]->record(push, n > 1 ? &shards[(size_t)r] : nullptr));
        }
        {                                                               // (one device: a one-rank communicator, the same path)
            const size_t rows = (size_t)vrt_shard_rows(H, &shards[0]);
            if (!full.color || full.width != res[0] || full.height != res[1]) {
                size_t px = (size_t)W * H;
                full.color = std::make_shared<DeviceBuffer<uint8_t>>(engines[0], px * 4); full.normal = std::make_shared<DeviceBuffer<int8_t>>(engines[0], px * 4);
                full.position = std::make_shared<DeviceBuffer<float>>(engines[0], px * 4);
                full.width = res[0]; full.height = res[1];
                packed.clear(); gathered.reset();
                for (int r = 0; r < n; r++) packed.push_back(std::make_shared<DeviceBuffer<uint8_t>>(engines[(size_t)r], rows * (size_t)W * 16));
                gathered = std::make_shared<DeviceBuffer<uint8_t>>(engines[0], (size_t)n * rows * (size_t)W * 16);
            }
            for (int plane = 0; plane < 3; plane++) {
                const int bpp = plane == 2 ? 16 : 4;
                const size_t bytes = rows * (size_t)W * (size_t)bpp;
                for (int r = 0; r < n; r++) {
                    const void* src = plane == 0 ? (const void*)gbs[(size_t)r].color->ptr : (plane == 1 ? (const void*)gbs[(size_t)r].normal->ptr : (const void*)gbs[(size_t)r].position->ptr);
                    check(vrt_pack_rows(engines[(size_t)r]->ctx, src, packed[(size_t)r]->ptr, W, H, bpp, &shards[(size_t)r]));
                }
                check(vrt_group_start());                               // the n ranks of this process: one grouped gather
                for (int r = 0; r < n; r++)
                    check(vrt_gather_strips(engines[(size_t)r]->ctx, comms[(size_t)r], 0, packed[(size_t)r]->ptr, r == 0 ? gathered->ptr : nullptr, bytes));
                check(vrt_group_end());
                void* dst = plane == 0 ? (void*)full.color->ptr : (plane == 1 ? (void*)full.normal->ptr : (void*)full.position->ptr);
                for (int r = 0; r < n; r++)
                    check(vrt_unpack_rows(engines[0]->ctx, gathered->ptr + (size_t)r * bytes, dst, W, H, bpp, &shards[(size_t)r]));
                for (int r = 1; r < n; r++) engines[(size_t)r]->waitIdle();   // the send buffers are reused by the next plane
                engines[0]->waitIdle();
            }
        }
        const uint8_t* img = settings->denoiserSettings.enable ? denoiser->record(full) : full.color->ptr;
        std::vector<uint8_t> host((size_t)W * H * 4);
        check(vrt_memcpy_d2h(engines[0]->ctx, host.data(), img, host.size()));
        if (outW) *outW = res[0];
        if (outH) *outH = res[1];
        return host;
    }
private:
    std::vector<std::shared_ptr<Engine>> engines; std::shared_ptr<VoxelRenderSettings> settings; std::vector<std::shared_ptr<VoxelScene>> scenes;
    int stripRows; std::vector<vrt_comm*> comms; std::vector<std::unique_ptr<GeometryStage>> stages; std::unique_ptr<DenoiserStage> denoiser;
    GeometryBuffer full; std::vector<std::shared_ptr<DeviceBuffer<uint8_t>>> packed; std::shared_ptr<DeviceBuffer<uint8_t>> gathered;
};

} // namespace vrt_host
