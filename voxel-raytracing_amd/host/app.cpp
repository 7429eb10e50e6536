// app.cpp -- offline counterpart of the reference's App (source/app.cpp:8-27, run/main.cpp:3): load a scene,
// set camera + settings, render through the C++ host mirror (voxels.hpp) and write the RGBA image.
// Errors propagate as exceptions to run(), are printed, and the process exits with EXIT_FAILURE.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "voxels.hpp"

using namespace vrt_host;

namespace {

std::vector<uint8_t> readFile(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::streamsize n = f.tellg(); f.seekg(0);
    std::vector<uint8_t> b((size_t)n);
    if (!f.read((char*)b.data(), n)) throw std::runtime_error("cannot read " + path);
    return b;
}

// "VRTD" dense-scene container written by the Python harness: dims, voxels, palette, sky, noise
std::shared_ptr<VoxelScene> loadDense(const std::shared_ptr<Engine>& engine, const std::string& path)
{
    std::vector<uint8_t> b = readFile(path);
    size_t o = 0;
    auto u32 = [&]() { uint32_t v; if (o + 4 > b.size()) throw std::runtime_error("truncated dense scene"); memcpy(&v, &b[o], 4); o += 4; return v; };
    if (b.size() < 4 || memcmp(b.data(), "VRTD", 4) != 0) throw std::runtime_error("not a VRTD file");
    o = 4;
    uint32_t W = u32(), H = u32(), D = u32();
    size_t nv = (size_t)W * H * D;
    if (o + nv + 256 * sizeof(vrt_material) > b.size()) throw std::runtime_error("truncated dense scene");
    const uint8_t* vox = &b[o]; o += nv;
    std::vector<vrt_material> pal(256); memcpy(pal.data(), &b[o], 256 * sizeof(vrt_material)); o += 256 * sizeof(vrt_material);
    auto scene = std::make_shared<VoxelScene>(engine, vox, W, H, D, pal.data());
    if (o < b.size()) { uint32_t w = u32(), h = u32(); std::vector<float> sky((size_t)w * h * 4); memcpy(sky.data(), &b[o], sky.size() * 4); o += sky.size() * 4; scene->setSkybox(sky.data(), w, h); }
    if (o < b.size()) { uint32_t w = u32(), h = u32(); scene->setBlueNoise(&b[o], w, h); o += (size_t)w * h * 4; }
    return scene;
}

int run(int argc, char** argv)
{
    try {
        std::string vox, dense, out, raw, dumpPush, sky, noise, png;
        auto settings = std::make_shared<VoxelRenderSettings>();
        vec3 pos{8, 8, -50}; float yaw = 90, pitch = 0; bool havePos = false; int device = 0;
        std::vector<int> devices;                                          // --devices a,b,...: one process drives several GPUs
        int frames = 1; uint32_t windowW = 0, windowH = 0; float flyForward = 0, flyStrafe = 0, flyMouseX = 0; bool temporal = false;
        for (int i = 1; i < argc; i++) {
            std::string a = argv[i];
            auto next = [&]() { if (i + 1 >= argc) throw std::runtime_error("missing value for " + a); return std::string(argv[++i]); };
            if (a == "--vox") vox = next(); else if (a == "--dense") dense = next(); else if (a == "--out") out = next();
            else if (a == "--raw") raw = next(); else if (a == "--dump-push") dumpPush = next();
            else if (a == "--sky") sky = next(); else if (a == "--noise") noise = next(); else if (a == "--png") png = next();
            else if (a == "--width") settings->targetResolution[0] = (uint32_t)std::stoul(next());
            else if (a == "--height") settings->targetResolution[1] = (uint32_t)std::stoul(next());
            else if (a == "--pos") { pos.x = std::stof(next()); pos.y = std::stof(next()); pos.z = std::stof(next()); havePos = true; }
            else if (a == "--yaw") yaw = std::stof(next()); else if (a == "--pitch") pitch = std::stof(next());
            else if (a == "--no-fsr") settings->fsrSetttings.enable = false;
            else if (a == "--no-denoise") settings->denoiserSettings.enable = false;
            else if (a == "--iterations") settings->denoiserSettings.iterations = std::stoi(next());
            else if (a == "--ao") settings->occlusionSettings.numSamples = std::stoi(next());
            else if (a == "--bounces") settings->traceSettings.maxReflections = (uint32_t)std::stoul(next());
            else if (a == "--no-shadows") settings->traceSettings.shadows = false;
            else if (a == "--primary-only") { settings->occlusionSettings.numSamples = 0; settings->traceSettings.shadows = false; settings->traceSettings.maxReflections = 0; }
            else if (a == "--device") device = std::stoi(next());
            else if (a == "--devices") { std::string l = next(); size_t p0 = 0; while (p0 <= l.size()) { size_t c = l.find(',', p0); if (c == std::string::npos) c = l.size(); if (c > p0) devices.push_back(std::stoi(l.substr(p0, c - p0))); p0 = c + 1; } }
            else if (a == "--frames") frames = std::max(1, std::stoi(next()));                 // frames to run (update + render each)
            else if (a == "--temporal") temporal = true;                                       // accumulate jittered frames + upscale
            else if (a == "--window") { windowW = (uint32_t)std::stoul(next()); windowH = (uint32_t)std::stoul(next()); }
            else if (a == "--fly") { flyForward = std::stof(next()); flyStrafe = std::stof(next()); flyMouseX = std::stof(next()); }
            else throw std::runtime_error("unknown argument " + a);
        }
        if (!devices.empty()) {
            // screen strips over the listed GPUs, RCCL gather to the first one (ShardedRenderer); the scene is loaded on each
            std::vector<std::shared_ptr<Engine>> engines; std::vector<std::shared_ptr<VoxelScene>> scenes;
            for (int d : devices) {
                auto e = std::make_shared<Engine>(d);
                std::shared_ptr<VoxelScene> sc = !dense.empty() ? loadDense(e, dense) : std::make_shared<VoxelScene>(e, vox.empty() ? settings->voxPath : vox);
                if (!sky.empty()) sc->setSkybox(sky);
                if (!noise.empty()) sc->setBlueNoise(noise);
                engines.push_back(e); scenes.push_back(sc);
            }
            VoxelRenderer cameraOnly(engines[0], settings, scenes[0]);                         // push constants as the single-GPU path builds them
            if (!havePos) pos = {scenes[0]->width / 2.0f, scenes[0]->height / 2.0f, -0.8f * scenes[0]->depth};
            cameraOnly.camera().position = pos; cameraOnly.camera().yaw = yaw; cameraOnly.camera().pitch = pitch;
            cameraOnly.camera().updateDirectionVectors();
            ShardedRenderer sharded(engines, settings, scenes);
            uint32_t res[2] = {0, 0};
            std::vector<uint8_t> img = sharded.render(cameraOnly.pushConstants(), &res[0], &res[1]);
            if (!raw.empty()) std::ofstream(raw, std::ios::binary).write((const char*)img.data(), (std::streamsize)img.size());
            if (!png.empty()) check(vrt_image_write_png(png.c_str(), img.data(), res[0], res[1]));
            std::printf("rendered %ux%u on %zu device(s), RCCL gather\n", res[0], res[1], devices.size());
            return EXIT_SUCCESS;
        }
        auto engine = std::make_shared<Engine>(device);
        std::shared_ptr<VoxelScene> scene;
        if (!dense.empty()) scene = loadDense(engine, dense);
        else scene = std::make_shared<VoxelScene>(engine, vox.empty() ? settings->voxPath : vox);
        if (!sky.empty()) scene->setSkybox(sky);
        if (!noise.empty()) scene->setBlueNoise(noise);
        VoxelRenderer renderer(engine, settings, scene);
        if (!havePos) pos = {scene->width / 2.0f, scene->height / 2.0f, -0.8f * scene->depth};
        renderer.camera().position = pos; renderer.camera().yaw = yaw; renderer.camera().pitch = pitch;
        renderer.camera().updateDirectionVectors();
        renderer.temporal = temporal; renderer.windowW = windowW; renderer.windowH = windowH;
        std::vector<uint8_t> img; uint32_t res[2] = {0, 0};
        const bool moving = flyForward != 0 || flyStrafe != 0 || flyMouseX != 0;
        for (int f = 0; f < frames; f++) {                                                     // App::run loop (source/app.cpp:18-27)
            if (frames > 1 || temporal) {
                if (flyMouseX != 0) renderer.camera().mouse(flyMouseX, 0.0f);
                renderer.update(1.0f / 60.0f, flyForward, flyStrafe);
                if (moving) renderer.upscaler().reset();                                       // no reprojection: history is per pose
            }
            img = renderer.render(&res[0], &res[1]);
        }
        if (!dumpPush.empty()) { vrt_push p = renderer.pushConstants(); std::ofstream(dumpPush, std::ios::binary).write((const char*)&p, sizeof p); }
        if (!raw.empty()) std::ofstream(raw, std::ios::binary).write((const char*)img.data(), (std::streamsize)img.size());
        if (!png.empty()) check(vrt_image_write_png(png.c_str(), img.data(), res[0], res[1]));
        if (!out.empty()) {
            std::ofstream f(out, std::ios::binary);
            f << "P6\n" << res[0] << " " << res[1] << "\n255\n";
            for (size_t i = 0; i < img.size(); i += 4) f.write((const char*)&img[i], 3);
        }
        std::printf("rendered %ux%u scene %ux%ux%u\n", res[0], res[1], scene->width, scene->height, scene->depth);
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}

} // namespace

int main(int argc, char** argv) { return run(argc, argv); }
