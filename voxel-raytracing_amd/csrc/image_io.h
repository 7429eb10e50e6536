// image_io.h -- asset decoders / writers (see image_io.cpp).
#pragma once

#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/vrt.h"

namespace vrt {

struct LoadedImage {
    bool is_hdr = false;
    uint32_t w = 0, h = 0;
    std::vector<float> f32;      // is_hdr: linear RGBA
    std::vector<uint8_t> u8;     // else: RGBA8
};

int image_load(const char* path, LoadedImage& img, std::string& err);
int image_write_png(const char* path, const uint8_t* rgba8, uint32_t w, uint32_t h, std::string& err);
int image_write_ppm(const char* path, const uint8_t* rgba8, uint32_t w, uint32_t h, std::string& err);
int image_write_pfm(const char* path, const float* rgb, uint32_t w, uint32_t h, uint32_t stride_floats, std::string& err);

} // namespace vrt
