// vox_reader.h -- host-side .vox ingestion (see vox_reader.cpp).
#pragma once

#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/vrt.h"

namespace vrt {

struct FlatScene {
    uint32_t dims[3] = {0, 0, 0};        // width, height, depth (texture x, y, z)
    std::vector<uint8_t> voxels;         // x + y*W + z*W*H
    vrt_material palette[256];
    uint32_t num_instances = 0;
    uint64_t dropped = 0;                // stores the reference would have made outside its allocation
};

// Returns VRT_OK / VRT_ERR_PARSE / VRT_ERR_NO_INSTANCE / VRT_ERR_UNSUPPORTED; err gets the message.
int vox_flatten(const uint8_t* buf, size_t n, FlatScene& out, std::string& err);

} // namespace vrt
