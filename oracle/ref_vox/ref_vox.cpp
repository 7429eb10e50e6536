// ref_vox.cpp -- ORACLE-side driver (test infrastructure, NOT product code).
//
// Compiles the reference's own MagicaVoxel parser, thirdparty/opengametools/include/ogt_vox.h,
// from where it lies under /root/reference (never copied into this repo), and restates on top of it
// the scene flatten of source/voxels/resource/voxel_scene.cpp:9-31,53-117 (glm replaced by plain
// arithmetic; all operands are integers or half-integers, so fp32 evaluation order is immaterial).
// Output: oracle/_ref/libvrt_refvox.so (git-ignored, travels to the GPU box prebuilt).
//
// Used (a) to validate the product's own .vox reader (voxel-raytracing_amd/csrc/vox_reader.cpp) and
// (b) by tests/golden/make_vox_fixtures.py to write the .vox fixtures with ogt_vox_write_scene and
// record the expected flattened volumes.
#define OGT_VOX_IMPLEMENTATION
#include "ogt_vox.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct ivec3 { int x, y, z; };

// calc_vox_pivot, voxel_scene.cpp:9-16
ivec3 calc_pivot(const ogt_vox_model* m)
{
    return { (int)std::floor(m->size_x / 2.0f), (int)std::floor(m->size_y / 2.0f), (int)std::floor(m->size_z / 2.0f) };
}

// apply_vox_transform + apply_transform, voxel_scene.cpp:18-31: floor(M * (p + 0.5 - pivot, 1))
ivec3 apply_xform(const ogt_vox_transform& t, const ivec3& pivot, const ivec3& p)
{
    float vx = (float)p.x + 0.5f - (float)pivot.x;
    float vy = (float)p.y + 0.5f - (float)pivot.y;
    float vz = (float)p.z + 0.5f - (float)pivot.z;
    float vw = 1.0f;
    // glm mat4 * vec4 with columns (m00..m03), (m10..m13), (m20..m23), (m30..m33)
    float rx = (t.m00 * vx + t.m10 * vy) + (t.m20 * vz + t.m30 * vw);
    float ry = (t.m01 * vx + t.m11 * vy) + (t.m21 * vz + t.m31 * vw);
    float rz = (t.m02 * vx + t.m12 * vy) + (t.m22 * vz + t.m32 * vw);
    return { (int)std::floor(rx), (int)std::floor(ry), (int)std::floor(rz) };
}

// ogt_vox_sample_instance_transform dereferences scene->groups[instance->group_index]; for a legacy file
// without any nTRN/nGRP/nSHP chunk the parser creates one instance with group_index 0 but NO group
// (ogt_vox.h:1761-1773), so the reference itself crashes there (NULL dereference, voxel_scene.cpp:60).
// Canonical resolution: such an instance keeps its own (identity) transform.
ogt_vox_transform sample_xform(const ogt_vox_instance* inst, const ogt_vox_scene* scene)
{
    if (scene->num_groups == 0) return inst->transform;
    return ogt_vox_sample_instance_transform(inst, 0, scene);
}

} // namespace

extern "C" {

// Returns 0 ok, 1 "Could not parse voxel scene", 2 "Voxel scene does not contain an instance."
// (voxel_scene.cpp:44-50).  *voxels is malloc'd (refvox_free).  palette: 256 x {r,g,b,a,metallic,0,0,0}.
int refvox_flatten(const uint8_t* buf, uint32_t size, uint32_t dims[3], uint8_t** voxels,
                   float* palette, uint32_t* num_instances, uint64_t* dropped)
{
    const ogt_vox_scene* scene = ogt_vox_read_scene(buf, size);
    if (!scene) return 1;
    if (scene->num_instances < 1) { ogt_vox_destroy_scene(scene); return 2; }
    *num_instances = scene->num_instances;

    ivec3 mn = {100000, 100000, 100000}, mx = {-100000, -100000, -100000};      // :53-54
    for (uint32_t i = 0; i < scene->num_instances; i++) {                         // :55-71
        const ogt_vox_instance* inst = &scene->instances[i];
        const ogt_vox_model* model = scene->models[inst->model_index];
        ogt_vox_transform xf = sample_xform(inst, scene);
        ivec3 pivot = calc_pivot(model);
        ivec3 c1 = apply_xform(xf, pivot, {0, 0, 0});
        ivec3 c2 = apply_xform(xf, pivot, {(int)model->size_x, (int)model->size_y, (int)model->size_z});
        mn.x = std::min(mn.x, std::min(c1.x, c2.x)); mn.y = std::min(mn.y, std::min(c1.y, c2.y)); mn.z = std::min(mn.z, std::min(c1.z, c2.z));
        mx.x = std::max(mx.x, std::max(c1.x, c2.x)); mx.y = std::max(mx.y, std::max(c1.y, c2.y)); mx.z = std::max(mx.z, std::max(c1.z, c2.z));
    }
    uint32_t width = (uint32_t)(mx.x - mn.x), height = (uint32_t)(mx.z - mn.z), depth = (uint32_t)(mx.y - mn.y);  // :72-74
    dims[0] = width; dims[1] = height; dims[2] = depth;
    size_t total = (size_t)width * height * depth;
    uint8_t* data = (uint8_t*)calloc(total ? total : 1, 1);
    uint64_t drop = 0;
    for (uint32_t i = 0; i < scene->num_instances; i++) {                         // :81-105
        const ogt_vox_instance* inst = &scene->instances[i];
        const ogt_vox_model* model = scene->models[inst->model_index];
        ivec3 pivot = calc_pivot(model);
        ogt_vox_transform xf = sample_xform(inst, scene);
        for (uint32_t x = 0; x < model->size_x; x++)
            for (uint32_t y = 0; y < model->size_y; y++)
                for (uint32_t z = 0; z < model->size_z; z++) {
                    size_t vp = x + (size_t)y * model->size_x + (size_t)z * model->size_x * model->size_y;
                    uint8_t vox = model->voxel_data[vp];
                    if (vox == 0) continue;
                    ivec3 t = apply_xform(xf, pivot, {(int)x, (int)y, (int)z});
                    long long tx = t.x - mn.x, ty = t.y - mn.y, tz = t.z - mn.z;
                    // scenePos = x + z*width + y*width*height (:99).  The reference performs this store
                    // unchecked; a position outside the allocation is a heap overflow there and is
                    // dropped (counted) here.
                    long long sp = tx + tz * (long long)width + ty * (long long)width * (long long)height;
                    if (sp < 0 || (size_t)sp >= total) { drop++; continue; }
                    data[sp] = vox;
                }
    }
    for (int m = 0; m < 256; m++) {                                               // :108-117
        const ogt_vox_rgba c = scene->palette.color[m];
        float* p = palette + m * 8;
        p[0] = std::pow(c.r / 255.0f, 2.2f); p[1] = std::pow(c.g / 255.0f, 2.2f);
        p[2] = std::pow(c.b / 255.0f, 2.2f); p[3] = std::pow(c.a / 255.0f, 2.2f);
        p[4] = scene->materials.matl[m].metal; p[5] = p[6] = p[7] = 0.0f;
    }
    ogt_vox_destroy_scene(scene);
    *voxels = data;
    if (dropped) *dropped = drop;
    return 0;
}

void refvox_free(void* p) { free(p); }

// Serialise a scene description with the reference's ogt_vox_write_scene.
// palette_rgba: 256*4 in scene order (entry i colours voxel id i).  metal[i] < 0 => no MATL entry.
// Transforms: 16 floats each in ogt_vox_transform member order.  group 0 must be the root
// (parent = UINT32_MAX).  Returns malloc'd buffer (refvox_free).
int refvox_write(uint32_t num_models, const uint32_t* sizes, const uint8_t* const* voxel_data,
                 uint32_t num_groups, const float* group_xforms, const uint32_t* group_parents,
                 uint32_t num_instances, const uint32_t* inst_model, const uint32_t* inst_group,
                 const float* inst_xforms, const uint8_t* inst_hidden,
                 const uint8_t* palette_rgba, const float* metal,
                 uint8_t** out_buf, uint32_t* out_size)
{
    std::vector<ogt_vox_model> models(num_models);
    std::vector<const ogt_vox_model*> model_ptrs(num_models);
    for (uint32_t i = 0; i < num_models; i++) {
        models[i].size_x = sizes[i * 3 + 0]; models[i].size_y = sizes[i * 3 + 1]; models[i].size_z = sizes[i * 3 + 2];
        models[i].voxel_hash = 0; models[i].voxel_data = voxel_data[i];
        model_ptrs[i] = &models[i];
    }
    std::vector<ogt_vox_group> groups(num_groups);
    for (uint32_t i = 0; i < num_groups; i++) {
        memset(&groups[i], 0, sizeof(ogt_vox_group));
        memcpy(&groups[i].transform, group_xforms + i * 16, sizeof(float) * 16);
        groups[i].parent_group_index = group_parents[i];
    }
    std::vector<ogt_vox_instance> insts(num_instances);
    for (uint32_t i = 0; i < num_instances; i++) {
        memset(&insts[i], 0, sizeof(ogt_vox_instance));
        memcpy(&insts[i].transform, inst_xforms + i * 16, sizeof(float) * 16);
        insts[i].model_index = inst_model[i];
        insts[i].group_index = inst_group[i];
        insts[i].hidden = inst_hidden ? inst_hidden[i] != 0 : false;
    }
    ogt_vox_layer layer; memset(&layer, 0, sizeof layer);
    ogt_vox_scene scene; memset(&scene, 0, sizeof scene);
    scene.num_models = num_models; scene.models = model_ptrs.data();
    scene.num_instances = num_instances; scene.instances = insts.data();
    scene.num_layers = 1; scene.layers = &layer;
    scene.num_groups = num_groups; scene.groups = groups.data();
    memcpy(&scene.palette, palette_rgba, 1024);
    for (int i = 0; i < 256; i++) {
        if (metal && metal[i] >= 0.0f) {
            scene.materials.matl[i].content_flags = k_ogt_vox_matl_have_metal;
            scene.materials.matl[i].type = ogt_matl_type_metal;
            scene.materials.matl[i].metal = metal[i];
        }
    }
    uint32_t n = 0;
    uint8_t* b = ogt_vox_write_scene(&scene, &n);
    if (!b) return 1;
    uint8_t* copy = (uint8_t*)malloc(n);
    memcpy(copy, b, n);
    ogt_vox_free(b);
    *out_buf = copy; *out_size = n;
    return 0;
}

} // extern "C"
