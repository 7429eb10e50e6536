// vox_reader.cpp -- MagicaVoxel .vox ingestion: chunk reader + scene-graph flatten + dense-volume build.
//
// Host-side counterpart of the reference's scene load:
//   thirdparty/opengametools/include/ogt_vox.h:1179-1991  (chunk loop, instance generation, IMAP remap,
//                                                          palette rotate), :861-907 (_r/_t decode)
//   source/voxels/resource/voxel_scene.cpp:9-31,53-117      (pivot, bbox, Y/Z swap scatter, palette)
// Written from the published .vox format; validated chunk-for-chunk against the reference's ogt_vox.h
// through oracle/_ref (tests/test_vox_reader.py) -- it shares no code with it.
//
// Behaviours reproduced on purpose (they decide what the renderer sees):
//   * only frame 0 of nTRN / nSHP; hidden flags and layers are ignored (voxel_scene.cpp never tests them);
//   * instances are visited depth-first in file child order, later instances overwrite earlier ones;
//   * an IMAP chunk remaps EVERY cell of every model, empty ones included (ogt_vox.h:1817-1826);
//   * scene = x + z*width + y*width*height, width = dx, height = dz, depth = dy (voxel_scene.cpp:72-74,99);
//     a store that would land outside the allocation (possible with mirrored rotations, where the
//     reference overflows its heap buffer) is dropped and counted.
#include "vox_reader.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace vrt {
namespace {

struct Reader {
    const uint8_t* p; size_t n, off;
    bool ok(size_t k) const { return off + k <= n; }
    bool u32(uint32_t& v) { if (!ok(4)) return false; memcpy(&v, p + off, 4); off += 4; return true; }
    bool i32(int32_t& v) { uint32_t u; if (!u32(u)) return false; v = (int32_t)u; return true; }
    bool f32(float& v) { if (!ok(4)) return false; memcpy(&v, p + off, 4); off += 4; return true; }
    bool str(std::string& s) { uint32_t k; if (!u32(k) || !ok(k)) return false; s.assign((const char*)p + off, k); off += k; return true; }
};

struct Dict {
    std::vector<std::pair<std::string, std::string>> kv;
    const std::string* get(const char* key) const {
        for (auto& e : kv) if (strcasecmp(e.first.c_str(), key) == 0) return &e.second;
        return nullptr;
    }
};

bool read_dict(Reader& r, Dict& d)
{
    d.kv.clear();
    uint32_t n;
    if (!r.u32(n) || n > 256) return false;
    for (uint32_t i = 0; i < n; i++) {
        std::string k, v;
        if (!r.str(k) || !r.str(v)) return false;
        d.kv.emplace_back(std::move(k), std::move(v));
    }
    return true;
}

// out = R * v + t with R rows r[0..2]
struct Xform { int r[3][3]; int t[3]; };

Xform identity() { Xform x = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, {0, 0, 0}}; return x; }

// apply child first, then parent
Xform compose(const Xform& child, const Xform& parent)
{
    Xform o;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            int s = 0;
            for (int k = 0; k < 3; k++) s += parent.r[i][k] * child.r[k][j];
            o.r[i][j] = s;
        }
        int s = parent.t[i];
        for (int k = 0; k < 3; k++) s += parent.r[i][k] * child.t[k];
        o.t[i] = s;
    }
    return o;
}

// _r : packed rotation byte (bits 0-1 / 2-3: column of the non-zero entry in rows 0 / 1; bits 4-6: row signs)
// _t : "x y z"
bool decode_frame(const Dict& d, Xform& x)
{
    x = identity();
    if (const std::string* rs = d.get("_r")) {
        uint32_t bits = (uint32_t)atoi(rs->c_str());
        uint32_t i0 = bits & 3, i1 = (bits >> 2) & 3;
        if (i0 > 2 || i1 > 2 || i0 == i1) return false;
        uint32_t i2 = 3 - i0 - i1;
        memset(x.r, 0, sizeof x.r);
        x.r[0][i0] = (bits & 16) ? -1 : 1;
        x.r[1][i1] = (bits & 32) ? -1 : 1;
        x.r[2][i2] = (bits & 64) ? -1 : 1;
    }
    if (const std::string* ts = d.get("_t")) {
        const char* c = ts->c_str(); char* e;
        for (int k = 0; k < 3; k++) { x.t[k] = (int)strtol(c, &e, 0); if (e == c) break; c = e; }
    }
    return true;
}

struct Model { uint32_t sx = 0, sy = 0, sz = 0; std::vector<uint8_t> data; bool present = false; };

struct Node {
    enum Type { None, Trn, Grp, Shp } type = None;
    Xform xf;                     // Trn: first frame
    uint32_t child = 0;           // Trn
    std::vector<uint32_t> kids;   // Grp
    uint32_t model = 0;           // Shp: first model
};

struct Instance { uint32_t model; Xform world; };

bool walk(const std::vector<Node>& nodes, const std::vector<Model>& models, uint32_t id, const Xform& parent_world,
          const Xform* last_trn, std::vector<Instance>& out, int depth)
{
    if (depth > 512 || id >= nodes.size()) return false;
    const Node& nd = nodes[id];
    switch (nd.type) {
    case Node::Trn:
        return walk(nodes, models, nd.child, parent_world, &nd.xf, out, depth + 1);
    case Node::Grp: {
        if (!last_trn) return false;
        Xform gw = compose(*last_trn, parent_world);
        for (uint32_t k : nd.kids)
            if (!walk(nodes, models, k, gw, nullptr, out, depth + 1)) return false;
        return true;
    }
    case Node::Shp:
        if (!last_trn) return false;
        if (nd.model < models.size() && models[nd.model].present)
            out.push_back({nd.model, compose(*last_trn, parent_world)});
        return true;
    default:
        return false;
    }
}

inline uint32_t chunk_id(const char* s) { return (uint32_t)s[0] | (uint32_t)s[1] << 8 | (uint32_t)s[2] << 16 | (uint32_t)s[3] << 24; }

// The palette MagicaVoxel assumes when a file carries no RGBA chunk (file-format spec): the 6x6x6
// colour cube minus black, four 10-step ramps, black.
void default_palette(uint8_t pal[256][4])
{
    static const uint8_t lv[6] = {0xff, 0xcc, 0x99, 0x66, 0x33, 0x00};
    static const uint8_t ramp[10] = {0xee, 0xdd, 0xbb, 0xaa, 0x88, 0x77, 0x55, 0x44, 0x22, 0x11};
    for (int i = 0; i < 215; i++) { pal[i][0] = lv[i / 36]; pal[i][1] = lv[(i / 6) % 6]; pal[i][2] = lv[i % 6]; pal[i][3] = 0xff; }
    for (int k = 0; k < 10; k++) {
        uint8_t v = ramp[k];
        uint8_t* r = pal[215 + k]; r[0] = v; r[1] = 0; r[2] = 0; r[3] = 0xff;
        uint8_t* g = pal[225 + k]; g[0] = 0; g[1] = v; g[2] = 0; g[3] = 0xff;
        uint8_t* b = pal[235 + k]; b[0] = 0; b[1] = 0; b[2] = v; b[3] = 0xff;
        uint8_t* y = pal[245 + k]; y[0] = v; y[1] = v; y[2] = v; y[3] = 0xff;
    }
    pal[255][0] = pal[255][1] = pal[255][2] = 0; pal[255][3] = 0xff;
}

inline int floor_half(long long twice) { return (int)(twice >> 1); }   // floor(twice / 2), arithmetic shift

// floor(M * (p + 0.5 - pivot)) in exact integer arithmetic on doubled coordinates (voxel_scene.cpp:18-21)
void xform_point(const Xform& w, const int pivot[3], const int p[3], int out[3])
{
    long long v2[3] = {2LL * p[0] + 1 - 2LL * pivot[0], 2LL * p[1] + 1 - 2LL * pivot[1], 2LL * p[2] + 1 - 2LL * pivot[2]};
    for (int i = 0; i < 3; i++) {
        long long s = 2LL * w.t[i];
        for (int k = 0; k < 3; k++) s += (long long)w.r[i][k] * v2[k];
        out[i] = floor_half(s);
    }
}

} // namespace

int vox_flatten(const uint8_t* buf, size_t n, FlatScene& fs, std::string& err)
{
    Reader r{buf, n, 0};
    uint32_t magic = 0, version = 0;
    if (!r.u32(magic) || !r.u32(version) || magic != chunk_id("VOX ") || (version != 150 && version != 200)) {
        err = "Could not parse voxel scene"; return VRT_ERR_PARSE;
    }
    std::vector<Model> models;
    std::vector<Node> nodes;
    uint8_t pal[256][4];
    default_palette(pal);
    float metal[256] = {0};
    uint8_t imap[256]; bool have_imap = false;
    uint32_t sx = 0, sy = 0, sz = 0;
    Dict dict;

    while (r.n - r.off >= 12) {
        uint32_t id, size, child_size;
        r.u32(id); r.u32(size); r.u32(child_size);
        size_t body = r.off;
        if (id == chunk_id("MAIN")) continue;                 // children follow inline
        if (!r.ok(size)) break;                               // truncated file: stop like a short read would
        size_t end = body + size;
        bool good = true;
        if (id == chunk_id("SIZE")) {
            good = r.u32(sx) && r.u32(sy) && r.u32(sz);
        } else if (id == chunk_id("XYZI")) {
            uint32_t count = 0;
            good = r.u32(count);
            Model m;
            if (good && count != 0) {
                if (!sx || !sy || !sz || (uint64_t)sx * sy * sz > (1ull << 31)) { err = "Could not parse voxel scene"; return VRT_ERR_PARSE; }
                m.sx = sx; m.sy = sy; m.sz = sz; m.present = true;
                m.data.assign((size_t)sx * sy * sz, 0);
                size_t avail = (r.n - r.off) / 4;
                size_t todo = count < avail ? count : avail;
                const uint8_t* v = r.p + r.off;
                for (size_t i = 0; i < todo; i++) {
                    uint32_t x = v[i * 4], y = v[i * 4 + 1], z = v[i * 4 + 2];
                    if (x < sx && y < sy && z < sz) m.data[x + (size_t)y * sx + (size_t)z * sx * sy] = v[i * 4 + 3];
                }
            }
            models.push_back(std::move(m));
        } else if (id == chunk_id("RGBA")) {
            if (size >= 1024) memcpy(pal, r.p + r.off, 1024); else good = false;
        } else if (id == chunk_id("nTRN")) {
            uint32_t node_id = 0, child = 0, reserved, layer, frames = 0;
            good = r.u32(node_id) && read_dict(r, dict) && r.u32(child) && r.u32(reserved) && r.u32(layer) && r.u32(frames);
            Xform first = identity();
            for (uint32_t f = 0; good && f < frames; f++) {
                good = read_dict(r, dict);
                if (good && f == 0) good = decode_frame(dict, first);
            }
            if (good && frames > 0 && node_id < (1u << 24)) {
                if (node_id >= nodes.size()) nodes.resize(node_id + 1);
                Node& nd = nodes[node_id]; nd.type = Node::Trn; nd.xf = first; nd.child = child;
            } else good = false;
        } else if (id == chunk_id("nGRP")) {
            uint32_t node_id = 0, nkids = 0;
            good = r.u32(node_id) && read_dict(r, dict) && r.u32(nkids) && r.ok((size_t)nkids * 4) && node_id < (1u << 24);
            if (good) {
                if (node_id >= nodes.size()) nodes.resize(node_id + 1);
                Node& nd = nodes[node_id]; nd.type = Node::Grp; nd.kids.resize(nkids);
                for (uint32_t k = 0; k < nkids; k++) r.u32(nd.kids[k]);
            }
        } else if (id == chunk_id("nSHP")) {
            uint32_t node_id = 0, nmodels = 0, first_model = 0;
            good = r.u32(node_id) && read_dict(r, dict) && r.u32(nmodels) && nmodels > 0 && node_id < (1u << 24);
            for (uint32_t k = 0; good && k < nmodels; k++) {
                uint32_t mid; good = r.u32(mid) && read_dict(r, dict);
                if (k == 0) first_model = mid;
            }
            if (good) {
                if (node_id >= nodes.size()) nodes.resize(node_id + 1);
                Node& nd = nodes[node_id]; nd.type = Node::Shp; nd.model = first_model;
            }
        } else if (id == chunk_id("IMAP")) {
            if (size >= 256) { memcpy(imap, r.p + r.off, 256); have_imap = true; } else good = false;
        } else if (id == chunk_id("MATL")) {
            int32_t mid = 0;
            good = r.i32(mid) && read_dict(r, dict);
            if (good) if (const std::string* ms = dict.get("_metal")) metal[mid & 0xFF] = (float)atof(ms->c_str());
        } else if (id == chunk_id("MATT")) {
            int32_t mid = 0, type = 0; float weight = 0.0f;
            good = r.i32(mid) && r.i32(type) && r.f32(weight);
            if (good && type == 1) metal[mid & 0xFF] = weight;
        }
        if (!good) { err = "Could not parse voxel scene"; return VRT_ERR_PARSE; }
        r.off = end;                                           // every chunk is skipped by its declared size
    }

    // instances (ogt_vox.h:1675-1759)
    std::vector<Instance> inst;
    if (!nodes.empty()) {
        if (!walk(nodes, models, 0, identity(), nullptr, inst, 0)) { err = "Could not parse voxel scene"; return VRT_ERR_PARSE; }
    } else if (models.size() == 1 && models[0].present) {
        inst.push_back({0, identity()});
    }

    // IMAP (ogt_vox.h:1793-1827)
    if (have_imap) {
        uint8_t inv[256];
        for (int i = 0; i < 256; i++) inv[imap[i]] = (uint8_t)i;
        uint8_t oldp[256][4]; memcpy(oldp, pal, sizeof pal);
        for (int i = 0; i < 256; i++) memcpy(pal[i], oldp[(imap[i] + 255) & 0xFF], 4);
        float oldm[256]; memcpy(oldm, metal, sizeof metal);
        for (int i = 0; i < 256; i++) metal[i] = oldm[imap[(i + 255) & 0xFF]];
        for (Model& m : models) for (uint8_t& v : m.data) v = (uint8_t)(1 + inv[v]);
    }
    // palette rotate (ogt_vox.h:1834-1840): entry i colours voxel id i; id 0 is the empty voxel
    {
        uint8_t last[4]; memcpy(last, pal[255], 4);
        for (int i = 255; i > 0; i--) memcpy(pal[i], pal[i - 1], 4);
        memcpy(pal[0], last, 4); pal[0][3] = 0;
    }

    if (inst.empty()) { err = "Voxel scene does not contain an instance."; return VRT_ERR_NO_INSTANCE; }
    fs.num_instances = (uint32_t)inst.size();

    // bounding box over the two transformed corners of every instance (voxel_scene.cpp:53-71)
    int mn[3] = {100000, 100000, 100000}, mx[3] = {-100000, -100000, -100000};
    for (const Instance& in : inst) {
        const Model& m = models[in.model];
        int pivot[3] = {(int)(m.sx / 2), (int)(m.sy / 2), (int)(m.sz / 2)};
        int c0[3] = {0, 0, 0}, c1[3] = {(int)m.sx, (int)m.sy, (int)m.sz}, a[3], b[3];
        xform_point(in.world, pivot, c0, a);
        xform_point(in.world, pivot, c1, b);
        for (int k = 0; k < 3; k++) {
            int lo = a[k] < b[k] ? a[k] : b[k], hi = a[k] < b[k] ? b[k] : a[k];
            if (lo < mn[k]) mn[k] = lo;
            if (hi > mx[k]) mx[k] = hi;
        }
    }
    long long width = (long long)mx[0] - mn[0], height = (long long)mx[2] - mn[2], depth = (long long)mx[1] - mn[1];
    if (width <= 0 || height <= 0 || depth <= 0 || width > 4096 || height > 4096 || depth > 4096) {
        err = "Voxel scene has an unsupported extent"; return VRT_ERR_UNSUPPORTED;
    }
    fs.dims[0] = (uint32_t)width; fs.dims[1] = (uint32_t)height; fs.dims[2] = (uint32_t)depth;
    size_t total = (size_t)width * (size_t)height * (size_t)depth;
    fs.voxels.assign(total, 0);
    fs.dropped = 0;
    for (const Instance& in : inst) {                          // voxel_scene.cpp:81-105
        const Model& m = models[in.model];
        int pivot[3] = {(int)(m.sx / 2), (int)(m.sy / 2), (int)(m.sz / 2)};
        for (uint32_t x = 0; x < m.sx; x++)
            for (uint32_t y = 0; y < m.sy; y++)
                for (uint32_t z = 0; z < m.sz; z++) {
                    uint8_t v = m.data[x + (size_t)y * m.sx + (size_t)z * m.sx * m.sy];
                    if (!v) continue;
                    int p[3] = {(int)x, (int)y, (int)z}, t[3];
                    xform_point(in.world, pivot, p, t);
                    long long tx = (long long)t[0] - mn[0], ty = (long long)t[1] - mn[1], tz = (long long)t[2] - mn[2];
                    long long sp = tx + tz * width + ty * width * height;
                    if (sp < 0 || (size_t)sp >= total) { fs.dropped++; continue; }
                    fs.voxels[(size_t)sp] = v;
                }
    }
    for (int i = 0; i < 256; i++) {                            // voxel_scene.cpp:108-117
        vrt_material& mt = fs.palette[i];
        for (int k = 0; k < 4; k++) mt.diffuse[k] = powf((float)pal[i][k] / 255.0f, 2.2f);
        mt.metallic = metal[i];
        mt.pad[0] = mt.pad[1] = mt.pad[2] = 0.0f;
    }
    return VRT_OK;
}

} // namespace vrt
