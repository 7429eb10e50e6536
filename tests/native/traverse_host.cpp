// traverse_host.cpp -- TEST HELPER: compiles the product's traversal header (csrc/vrt_traverse.h) for the
// host so that its three strategies can be compared with the oracle on millions of rays without a GPU.
// Built on demand by tests/test_traverse_host.py with g++; never linked into libvrt_hip.so.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../voxel-raytracing_amd/csrc/vrt_traverse.h"

using namespace vrt;

struct HostVolume {
    VolumeView v;
    std::vector<uint8_t> vox;
    std::vector<uint64_t> o1, o2, o3;
    std::vector<uint8_t> df;
};

static void build_up(const std::vector<uint64_t>& lo, int lx, int ly, int lz, std::vector<uint64_t>& hi, int hx, int hy, int hz)
{
    hi.assign((size_t)hx * hy * hz, 0);
    for (int z = 0; z < lz; z++) for (int y = 0; y < ly; y++) for (int x = 0; x < lx; x++)
        if (lo[(size_t)x + ((size_t)y + (size_t)z * ly) * lx])
            hi[(size_t)(x >> 2) + ((size_t)(y >> 2) + (size_t)(z >> 2) * hy) * hx] |= 1ull << cell_bit(x, y, z);
}

extern "C" {

void* th_create(const uint8_t* vox, int W, int H, int D)
{
    HostVolume* h = new HostVolume();
    h->vox.assign(vox, vox + (size_t)W * H * D);
    VolumeView& v = h->v;
    v.W = W; v.H = H; v.D = D;
    v.n1x = (W + 3) / 4; v.n1y = (H + 3) / 4; v.n1z = (D + 3) / 4;
    v.n2x = (v.n1x + 3) / 4; v.n2y = (v.n1y + 3) / 4; v.n2z = (v.n1z + 3) / 4;
    v.n3x = (v.n2x + 3) / 4; v.n3y = (v.n2y + 3) / 4; v.n3z = (v.n2z + 3) / 4;
    h->o1.assign((size_t)v.n1x * v.n1y * v.n1z, 0);
    for (int z = 0; z < D; z++) for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
        if (h->vox[(size_t)x + ((size_t)y + (size_t)z * H) * W])
            h->o1[(size_t)(x >> 2) + ((size_t)(y >> 2) + (size_t)(z >> 2) * v.n1y) * v.n1x] |= 1ull << cell_bit(x, y, z);
    build_up(h->o1, v.n1x, v.n1y, v.n1z, h->o2, v.n2x, v.n2y, v.n2z);
    build_up(h->o2, v.n2x, v.n2y, v.n2z, h->o3, v.n3x, v.n3y, v.n3z);
    // clearance fields: per octant three one-sided 1-D min-max passes (same definition as k_df_pass, written independently)
    {
        const int CAP = 127;
        size_t stride = df_field_bytes(W, H, D);                 // padded x-fastest fields (vrt_traverse.h df_index)
        h->df.assign(8 * stride, 0);
        std::vector<uint8_t> a((size_t)W * H * D), b((size_t)W * H * D);
        for (int o = 0; o < 8; o++) {
            int sg[3] = {(o & 1) ? 1 : -1, (o & 2) ? 1 : -1, (o & 4) ? 1 : -1};
            for (size_t i = 0; i < a.size(); i++) a[i] = h->vox[i] ? 0 : CAP + 1;
            for (int axis = 0; axis < 3; axis++) {
                for (int z = 0; z < D; z++) for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
                    size_t i = (size_t)x + ((size_t)y + (size_t)z * H) * W;
                    int pos = axis == 0 ? x : (axis == 1 ? y : z), dim = axis == 0 ? W : (axis == 1 ? H : D);
                    long long st = (axis == 0 ? 1 : (axis == 1 ? (long long)W : (long long)W * H)) * sg[axis];
                    int best = a[i];
                    for (int t = 1; t < best; t++) {
                        int q = pos + t * sg[axis];
                        int val = (q < 0 || q >= dim) ? 0 : a[(long long)i + t * st];
                        int m = val > t ? val : t;
                        if (m < best) best = m;
                    }
                    b[i] = (uint8_t)best;
                }
                a.swap(b);
            }
            for (int z = 0; z < D; z++) for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
                size_t i = (size_t)x + ((size_t)y + (size_t)z * H) * W;
                h->df[(size_t)o * stride + df_index(v, x, y, z)] = a[i] > CAP ? CAP : a[i];
            }
        }
        v.df_stride = stride;
    }
    v.df = h->df.data();
    v.vox = h->vox.data(); v.occ1 = h->o1.data(); v.occ2 = h->o2.data(); v.occ3 = h->o3.data();
    return h;
}

void th_destroy(void* p) { delete (HostVolume*)p; }

// out per ray: 12 uint32: material, mask, mx,my,mz, side bits x3, p bits x3, fetches ; stats: 6 uint32 summed
void th_trace(void* p, int trav, int n, const float* starts, const float* dirs, uint32_t maxSteps, uint32_t* out, uint64_t* stats)
{
    HostVolume* h = (HostVolume*)p;
    TraceStats total;
    for (int i = 0; i < n; i++) {
        f3 s = mk3(starts[i * 3], starts[i * 3 + 1], starts[i * 3 + 2]);
        f3 d = mk3(dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2]);
        RayInt r;
        if (trav == VRT_TRAVERSAL_JUMP) {
            TraceStats st;
            trace_jump(h->v, h->v.occ2, h->v.occ3, s, d, maxSteps, r, st);
            total.literal += st.literal; total.jumps1 += st.jumps1; total.jumps2 += st.jumps2; total.jumps3 += st.jumps3;
            total.retrace += st.retrace; total.lookups += st.lookups;
        } else if (trav == VRT_TRAVERSAL_BITMASK) {
            trace_literal<VRT_TRAVERSAL_BITMASK>(h->v, h->v.occ2, s, d, maxSteps, r);
        } else if (trav == VRT_TRAVERSAL_DFJ) {
            TraceStats st;
            trace_dfj(h->v, s, d, maxSteps, r, st);
            total.jumps1 += st.jumps1; total.jumps3 += st.jumps3; total.lookups += st.lookups; total.retrace += st.retrace;
        } else if (trav == VRT_TRAVERSAL_DF) {
            TraceStats st;
            trace_df(h->v, s, d, maxSteps, r, st);
            total.jumps1 += st.jumps1; total.jumps2 += st.jumps2; total.lookups += st.lookups;
        } else {
            trace_literal<VRT_TRAVERSAL_DENSE>(h->v, h->v.occ2, s, d, maxSteps, r);
        }
        uint32_t* o = out + (size_t)i * 12;
        o[0] = r.material; o[1] = r.material ? r.mask : 0; o[2] = r.material ? (uint32_t)r.mx : 0; o[3] = r.material ? (uint32_t)r.my : 0;
        o[4] = r.material ? (uint32_t)r.mz : 0;
        o[5] = r.material ? f2u(r.side.x) : 0; o[6] = r.material ? f2u(r.side.y) : 0; o[7] = r.material ? f2u(r.side.z) : 0;
        o[8] = f2u(r.pos.x); o[9] = f2u(r.pos.y); o[10] = f2u(r.pos.z); o[11] = r.fetches;
    }
    if (stats) { stats[0] = total.literal; stats[1] = total.jumps1; stats[2] = total.jumps2; stats[3] = total.jumps3; stats[4] = total.retrace; stats[5] = total.lookups; }
}

} // extern "C"
