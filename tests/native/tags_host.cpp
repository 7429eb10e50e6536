// tags_host.cpp -- TEST HELPER: the product's tile-tag projection (csrc/vrt_tags.h) compiled for the host, v_rcp_f32 modelled
// as the correctly rounded reciprocal moved by -1 / 0 / +1 ulp.  Built on demand by tests/test_tile_tag_bound.py; never linked
// into libvrt_hip.so.
#include <cstdint>
#include "../../voxel-raytracing_amd/csrc/vrt_tags.h"

using namespace vrt;

extern "C" {

// n cases; cams: 14 floats each (U, V, C, cam, W, H); cells: 4 floats each (lo xyz, ext); out: 7 floats each
// (x0, x1, y0, y1, ex, ey, status)
void tags_project(int n, const float* cams, const float* cells, int rcp_ulps, float* out)
{
    g_tag_rcp_ulps = rcp_ulps;
    for (int i = 0; i < n; i++) {
        TagCam k;
        const float* c = cams + 14 * i;
        for (int a = 0; a < 3; a++) { k.U[a] = c[a]; k.V[a] = c[3 + a]; k.C[a] = c[6 + a]; k.cam[a] = c[9 + a]; }
        k.W = c[12]; k.H = c[13];
        float x0, x1, y0, y1, ex, ey;
        const int st = tag_project(k, cells + 4 * i, cells[4 * i + 3], x0, x1, y0, y1, ex, ey);
        float* o = out + 7 * i;
        o[0] = x0; o[1] = x1; o[2] = y0; o[3] = y1; o[4] = ex; o[5] = ey; o[6] = (float)st;
    }
    g_tag_rcp_ulps = 0;
}

// bands: 2 floats each (ylo, yhi); out: 1 byte each: the cell is certainly outside the band's rows (tag_band_cull)
void tags_band_cull(int n, const float* cams, const float* cells, const float* bands, int rcp_ulps, unsigned char* out)
{
    g_tag_rcp_ulps = rcp_ulps;
    for (int i = 0; i < n; i++) {
        TagCam k;
        const float* c = cams + 14 * i;
        for (int a = 0; a < 3; a++) { k.U[a] = c[a]; k.V[a] = c[3 + a]; k.C[a] = c[6 + a]; k.cam[a] = c[9 + a]; }
        k.W = c[12]; k.H = c[13];
        out[i] = tag_band_cull(k, cells + 4 * i, cells[4 * i + 3], bands[2 * i], bands[2 * i + 1]) ? 1 : 0;
    }
    g_tag_rcp_ulps = 0;
}

} // extern "C"
