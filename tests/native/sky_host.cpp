// sky_host.cpp -- TEST HELPER: the product's sky-texel fast path (csrc/vrt_sky.h) compiled for the host with its three
// hardware functions (v_rcp_f32, v_rsq_f32, v_sqrt_f32: 1 ulp each) modelled as the correctly rounded value moved by -1, 0 or
// +1 ulp, next to the numeric spec's own texel (csrc/vrt_spec.h: normalize3, atan2_spec, asin_spec, wrap_texel).  Built on
// demand by tests/test_sky_fast.py with g++; never linked into libvrt_hip.so.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../voxel-raytracing_amd/csrc/vrt_sky.h"

using namespace vrt;

static inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

extern "C" {

// n directions v (unnormalised, float32 triples).  ulp_mode: 0 = ideal hardware functions, 1 = each call moved by a pseudo-
// random -1 / 0 / +1 ulp (seeded per direction), 2 / 3 = all three by +1 / -1.
// out (per direction, 8 floats): fast u * w, fast v * h, spec u, spec v, fast tx, fast ty (as floats), sure, spec texel match
// returns the number of directions whose fast texel is "sure" and differs from the spec's (must be 0)
uint64_t sky_compare(uint64_t n, const float* v, uint32_t w, uint32_t h, int ulp_mode, uint32_t seed, float* out, int nthreads,
                     double* max_du, double* max_dv, uint64_t* n_sure)
{
    const SkyFastConsts k = sky_fast_consts(w, h);
    if (nthreads < 1) nthreads = 1;
    std::vector<uint64_t> wrong((size_t)nthreads, 0), sure((size_t)nthreads, 0);
    std::vector<double> du((size_t)nthreads, 0.0), dv((size_t)nthreads, 0.0);
    auto work = [&](int th) {
        for (uint64_t i = (uint64_t)th; i < n; i += (uint64_t)nthreads) {
            const float vx = v[3 * i], vy = v[3 * i + 1], vz = v[3 * i + 2];
            if (ulp_mode == 1) {
                const uint32_t r = mix((uint32_t)i * 2654435761u ^ seed);
                g_sky_ulps[0] = (int)(r % 3u) - 1; g_sky_ulps[1] = (int)((r >> 8) % 3u) - 1; g_sky_ulps[2] = (int)((r >> 16) % 3u) - 1;
            } else if (ulp_mode == 2) { g_sky_ulps[0] = g_sky_ulps[1] = g_sky_ulps[2] = 1; }
            else if (ulp_mode == 3) { g_sky_ulps[0] = g_sky_ulps[1] = g_sky_ulps[2] = -1; }
            else { g_sky_ulps[0] = g_sky_ulps[1] = g_sky_ulps[2] = 0; }
            uint32_t tx = 0, ty = 0;
            float un = 0.0f, vn = 0.0f;
            const bool ok = k.w != 0u && sky_texel_fast(vx, vy, vz, k, tx, ty, un, vn);
            // the spec: skyColor(normalize(v)) (vrt_device.hip sky_color)
            const f3 d = normalize3(mk3(vx, vy, vz));
            const float us = atan2_spec(d.z, d.x) * 0.1591f + 0.5f;
            const float vs = asin_spec(-d.y) * 0.3183f + 0.5f;
            const uint32_t sx = wrap_texel(us, w), sy = wrap_texel(vs, h);
            // distance of the coordinates, in u and v themselves, over the lanes the range conditions admit (the guard band
            // is what this distance has to stay inside, so lanes near an edge count here too)
            const float ay = fabsf(vy), lo = fminf(fabsf(vx), fabsf(vz));
            const bool in_range = k.w != 0u && ay >= 0x1p-40f && lo >= 0x1p-40f && fabsf(d.y) <= VRT_SKY_A_MAX - 1e-3f;
            if (in_range) {
                // (the spec's coordinate as the kernel forms it: fract * size, rounded once more)
                const float fus = us - floorf(us), fvs = vs - floorf(vs);
                const double eu = fabs((double)un - (double)(fus * (float)w)) / (double)w, ev = fabs((double)vn - (double)(fvs * (float)h)) / (double)h;
                if (eu > du[(size_t)th]) du[(size_t)th] = eu;
                if (ev > dv[(size_t)th]) dv[(size_t)th] = ev;
            }
            if (ok) {
                sure[(size_t)th]++;
                if (tx != sx || ty != sy) wrong[(size_t)th]++;
            }
            if (out) {
                float* o = out + 8 * i;
                o[0] = un; o[1] = vn; o[2] = us; o[3] = vs; o[4] = (float)tx; o[5] = (float)ty; o[6] = ok ? 1.0f : 0.0f;
                o[7] = (tx == sx && ty == sy) ? 1.0f : 0.0f;
            }
        }
    };
    std::vector<std::thread> ts;
    for (int t = 0; t < nthreads; t++) ts.emplace_back(work, t);
    for (auto& t : ts) t.join();
    uint64_t bad = 0, ns = 0;
    for (int t = 0; t < nthreads; t++) { bad += wrong[(size_t)t]; ns += sure[(size_t)t]; }
    double mu = 0.0, mv = 0.0;
    for (int t = 0; t < nthreads; t++) { mu = du[(size_t)t] > mu ? du[(size_t)t] : mu; mv = dv[(size_t)t] > mv ? dv[(size_t)t] : mv; }
    if (max_du) *max_du = mu;
    if (max_dv) *max_dv = mv;
    if (n_sure) *n_sure = ns;
    return bad;
}

} // extern "C"
