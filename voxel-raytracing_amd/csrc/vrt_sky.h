// vrt_sky.h -- which sky texel does a ray that misses everything land on?  (skyColor, voxel_volume.frag:98-105)
//
// For a pixel whose ray cannot hit anything (a wave outside the frame's box rectangle, or on a block without a tile tag) the
// ONLY observable of normalize(), atan(), asin() is the pair of integers (x, y) of the sky texel: the colour is a table
// look-up.  The numeric spec (vrt_spec.h) computes it with IEEE divisions, an IEEE square root and the Cephes polynomials:
// ~300 vector instructions per wave.  This header computes the same texel COORDINATES u * sky_w, v * sky_h with the
// hardware's 1-ulp reciprocal / reciprocal square root / square root and one short polynomial each, to within a bound
// `guard` (in texels) of the spec's own values, and says "don't know" for a lane that lies within that bound of a texel
// edge (or outside the range the bound was derived for).  A wave with such a lane takes the exact path as before, every
// other wave has by construction the texels the spec gives.  DESIGN.md 5 "Sky texel" derives the bound; tests/
// test_sky_fast.py measures the distance to the oracle's values on the CPU under worst-case +-1 ulp perturbation of the
// three hardware functions, tests/test_gpu_sky.py on the GPU itself (vrt_debug_sky_texels).
//
// Device: compiled into K1.  Host (tests/native/sky_host.cpp): the same text with the three hardware functions modelled as
// the correctly rounded result moved by a caller-chosen number of ulps.
#pragma once

#include "vrt_spec.h"

namespace vrt {

// per sky texture, made by the host when the sky is set (sky_fast_consts below)
struct SkyFastConsts {
    float ku, hu;          // u * sky_w = atan2(vz, vx) * ku + hu      (ku = RN(0.1591f * sky_w), hu = 0.5 * sky_w)
    float kv, hv;          // v * sky_h = asin(-vy / |v|) * kv + hv
    float tu, tv;          // 0.5 - guard: a lane is sure of its texel while |fract(coordinate) - 0.5| < t
    float wm1, hm1;        // sky_w - 1, sky_h - 1
    uint32_t w, h;         // texture size; w == 0: no fast path for this sky
};

// |vy| / |v| above which the fast path does not answer: d asin(a) / da = 1 / sqrt(1 - a^2) multiplies the difference
// between the two normalisations (3.6 at 0.96; rays within 16 degrees of the vertical take the exact path)
#define VRT_SKY_A_MAX 0.96f

// Error budget: how far the fast coordinate may lie from the spec's own (fp32) coordinate, in units of u and v themselves
// (i.e. before the multiplication by the texture size).  DESIGN.md 5 "Sky texel" adds up the roundings of both sides:
//   u: spec <= 2.1e-7 (two divisions by |v|, min/max quotient, Cephes atanf, pi/2 - r, pi - r, * 0.1591, + 0.5, fract * w),
//      fast <= 1.7e-7 (v_rcp_f32 1 ulp, degree-8 polynomial 1.04e-7 rad, the same two reflections, one fma)      => 3.8e-7
//   v: spec <= 4.3e-7, fast <= 4.7e-7 at |vy| / |v| = VRT_SKY_A_MAX, where d asin / da = 3.57 multiplies the two
//      normalisations' relative errors (2.1e-7: three roundings under the root, root, division; 2.7e-7: fused sum, v_rsq_f32
//      1 ulp, product)                                                                                            => 9.0e-7
// Measured (tests/test_sky_fast.py: 2 x 10^7 directions per texture size, each hardware function moved by -1 / 0 / +1 ulp):
// 1.2e-7 and 3.6e-7.  EPS_x + 2.4e-7 is the budget (the second term: the roundings of the two final multiplications by the
// texture size, kept apart because they scale with the coordinate, not with the angle).
#define VRT_SKY_EPS_U 3.0e-7f
#define VRT_SKY_EPS_V 8.0e-7f

inline SkyFastConsts sky_fast_consts(uint32_t w, uint32_t h)
{
    SkyFastConsts k;
    k.w = 0u; k.h = 0u; k.ku = k.hu = k.kv = k.hv = k.wm1 = k.hm1 = 0.0f; k.tu = k.tv = -1.0f;
    if (w == 0u || h == 0u || w > (1u << 20) || h > (1u << 20)) return k;
    // (k_primary addresses the RGBA8 copy of the sky with a 32-bit BYTE offset, (y * w + x) << 2: 2^30 texels would wrap it)
    if ((uint64_t)w * (uint64_t)h >= (1ull << 30)) return k;
    const float fw = (float)w, fh = (float)h;
    // guard = eps * size (the coordinate's own error) + 2 ulps of the product (the spec rounds fract * size once more, the fast
    // path's fused multiply-add rounds once)
    const float gu = VRT_SKY_EPS_U * fw + 2.4e-7f * fw, gv = VRT_SKY_EPS_V * fh + 2.4e-7f * fh;
    if (!(gu < 0.125f) || !(gv < 0.125f)) return k;            // textures so fine that the band would swallow the texel
    k.w = w; k.h = h;
    k.ku = 0.1591f * fw; k.hu = 0.5f * fw;
    k.kv = 0.3183f * fh; k.hv = 0.5f * fh;
    k.tu = 0.5f - gu; k.tv = 0.5f - gv;
    k.wm1 = (float)(w - 1u); k.hm1 = (float)(h - 1u);
    return k;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float sky_rcp(float x)  { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sky_rsq(float x)  { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sky_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float sky_fract(float x) { return __builtin_amdgcn_fractf(x); }
// clamp(x, 0, hi) in one instruction; a NaN comes out as a number in range as well
__device__ __forceinline__ uint32_t sky_index(float x, float hi) { return (uint32_t)__builtin_amdgcn_fmed3f(x, 0.0f, hi); }
#else
// host model: the correctly rounded value moved by g_sky_ulps[i] ulps (tests set them; 0 = the ideal function)
static thread_local int g_sky_ulps[3] = {0, 0, 0};
inline float sky_nudge(float x, int ulps)
{
    union { float f; int32_t i; } c; c.f = x;
    c.i += ulps;                                                // positive finite values only
    return c.f;
}
inline float sky_rcp(float x)  { return sky_nudge((float)(1.0 / (double)x), g_sky_ulps[0]); }
inline float sky_rsq(float x)  { return sky_nudge((float)(1.0 / sqrt((double)x)), g_sky_ulps[1]); }
inline float sky_sqrt(float x) { return x > 0.0f ? sky_nudge((float)sqrt((double)x), g_sky_ulps[2]) : 0.0f; }
inline float sky_fract(float x) { return x - floorf(x); }
inline uint32_t sky_index(float x, float hi) { return x > 0.0f ? (uint32_t)(x < hi ? x : hi) : 0u; }
#endif

// vx, vy, vz: the UNNORMALISED direction of main()'s ray generation (frag:312-319), bit for bit the spec's values.
// true: (tx, ty) is the texel skyColor(normalize(v)) reads under the numeric spec.  false: not decided here.
// Every lane computes everything (no divergence); the caller votes on the result.
// (un, vn: the coordinates themselves, u * sky_w and v * sky_h -- the tests measure their distance to the spec's)
VRT_HD bool sky_texel_fast(float vx, float vy, float vz, const SkyFastConsts& k, uint32_t& tx, uint32_t& ty, float& un, float& vn)
{
    const float ax = fabsf(vx), ay = fabsf(vy), az = fabsf(vz);
    // ---- u: atan2(vz, vx) needs no normalisation (the spec's d.z / d.x differs from vz / vx by two roundings) ----
    const float hi = fmaxf(ax, az), lo = fminf(ax, az);
    const float t = lo * sky_rcp(hi);                           // in [0, 1]
    const float z = t * t;
    // atan(t) = t * P(t^2) on [0, 1], degree 8 (tools/fit_sky_poly.py: 1.04e-7 rad in float32 arithmetic)
    float p = __builtin_fmaf(2.903554356e-03f, z, -1.628301479e-02f);
    p = __builtin_fmaf(p, z, 4.303938150e-02f);
    p = __builtin_fmaf(p, z, -7.533676922e-02f);
    p = __builtin_fmaf(p, z, 1.065467820e-01f);
    p = __builtin_fmaf(p, z, -1.420713365e-01f);
    p = __builtin_fmaf(p, z, 1.999305338e-01f);
    p = __builtin_fmaf(p, z, -3.333309293e-01f);
    p = __builtin_fmaf(p, z, 1.0f);
    float r = p * t;
    if (az > ax) r = kPi_2 - r;
    if (vx < 0.0f) r = kPi - r;
    r = copysignf(r, vz);
    un = __builtin_fmaf(r, k.ku, k.hu);
    // ---- v: asin(-vy / |v|), Cephes asinf's two ranges (vrt_spec.h asin_spec) on hardware rsq / sqrt ----
    const float l2 = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
    const float a = ay * sky_rsq(l2);
    const bool big = a > 0.5f;
    const float zb = __builtin_fmaf(a, -0.5f, 0.5f);
    const float zz = big ? zb : a * a;
    const float s = big ? sky_sqrt(zb) : a;
    float q = __builtin_fmaf(4.2163199048e-2f, zz, 2.4181311049e-2f);
    q = __builtin_fmaf(q, zz, 4.5470025998e-2f);
    q = __builtin_fmaf(q, zz, 7.4953002686e-2f);
    q = __builtin_fmaf(q, zz, 1.6666752422e-1f);
    q = __builtin_fmaf(q * zz, s, s);
    if (big) q = __builtin_fmaf(-2.0f, q, kPi_2);
    q = copysignf(q, -vy);
    vn = __builtin_fmaf(q, k.kv, k.hv);
    // ---- the texel, and whether the lane is sure of it ----
    const float fu = sky_fract(un), fv = sky_fract(vn);
    // (written so that a NaN anywhere answers "not sure")
    // (bitwise, not short-circuit: six compares and scalar ANDs of their masks, no divergent branch)
    const bool sure = (fabsf(fu - 0.5f) < k.tu) & (fabsf(fv - 0.5f) < k.tv) & (a <= VRT_SKY_A_MAX) &
                      (lo >= 0x1p-40f) & (ay >= 0x1p-40f) & (l2 <= 0x1p80f);
    tx = sky_index(un, k.wm1);                                  // truncation; both coordinates are positive
    ty = sky_index(vn, k.hm1);
    return sure;
}
VRT_HD bool sky_texel_fast(float vx, float vy, float vz, const SkyFastConsts& k, uint32_t& tx, uint32_t& ty)
{
    float un, vn;
    return sky_texel_fast(vx, vy, vz, k, tx, ty, un, vn);
}

} // namespace vrt
