// vrt_spec.h -- the numeric specification of the hot path, shared by the HIP kernels (device) and the
// host-side code of libvrt_hip.so.  Everything is fp32, evaluated in the written order with
// -ffp-contract=off (no FMA contraction), IEEE division and square root, so that results are
// reproducible bit-for-bit by any other implementation of the same formulas (the CPU oracle under
// oracle/ is one; it shares no code with this file).
//
// What GLSL leaves to the implementation (precision of normalize/atan/asin/exp, float->UNORM rounding,
// nearest-texel tie-breaking) is pinned here; DESIGN.md "Numeric spec" lists each choice with the
// shader line it resolves.
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VRT_HD __host__ __device__ __forceinline__
#else
#define VRT_HD inline
#endif

namespace vrt {

struct f3 { float x, y, z; };

VRT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
VRT_HD float fsign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
VRT_HD float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// (Measured and dropped, round 2: the compiler's IEEE sqrt / division expansions without their rescaling and fix-up steps
// behind a wave-uniform operand-range test -- exact, 7-9 instructions shorter each, and 0.3-1 % SLOWER in K1: the test and
// its branch cost what the scaling instructions did.)
VRT_HD float sqrt_spec(float x) { return sqrtf(x); }
VRT_HD float div_spec(float x, float y) { return x / y; }

VRT_HD float len3(f3 a) { return sqrt_spec(dot3(a, a)); }

// GLSL normalize(); normalize(0) := 0 (canonical rule A).
VRT_HD f3 normalize3(f3 a)
{
    float l = len3(a);
    if (l == 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    return mk3(a.x / l, a.y / l, a.z / l);
}

constexpr float kPi   = 3.14159265358979323846f;
constexpr float kPi_2 = 1.57079632679489661923f;
constexpr float kPi_4 = 0.78539816339744830962f;

// atan on [0,1] (Cephes atanf: one reduction step at tan(pi/8), degree-4 polynomial in t^2).
VRT_HD float atan_unit(float t)
{
    float y0 = 0.0f;
    const bool red = t > 0.4142135623730950f;
#if defined(__HIP_DEVICE_COMPILE__)
    // the reduced argument for the lanes that need it: numerator in [-0.586, 0], denominator in (1.41, 2] -- always operands
    // the short division is exact for (NaN lanes, atan(0, 0), take the plain operator with the whole wave)
    if (__ballot(red) != 0ull) {
        const float n = t - 1.0f, d = t + 1.0f;
        const float q = div_spec(n, d);
        if (red) { y0 = kPi_4; t = q; }
    }
#else
    if (red) { y0 = kPi_4; t = (t - 1.0f) / (t + 1.0f); }
#endif
    float z = t * t;
    float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z
               - 3.33329491539e-1f) * z * t + t;
    return p + y0;
}

// GLSL atan(y, x) as used by skyColor (voxel_volume.frag:101); atan(0,0) := 0.
VRT_HD float atan2_spec(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    const bool zero = ax == 0.0f && ay == 0.0f;
    bool swap = ay > ax;
    // the smaller over the larger: ONE division of selected operands (the two-sided form executes both divisions in a
    // wave whose lanes disagree about `swap`)
    float t = div_spec(swap ? ax : ay, swap ? ay : ax);
    float r = atan_unit(t);
    if (swap) r = kPi_2 - r;
    if (x < 0.0f) r = kPi - r;
    if (y < 0.0f) r = -r;
    return zero ? 0.0f : r;
}

// GLSL asin (voxel_volume.frag:101), Cephes asinf; |x| > 1 clamps.
VRT_HD float asin_spec(float x)
{
    float a = fabsf(x);
    bool big = a > 0.5f;
    if (a > 1.0f) a = 1.0f;
    float z, s;
    if (big) { z = 0.5f * (1.0f - a); s = sqrt_spec(z); }
    else     { s = a; z = a * a; }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z
                + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * s + s;
    if (big) p = kPi_2 - (p + p);
    return (x < 0.0f) ? -p : p;
}

// GLSL exp (denoiser.frag:55,60,65), Cephes expf; exp(+-0) == 1 exactly, < -87 -> 0.
VRT_HD float exp_spec(float x)
{
    if (x != x) return x;
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return INFINITY;
    float fx = floorf(x * 1.44269504088896341f + 0.5f);
    x = x - fx * 0.693359375f;
    x = x - fx * -2.12194440e-4f;
    float z = x * x;
    float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x
                 + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    int n = (int)fx;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)(n + 127) << 23;
    return p * s.f;
}

// float -> UNORM8 / SNORM8 (render-target conversion of RGBA8_UNORM / RGBA8_SNORM, geometry_stage.cpp:22,32).
VRT_HD uint8_t unorm8(float c)
{
    c = fminf(fmaxf(c, 0.0f), 1.0f);
    return (uint8_t)floorf(c * 255.0f + 0.5f);
}
VRT_HD int8_t snorm8(float c)
{
    c = fminf(fmaxf(c, -1.0f), 1.0f);
    return (int8_t)floorf(c * 127.0f + 0.5f);
}

// nearest filter + repeat addressing (texture_2d.cpp:158-163): texel index of coordinate u in [.., ..).
VRT_HD uint32_t wrap_texel(float u, uint32_t n)
{
    float fr = u - floorf(u);
    if (!(fr >= 0.0f)) return 0;
    int32_t i = (int32_t)floorf(fr * (float)n);
    if (i >= (int32_t)n) i = (int32_t)n - 1;
    return (uint32_t)i;
}

// glm::gauss(vec2(x,y), vec2(0), vec2(2)) = exp(-(x^2+y^2)/8)  (denoiser_stage.cpp:57)
constexpr float kGauss0 = 1.0f;
constexpr float kGauss1 = 0.8824969025845955f;   // exp(-1/8)
constexpr float kGauss2 = 0.7788007830714049f;   // exp(-1/4)

} // namespace vrt
