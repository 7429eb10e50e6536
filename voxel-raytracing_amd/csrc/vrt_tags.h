// vrt_tags.h -- the projection behind the tile tags (k_tile_tags, vrt_device.hip): the screen rectangle of a cell of the volume
// under a frame's camera, in fp32, TOGETHER WITH A BOUND on how far the computed rectangle can lie from the true one.
//
// A block of pixels without a tag is not traced, so a rectangle that comes out too small is a silently wrong pixel.  The
// rectangle is therefore used only while its own error bound stays below what the margin of the tags absorbs:
//
//   ray of screen position (a, b) in [-1, 1]^2 (frag:312-319):  cam + lambda (C + a U + b V),   C = cd + jitter
//   [U V C] (a lambda, b lambda, lambda)^T = p - cam   =>   a = A / L, b = B / L, lambda = L / det  with
//   A = p . (V x C),  B = p . (C x U),  L = p . (U x V),  det = U . (V x C);   pixel x = a W/2 + W/2, y = b H/2 + H/2.
//
// Rounding (u = 2^-24, no fused operations; DESIGN.md 5 "Tile tags: the bound" has the steps):
//   a cross-product component x1 y2 - x2 y1 carries at most 3u m with m = |x1 y2| + |x2 y1| (two products, one difference,
//   and the one rounding of C = cd + jitter), the three-term dot products 3u more, p - cam one u:
//       |dA| <= 7u Q0 + u |A|,   Q0 = sum_i (|p_i| + ext) m0_i        (likewise B with m1, L with m2)
//   and a = A / L through v_rcp_f32 (1 ulp = 2u), one product, then a * W/2 + W/2 (two roundings):
//       |dx| <= (W/2) (7u (Q0 + |a| Q2) / |L| + u (2 |a|))  +  (W/2) 3u |a|  +  4u W/2
//            <= 8u (W/2) (Q0 + amax Q2) / Lmin + 10u (W/2) (amax + 1)           =: ex   (the factor 8 for 7 covers the evaluation of
//                                                                                       the bound itself in fp32)
// The tags grow a rectangle by VRT_TAG_MARGIN_PX = 2 pixels; a pixel belongs to the block its CENTRE lies in, which takes half
// a pixel of that; a cell whose ex or ey exceeds VRT_TAG_ERR_MAX_PX = 1.5 makes the frame's tags say nothing instead.
// tests/test_tile_tag_bound.py evaluates the computed rectangle against exact rational arithmetic over random and adversarial
// cameras (nearly coplanar bases, cells grazing the camera plane, far and tiny volumes): the true rectangle always lies within
// the computed one grown by the reported bound, and the bound is the number the constants below promise.
#pragma once

#include "vrt_spec.h"

namespace vrt {

#define VRT_TAG_MARGIN_PX 2.0f
#define VRT_TAG_ERR_MAX_PX 1.5f

struct TagCam {
    float U[3], V[3], C[3], cam[3];    // cam_right, planeV (cam_up * H / W), cd + jitter (as RayGenConsts holds them), cam_pos
    float W, H;
};

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float tag_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#else
static thread_local int g_tag_rcp_ulps = 0;                    // host model of v_rcp_f32: the correctly rounded value moved by this many ulps
inline float tag_rcp(float x)
{
    union { float f; int32_t i; } c; c.f = (float)(1.0 / (double)x);
    c.i += (c.f > 0.0f ? 1 : -1) * g_tag_rcp_ulps;
    return c.f;
}
#endif

// The rectangle of the cell [lo, lo + ext]^3 (the caller has grown it by its voxel of margin) in pixel coordinates, and the
// bound (ex, ey) on its error.  Returns 0: usable; bit 0: nothing is known (a corner behind or on the camera plane within
// rounding, a degenerate basis, a NaN): the frame's tags must say nothing; bit 1: the rectangle and its bound are valid but
// the bound is above VRT_TAG_ERR_MAX_PX (the caller may still drop a cell that lies off the screen by more than its bound).
VRT_HD int tag_project(const TagCam& k, const float lo[3], float ext, float& x0, float& x1, float& y0, float& y1, float& ex, float& ey)
{
    const float* U = k.U; const float* V = k.V; const float* C = k.C;
    const float c0[3] = {V[1] * C[2] - V[2] * C[1], V[2] * C[0] - V[0] * C[2], V[0] * C[1] - V[1] * C[0]};   // V x C
    const float c1[3] = {C[1] * U[2] - C[2] * U[1], C[2] * U[0] - C[0] * U[2], C[0] * U[1] - C[1] * U[0]};   // C x U
    const float c2[3] = {U[1] * V[2] - U[2] * V[1], U[2] * V[0] - U[0] * V[2], U[0] * V[1] - U[1] * V[0]};   // U x V
    // m: the magnitudes behind each cross-product component (what its rounding error is proportional to)
    const float m0[3] = {fabsf(V[1] * C[2]) + fabsf(V[2] * C[1]), fabsf(V[2] * C[0]) + fabsf(V[0] * C[2]), fabsf(V[0] * C[1]) + fabsf(V[1] * C[0])};
    const float m1[3] = {fabsf(C[1] * U[2]) + fabsf(C[2] * U[1]), fabsf(C[2] * U[0]) + fabsf(C[0] * U[2]), fabsf(C[0] * U[1]) + fabsf(C[1] * U[0])};
    const float m2[3] = {fabsf(U[1] * V[2]) + fabsf(U[2] * V[1]), fabsf(U[2] * V[0]) + fabsf(U[0] * V[2]), fabsf(U[0] * V[1]) + fabsf(U[1] * V[0])};
    const float det = U[0] * c0[0] + U[1] * c0[1] + U[2] * c0[2];
    const float mdet = fabsf(U[0]) * m0[0] + fabsf(U[1]) * m0[1] + fabsf(U[2]) * m0[2];
    // the sign of det must be certain (it decides what "in front of the camera" means): |d det| <= 7u mdet
    bool all = !(fabsf(det) > 1e-5f * mdet);
    // a, b, lambda are linear in p: the corner's numerators once, the other seven corners by additions
    const float p0[3] = {lo[0] - k.cam[0], lo[1] - k.cam[1], lo[2] - k.cam[2]};
    const float A0 = p0[0] * c0[0] + p0[1] * c0[1] + p0[2] * c0[2], B0 = p0[0] * c1[0] + p0[1] * c1[1] + p0[2] * c1[2];
    const float L0 = p0[0] * c2[0] + p0[1] * c2[1] + p0[2] * c2[2];
    const float pa[3] = {fabsf(p0[0]) + ext, fabsf(p0[1]) + ext, fabsf(p0[2]) + ext};
    const float Q0 = pa[0] * m0[0] + pa[1] * m0[1] + pa[2] * m0[2], Q1 = pa[0] * m1[0] + pa[1] * m1[1] + pa[2] * m1[2];
    const float Q2 = pa[0] * m2[0] + pa[1] * m2[1] + pa[2] * m2[2];
    const float hw = 0.5f * k.W, hh = 0.5f * k.H;
    const float sgn = det < 0.0f ? -1.0f : 1.0f;
    x0 = 1e30f; x1 = -1e30f; y0 = 1e30f; y1 = -1e30f;
    float lmin = 1e30f;                                        // the smallest L * sign(det) of the corners: > 0 <=> in front of the camera
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int c = 0; c < 8; c++) {
        const float dx = (c & 1) ? ext : 0.0f, dy = (c & 2) ? ext : 0.0f, dz = (c & 4) ? ext : 0.0f;
        const float An = A0 + (dx * c0[0] + dy * c0[1] + dz * c0[2]), Bn = B0 + (dx * c1[0] + dy * c1[1] + dz * c1[2]);
        const float Ln = L0 + (dx * c2[0] + dy * c2[1] + dz * c2[2]);
        lmin = fminf(lmin, Ln * sgn);
        const float rl = tag_rcp(Ln);
        const float fx = An * rl * hw + hw, fy = Bn * rl * hh + hh;
        x0 = fminf(x0, fx); x1 = fmaxf(x1, fx); y0 = fminf(y0, fy); y1 = fmaxf(y1, fy);
    }
    // in front of the camera plane with certainty: the smallest |L| must exceed its own error bound (7u Q2 + u |L|)
    const float u = 5.9604645e-8f;
    if (!(lmin > 16.0f * u * Q2)) all = true;
    if (!(x0 == x0) || !(x1 == x1) || !(y0 == y0) || !(y1 == y1)) all = true;
    // |a|, |b| of the corners from the rectangle itself
    const float amax = fmaxf(fabsf(x0 - hw), fabsf(x1 - hw)) * tag_rcp(hw), bmax = fmaxf(fabsf(y0 - hh), fabsf(y1 - hh)) * tag_rcp(hh);
    const float rL = tag_rcp(lmin);
    ex = 8.0f * u * hw * (Q0 + amax * Q2) * rL + 10.0f * u * hw * (amax + 1.0f);
    ey = 8.0f * u * hh * (Q1 + bmax * Q2) * rL + 10.0f * u * hh * (bmax + 1.0f);
    const bool loose = !(ex <= VRT_TAG_ERR_MAX_PX) || !(ey <= VRT_TAG_ERR_MAX_PX);
    return (all ? 1 : 0) | (loose ? 2 : 0);
}

// A rank of a sharded launch traces one band of pixel rows [ylo, yhi) of every frame, so seven cells in eight of an 8-GPU step's
// tag kernel project onto rows that are somebody else's.  This says so without the eight divisions: a cell's pixel row is
// y = (B / L) H/2 + H/2, so "every corner above the band" is  B sgn - t L sgn < 0  with t = (ylo - margin - H/2) / (H/2)  -- a
// LINEAR function of the corner, whose largest value over the box is its value at one corner plus the positive parts of its
// three increments; likewise "every corner below".  Only cells whose corners all lie in front of the camera plane with certainty
// are judged.  Rounding: B and L carry 7u Q1 and 7u Q2 (above), t two roundings on a value below 2 in magnitude for any band on
// the screen, the combination three more: the verdict is given only when it holds by more than 16u (Q1 + (|t| + 1) Q2).
// Returns true: no pixel row of [ylo, yhi), grown by the tags' margin, sees the cell -- the same cells the full projection would
// give no tag in the band (up to both computations' bounds, each of which errs on the side of tagging).
VRT_HD bool tag_band_cull(const TagCam& k, const float lo[3], float ext, float ylo, float yhi)
{
    const float* U = k.U; const float* V = k.V; const float* C = k.C;
    const float c0[3] = {V[1] * C[2] - V[2] * C[1], V[2] * C[0] - V[0] * C[2], V[0] * C[1] - V[1] * C[0]};   // V x C
    const float c1[3] = {C[1] * U[2] - C[2] * U[1], C[2] * U[0] - C[0] * U[2], C[0] * U[1] - C[1] * U[0]};   // C x U
    const float c2[3] = {U[1] * V[2] - U[2] * V[1], U[2] * V[0] - U[0] * V[2], U[0] * V[1] - U[1] * V[0]};   // U x V
    const float m0[3] = {fabsf(V[1] * C[2]) + fabsf(V[2] * C[1]), fabsf(V[2] * C[0]) + fabsf(V[0] * C[2]), fabsf(V[0] * C[1]) + fabsf(V[1] * C[0])};
    const float m1[3] = {fabsf(C[1] * U[2]) + fabsf(C[2] * U[1]), fabsf(C[2] * U[0]) + fabsf(C[0] * U[2]), fabsf(C[0] * U[1]) + fabsf(C[1] * U[0])};
    const float m2[3] = {fabsf(U[1] * V[2]) + fabsf(U[2] * V[1]), fabsf(U[2] * V[0]) + fabsf(U[0] * V[2]), fabsf(U[0] * V[1]) + fabsf(U[1] * V[0])};
    const float det = U[0] * c0[0] + U[1] * c0[1] + U[2] * c0[2];
    const float mdet = fabsf(U[0]) * m0[0] + fabsf(U[1]) * m0[1] + fabsf(U[2]) * m0[2];
    if (!(fabsf(det) > 1e-5f * mdet)) return false;            // (the sign of det decides what "in front" means: tag_project's own test)
    const float sgn = det < 0.0f ? -1.0f : 1.0f;
    const float p0[3] = {lo[0] - k.cam[0], lo[1] - k.cam[1], lo[2] - k.cam[2]};
    const float B0 = (p0[0] * c1[0] + p0[1] * c1[1] + p0[2] * c1[2]) * sgn, L0 = (p0[0] * c2[0] + p0[1] * c2[1] + p0[2] * c2[2]) * sgn;
    const float pa[3] = {fabsf(p0[0]) + ext, fabsf(p0[1]) + ext, fabsf(p0[2]) + ext};
    const float Q1 = pa[0] * m1[0] + pa[1] * m1[1] + pa[2] * m1[2], Q2 = pa[0] * m2[0] + pa[1] * m2[1] + pa[2] * m2[2];
    const float u = 5.9604645e-8f;
    // every corner in front of the camera plane, with certainty: the smallest L sgn over the box
    const float l0 = c2[0] * sgn * ext, l1 = c2[1] * sgn * ext, l2 = c2[2] * sgn * ext;
    const float lmin = L0 + (fminf(l0, 0.0f) + fminf(l1, 0.0f) + fminf(l2, 0.0f));
    if (!(lmin > 16.0f * u * Q2)) return false;
    const float hh = 0.5f * k.H, rh = tag_rcp(hh);
    const float m = VRT_TAG_MARGIN_PX;
    const float tlo = ((ylo - m) - hh) * rh, thi = ((yhi + m) - hh) * rh;
    const float b0 = c1[0] * sgn * ext, b1 = c1[1] * sgn * ext, b2 = c1[2] * sgn * ext;
    {   // above the band: max over the corners of B - tlo L < -E
        const float f0 = b0 - tlo * l0, f1 = b1 - tlo * l1, f2 = b2 - tlo * l2;
        const float fmax = (B0 - tlo * L0) + (fmaxf(f0, 0.0f) + fmaxf(f1, 0.0f) + fmaxf(f2, 0.0f));
        const float E = 16.0f * u * (Q1 + (fabsf(tlo) + 1.0f) * Q2);
        if (fmax < -E) return true;
    }
    {   // below the band: min over the corners of B - thi L > E
        const float f0 = b0 - thi * l0, f1 = b1 - thi * l1, f2 = b2 - thi * l2;
        const float fmin = (B0 - thi * L0) + (fminf(f0, 0.0f) + fminf(f1, 0.0f) + fminf(f2, 0.0f));
        const float E = 16.0f * u * (Q1 + (fabsf(thi) + 1.0f) * Q2);
        if (fmin > E) return true;
    }
    return false;
}

} // namespace vrt
