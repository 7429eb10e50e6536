// The verified denoiser pass: how far the cheap evaluation of a pixel can lie from the specified one.
//
// denoiser.frag:38-73 ends in a RGBA8 target: of everything a weighted pass computes for a pixel only floor(mean * 255 + 0.5)
// per channel is ever seen.  k_denoise_ver evaluates the mean with the hardware's exponential and reciprocal, fused
// multiply-adds and exact integer code distances, and then redoes -- the shader's own way, the numeric spec's operations
// one by one -- every pixel whose cheap mean lies within `guard` codes of a rounding boundary.  The result is the
// spec's, bit for bit, as long as
//
//        | y_fast - y_spec |  <=  guard          (both in RGBA8 code units, before the floor)
//
// for every input.  This header derives that bound.  Both evaluations are compared with the real-valued
//
//        R = sum_i b_i kappa_i w_i / sum_i kappa_i w_i,   w_i = exp(-(Dc/phi_c + Dn/(sw^2 phi_n) + Dp/phi_p)),
//
// b_i = colour code / 255 of tap i, kappa_i the kernel weight (the fp32 constant both use), Dc = sum (code difference)^2 / 255^2,
// Dn likewise over 127^2 (code -128 reads as -127), Dp = |position difference|^2, all in exact arithmetic.  eps = 2^-24.
//
// (a) a weight of the spec.  Decoded codes are RN(c / 255), RN(c / 127).  Over all pairs of codes the squared difference as the
//     spec computes it, RN(RN(RN(a/D) - RN(b/D))^2), differs from the exact ((a - b)/D)^2 by at most C1 |a - b| / D with
//     C1 = 2.12 eps (colour), 4.49 eps (normal) -- enumerated, tests/test_denoise_bound.py; equal codes give exactly 0.  With
//     sum_k |t_k| <= 2 sqrt(D) and three rounded additions:  |D_spec - D| <= 2 C1 sqrt(D) + 3.01 eps D.  The quotient by phi adds
//     one rounding (colour), two (normal: / sw^2, / phi); the position distance has relative error <= 6.03 eps (RN of each
//     difference, of each square, three additions) and its quotient 7.05 eps.  So, with Q = D / A  (A = phi_c, sw^2 phi_n, phi_p):
//         dQ_c <= 2 C1c sqrt(Q_c / phi_c) + 4.02 eps Q_c,   dQ_n <= 2 C1n sqrt(Q_n / A_n) + 5.03 eps Q_n,   dQ_p <= 7.05 eps Q_p
//     and since  exp(-Q) sqrt(Q) <= 0.4289,  exp(-Q) Q <= 0.3679:
//         w dQ  <=  ALPHA_S := 0.8578 (C1c / sqrt(phi_c) + C1n / sqrt(A_n)) + 0.3679 (4.02 + 5.03 + 7.05) eps.
//     min(exp_spec(q), 1) lies within EX = 1.364 eps (relative) of exp(q) for every fp32 q in [-87, -0] -- exhaustive scan,
//     tools/exp_spec_error.c; below -87 the spec returns 0 where exp(q) < 1.7e-38; two rounded products:
//         | w_spec - w |  <=  (3 EX + 2 eps) w + ALPHA_S + 1e-37          (second-order terms: the final factor 1.01)
// (b) the mean.  Perturbing the weights of the non-centre taps by dw_i (the centre tap's distances are 0, its weight is exactly 1
//     in both evaluations) moves R by at most  sum kappa_i |dw_i| max|b_i - R| / T'  with |b_i - R| <= 1 and T' >= kappa_centre:
//         | R(w_spec) - R |  <=  (3 EX + 2 eps) + ALPHA_S K / kappa_centre,      K = sum of the non-centre kappa_i.
//     The spec's own sums: a term RN(RN(b w) kappa) with b = RN(c/255) carries 3 eps, eight additions 8 eps, the total's terms
//     1 eps + 8 eps, the quotient 1 eps: 21.3 eps relative, on a value <= 1.  RN(RN(x 255) + 0.5): 255 eps + 256 eps codes.
// (c) the cheap evaluation.  Code distances are exact integers (< 2^24, exact as floats); the position distance by fused
//     multiply-adds: 6 eps.  e = fma(Dp, kp, fma(Dc, kc, fma(Dn, kn, lk))) with every k rounded once from double and
//     lk = RN(-log2 kappa): |e - e*| <= 8.05 eps e*, where kappa w = 2^-e*.  v_exp_f32 and v_rcp_f32 are accurate to 1 ulp
//     (<= 2 eps relative); x 2^-x <= 0.5307:
//         | W - kappa w |  <=  2 eps kappa w + 0.6931 * 8.05 * 0.5307 eps  =  2 eps kappa w + 2.962 eps     (+ 1e-37 flushed)
//     -- and 2 eps more: the exponent's constant is -log2 of the Gaussian itself, the spec's kappa that Gaussian rounded to fp32 --
//     and  | R(W) - R | <= 4 eps + n_taps' * 2.962 eps / kappa_centre.  Sums: the centre term (one rounding unless kappa = 1), eight
//     fused accumulations, eight additions of the total, the reciprocal 2 eps: 19 eps relative; fma(sum, r, 0.5): 256 eps codes.
//
// guard = 1.01 * 255 * [ (3 EX + 2 eps) + ALPHA_S K / kc + 21.3 eps  +  4 eps + n 2.962 eps / kc + 19 eps ] + 767 eps.
// At the reference's defaults (pass 1: phi 20.4 / 0.01 / 0.1, stepWidth 3) it is 3.2e-3 of a code: about 2 % of the pixels of
// a frame are redone.  Non-finite positions make the cheap mean NaN, which fails the guard's comparison: redone too.
#pragma once
#include <cmath>

namespace vrt {

constexpr double kDenEps = 5.9604644775390625e-8;        // 2^-24
constexpr double kDenC1Color = 2.2 * kDenEps;            // >= 2.12 eps (enumerated)
constexpr double kDenC1Normal = 4.6 * kDenEps;           // >= 4.49 eps (enumerated)
constexpr double kDenExpSpec = 1.4 * kDenEps;            // >= 1.364 eps (exhaustive)
constexpr double kDenGuardMax = 0.02;                    // beyond this the exact kernels run the pass themselves

// guard (RGBA8 codes) of a weighted pass; +inf when the pass is not one the bound covers
inline double denoise_guard(double phi_color, double phi_normal, double phi_pos, double step_width, bool shipped)
{
    const double eps = kDenEps;
    if (!(phi_color >= 0x1p-20 && phi_color <= 0x1p20 && phi_normal >= 0x1p-20 && phi_normal <= 0x1p20 &&
          phi_pos >= 0x1p-20 && phi_pos <= 0x1p20)) return INFINITY;
    if (!(step_width >= 1.0 && step_width <= 5.0 && step_width == std::floor(step_width))) return INFINITY;
    const double G0 = 1.0, G1 = 0.8824969025845955, G2 = 0.7788007830714049;
    const double kcen = shipped ? G2 : G0;                               // the centre tap's kernel weight
    const double K = shipped ? G2 + G0 : 4.0 * G1 + 4.0 * G2;            // the others'
    const int n = shipped ? 2 : 8;
    const double An = step_width * step_width * phi_normal;
    const double alpha_s = 0.8578 * (kDenC1Color / std::sqrt(phi_color) + kDenC1Normal / std::sqrt(An)) + 0.3679 * (4.02 + 5.03 + 7.05) * eps + 1e-37;
    const double spec = (3.0 * kDenExpSpec + 2.0 * eps) + alpha_s * K / kcen + 21.3 * eps;
    const double fast = 4.0 * eps + n * (2.962 * eps + 1e-37) / kcen + 19.0 * eps;
    return 1.01 * 255.0 * (spec + fast) + 767.0 * eps;
}

// Pass 0 (phi = +inf: every edge-stopping weight is exactly 1, the pass is a plain 3 x 3 blur; tap offset 1).  The spec: a term
// RN(RN(c / 255) kappa) carries 2 eps, eight additions 8 eps, the total's eight additions 8 eps, the quotient 1 eps: 19.2 eps
// relative on a value <= 1, then 511 eps codes of the two final roundings.  The cheap form: nine fused accumulations of
// code * kappa, the reciprocal of the weights' sum rounded once from double (the sum of the Gaussians themselves: within 2 eps
// of the sum of the spec's fp32 constants), one fused multiply-add: 12.1 eps relative, 256 eps.
// The separable cheap form (k_denoise_p0, round 4): h = fma(l, g, fma(r, g, c)) per row and fma(h_up, g, fma(h_down, g, h)) per pixel
// with g the fp32 constant kappa_1.  Edge taps weigh exactly kappa_1; corner taps weigh kappa_1^2, within 3.01 eps (relative) of the
// spec's kappa_2 (each constant is the Gaussian G1, G2 = G1^2 rounded once), and carry 4 kappa_2 / sum = 0.41 of the weight: 1.24 eps
// on the mean.  Four fused roundings, each at most eps of the final sum (every term is non-negative): 4 eps.  The reciprocal and the
// last multiply-add as above: 3.1 eps, 256 eps.  8.4 eps relative: inside the 12.1 the guard is made of.
inline double denoise_guard_pass0()
{
    return 1.01 * 255.0 * (19.2 + 12.1) * kDenEps + 767.0 * kDenEps;
}

} // namespace vrt
