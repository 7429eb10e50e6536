// vrt_traverse.h -- the grid march of voxel_volume.frag:109-174 (boxIntersection + traceRayInt) in three
// traversal strategies that all produce the SAME RayInt (hit cell, mask, sideDist, material) bit for bit:
//
//   DENSE    one R8 fetch per DDA iteration (the literal shader loop)
//   BITMASK  same iterations; the solid test reads the 4^3 occupancy word cached in registers and the
//            16^3 summary (LDS) before touching global memory; the R8 id is fetched once, at the hit
//   DF       the wave agrees (one DPP min-reduction) on a number of iterations no lane needs a memory test for (distance
//            field clearance) and runs them as pure ALU stepping; one gather per run instead of per iteration
//   JUMP     BITMASK inside occupied 4^3 cells; across EMPTY pyramid cells (4^3 / 16^3 / 64^3) one iteration
//            replaces all the DDA iterations up to the cell's exit -- exactly (see "exact jumps" below)
//
// Compiled for the device by vrt_device.hip and for the host by tests/native/traverse_host.cpp (unit tests
// of this very code against the oracle on millions of rays; the shipped library has no host render path).
//
// ---- exact jumps -------------------------------------------------------------------------------------
// The shader advances sideDist by repeated fp32 addition (frag:167), so after k steps along an axis
// sideDist = s (+) d (+) d ... (k roundings), which is not s + k*d.  Within one binade of s, however,
// RN(s + d) = s + Q*ulp(s) for a constant integer Q (d rounded to the ulp of s; a half-ulp tie rounds to
// even and, once the mantissa is even, keeps it even), and the bit pattern of a positive float is monotone
// in its value.  Hence, on the bit patterns, k additions are ONE integer multiply-add, S(j) = bits + j*Q,
// valid until the mantissa would overflow into the next binade.  The DDA's interleaving of the three
// axes is a merge of three such arithmetic sequences by value, ties stepping together (frag:164), so the
// state after any number of iterations is computable in closed form:
//   n_a  = steps axis a may take before leaving the empty cell (or the binade), T_a = S_a(n_a - 1)
//   T*   = min T_a: the sideDist value at which the last iteration of the jump happens
//   c_a  = n_a for the axes with T_a == T*, else #{j : S_a(j) <= T*} = floor((T* - bits_a)/Q_a) + 1
//   mask = axes whose last step was taken exactly at T*
// The final addition of every axis that reaches T* is performed in fp32, so binade crossings are literal.
// The iteration budget (MAX_RAY_STEPS) is tracked as bounds: every iteration steps 1..3 axes, so
// max_a(steps_a) <= iterations <= sum_a(steps_a); a hit whose bounds straddle the budget (only possible
// for rays with exact ties that run within a few steps of the budget) is re-traced literally.
#pragma once

#include "vrt_spec.h"

#ifndef VRT_TRAVERSAL_DENSE
#define VRT_TRAVERSAL_DENSE 1
#define VRT_TRAVERSAL_BITMASK 2
#define VRT_TRAVERSAL_JUMP 3
#define VRT_TRAVERSAL_DF 4
#define VRT_TRAVERSAL_DFJ 5
#endif
#define VRT_TRAVERSAL_BRICK 6     // brick scenes (vrt_scene_from_bricks): DF over a two-level clearance; chosen by AUTO
#define VRT_TRAVERSAL_DF_FAST 7   // internal: DF through the hand-written look-up loop (trace_df_fast); chosen by the host
#define VRT_TRAVERSAL_DF_FAST_CNT 8   // internal: the same through the loops' counting twins (VRT_FLAG_MARCHED_COUNTS / VRT_FLAG_LOOKUP_COUNTS)
#define VRT_TRAVERSAL_BRICK_CNT 9     // internal: the brick march with its counters (the same flags, and every launch that fills iteration-count planes)

namespace vrt {

// Read-only view of a scene's voxel data (device pointers on the device, host pointers in the tests).
struct VolumeView {
    const uint8_t*  vox;     // W*H*D, x + y*W + z*W*H
    const uint64_t* occ1;    // per 4^3 voxels, bit (x&3)|(y&3)<<2|(z&3)<<4
    const uint64_t* occ2;    // per 16^3
    const uint64_t* occ3;    // per 64^3
    const uint8_t*  df;      // 8 octant clearance fields, each x-fastest with a one-voxel border of zeros (df_index): field o
                             // (bit0: +x, bit1: +y, bit2: +z) holds per voxel 0 = solid, else min(63, side of the largest
                             // empty cube that has this voxel as its corner and extends towards the octant's signs;
                             // outside the volume counts as solid)
    uint64_t        df_stride;  // bytes between octant fields
    uint32_t        df_fast;    // 1: the allocation continues with a ninth field, the voxel ids in the same zero-bordered layout
                                // (field 8), and one byte 0xFF at offset 9 * df_stride, and all of it is addressable with
                                // 32-bit offsets (trace_df_fast)
    uint32_t        count_lookups; // 1 (VRT_FLAG_LOOKUP_COUNTS): r.fetches holds the bytes a ray's march asked for instead of its iterations
    uint32_t        count_marched; // 1 (VRT_FLAG_MARCHED_COUNTS): an any-hit ray that is decided a miss without stepping (its clearance covers
                                // what is left of its budget) reports the iterations it TOOK, not the budget the reference's loop would
                                // have spent -- the count planes then hold the product march's own work
    // brick scenes (vrt_scene_from_bricks; vox / occ* / df are null): the volume in 8^3 bricks.  All grids are padded by one
    // brick on every side (index (bx+1) + ((by+1) + (bz+1) * pby) * pbx), the border counting as outside the volume.
    const uint32_t* bgrid;      // 0 = empty brick, 0xFFFFFFFF = border (outside the volume), else 1 + index into bpool / bfine
    const uint8_t*  bcoarse;    // 8 octant fields over the padded grid: 0 = occupied brick or border, else min(16, side in BRICKS of
                                // the largest cube of empty bricks cornered here and extending towards the octant's signs)
    uint64_t        bcoarse_stride;
    const uint8_t*  bpool;      // 512 voxel ids per occupied brick, voxel (x,y,z) of the brick at x + 8y + 64z
    const uint8_t*  bfine;      // per occupied brick 8 octants x 512 voxels: 0 = solid, else min(16, side of the largest empty cube
                                // of VOXELS cornered here ...), looking through the brick's 26 neighbours
    int32_t         pbx, pby;
    const uint64_t* bentry;     // what a look-up of the march reads: ONE 8-byte word per brick of the padded grid (brick_entry_pack):
                                // bits 0..23 the pointer (0 empty, 0xFFFFFF border, else 1 + pool index), bits 24..31 "open" per octant,
                                // bits 32..63 the coarse clearance of the eight octants, four bits each (0 = occupied or border, else
                                // min(15, bricks)) -- bgrid and bcoarse folded into one load instead of two dependent ones
    uint32_t        df_own;      // 1: AO rays through df_any_loop (development switch)
    uint32_t        ao_batch;    // 1: the hand-written loop's kernels trace the AO rays of a wave from a pool in LDS every lane draws on (df_ao_pool_loop; context option "ao_batch")
    uint32_t        df_prefetch; // 1: the secondary rays' look-ups through trace_df_fast prefetch the neighbouring rows (development switch)
    uint32_t        df_thresh;   // 1: primary rays through df_prim_loop (long runs by threshold; launches that report no iteration counts)
    uint32_t        brick_open;  // 1: bit 7 of a coarse byte (no occupied brick is left in the box between this brick and the volume's
                                // corner in the octant's direction: a ray here is a miss) ends the march; 0: the bit is ignored   // padded grid dimensions in x and y
    int32_t W, H, D;
    int32_t n1x, n1y, n1z;
    int32_t n2x, n2y, n2z;
    int32_t n3x, n3y, n3z;
};

struct RayInt {            // RayHitInternal, voxel_volume.frag:33-41
    f3 pos, side, delta;
    int sx, sy, sz;        // rayStep
    int mx, my, mz;        // mapPos at loop exit
    uint32_t material;
    uint32_t mask;         // bit0..2
    uint32_t fetches;      // DENSE/BITMASK: iterations that sampled a voxel (frag:157); JUMP: upper bound
    uint32_t dbg0, dbg1;   // traversal diagnostics (outer iterations / near-regime iterations of trace_skip)
};

struct TraceStats {        // host-side instrumentation (tests); a no-op type is used on the device
    uint32_t literal = 0, jumps1 = 0, jumps2 = 0, jumps3 = 0, retrace = 0, lookups = 0;
};
struct NoStats {};
VRT_HD void st_literal(TraceStats& s) { s.literal++; }
VRT_HD void st_jump(TraceStats& s, int lvl) { if (lvl == 1) s.jumps1++; else if (lvl == 2) s.jumps2++; else s.jumps3++; }
VRT_HD void st_retrace(TraceStats& s) { s.retrace++; }
VRT_HD void st_lookup(TraceStats& s) { s.lookups++; }
VRT_HD void st_literal(NoStats&) {}
VRT_HD void st_jump(NoStats&, int) {}
VRT_HD void st_retrace(NoStats&) {}
VRT_HD void st_lookup(NoStats&) {}

VRT_HD uint32_t f2u(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    union { float f; uint32_t u; } c; c.f = f; return c.u;
#endif
}
VRT_HD float u2f(uint32_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    union { float f; uint32_t u; } c; c.u = u; return c.f;
#endif
}
VRT_HD float rcp_approx(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// ---- occupancy lookups ------------------------------------------------------------------------------

VRT_HD uint32_t cell_bit(int x, int y, int z) { return (uint32_t)(x & 3) | ((uint32_t)(y & 3) << 2) | ((uint32_t)(z & 3) << 4); }

// occ1 word of 4^3 cell (cx,cy,cz); the 16^3 summary is consulted first so empty space costs no global access.
template <class OP>
VRT_HD uint64_t fetch_cell(const VolumeView& v, OP o2, int cx, int cy, int cz)
{
    uint64_t w2 = o2[(cx >> 2) + ((cy >> 2) + (cz >> 2) * v.n2y) * v.n2x];
    if (!((w2 >> cell_bit(cx, cy, cz)) & 1ull)) return 0ull;
    return v.occ1[cx + (cy + cz * v.n1y) * v.n1x];
}

// Emptiness level of the pyramid at voxel (mx,my,mz): 3 = its 64^3 cell is empty, 2 = its 16^3 cell, 1 = its
// 4^3 cell, 0 = the 4^3 cell holds voxels (word = its occ1 bits).
template <class OP>
VRT_HD int lookup_level(const VolumeView& v, OP o2, OP o3, int mx, int my, int mz, uint64_t& word)
{
    int cx = mx >> 2, cy = my >> 2, cz = mz >> 2;
    int qx = cx >> 2, qy = cy >> 2, qz = cz >> 2;
    uint64_t w3 = o3[(qx >> 2) + ((qy >> 2) + (qz >> 2) * v.n3y) * v.n3x];
    word = 0ull;
    if (w3 == 0ull) return 3;
    if (!((w3 >> cell_bit(qx, qy, qz)) & 1ull)) return 2;
    uint64_t w2 = o2[qx + (qy + qz * v.n2y) * v.n2x];
    if (!((w2 >> cell_bit(cx, cy, cz)) & 1ull)) return 1;
    word = v.occ1[cx + (cy + cz * v.n1y) * v.n1x];
    return 0;
}

// Each clearance field is a plain x-fastest volume with a one-voxel border of zeros on every side, (W+2)(H+2)(D+2)
// bytes: voxel (x,y,z) lives at (x+1) + (y+1)*(W+2) + (z+1)*(W+2)*(H+2).  A run can carry a ray at most one voxel
// past a wall (the fields count the outside as solid), so the traversal may read the field wherever a run ends
// without a bounds test, and it keeps the index incrementally (two 24-bit multiply-adds per look-up).  An earlier
// layout in 4x4x4 bricks touched fewer cache lines per gather but cost 12 VALU ops of index arithmetic plus the
// bounds test per look-up; the kernel is bound by VALU issue, not by the vector-memory pipe.
VRT_HD size_t df_index(const VolumeView& v, int x, int y, int z)
{
    const size_t pw = (size_t)v.W + 2u, ph = (size_t)v.H + 2u;
    return (size_t)(x + 1) + ((size_t)(y + 1) + (size_t)(z + 1) * ph) * pw;
}
VRT_HD size_t df_field_bytes(int W, int H, int D)            // one padded field, rounded up to 256 B
{
    size_t n = ((size_t)W + 2u) * ((size_t)H + 2u) * ((size_t)D + 2u);
    return (n + 255u) & ~(size_t)255u;
}
// 32-bit incremental indexing: all eight fields below 4 GiB and a padded z-slice that fits a signed 24-bit multiply
VRT_HD bool df_small(const VolumeView& v)
{
    return 8ull * v.df_stride <= 0xFFFFFFFFull && ((uint64_t)v.W + 2u) * ((uint64_t)v.H + 2u) < (1ull << 23);
}
// trace_df_fast's layout (nine fields + the 0xFF byte, every offset 32 bits): the loop counts its offsets from `bias` =
// (W+2)(H+2) bytes IN FRONT of field 0, so the largest offset it forms is bias + 9 * field (the 0xFF byte), a hit's id read
// reaches bias + 8 * field + index, and a live lane's prefetch one slice (bias bytes) past its own index -- all of it must
// stay below 2^32, and the padded slice must fit the signed 24-bit multiply of the index recovery.
VRT_HD bool df_fast_layout_ok(int W, int H, int D)
{
    const uint64_t pwh = ((uint64_t)W + 2u) * ((uint64_t)H + 2u);
    return 2ull * pwh + 9ull * (uint64_t)df_field_bytes(W, H, D) + 256ull <= 0xFFFFFFFFull && pwh < (1ull << 23);
}
template <bool SMALL> struct IndexT;
template <> struct IndexT<true>  { typedef uint32_t type; typedef int32_t stype; };
template <> struct IndexT<false> { typedef size_t type;   typedef long long stype; };
// a * b for |a|, |b| < 2^23 (one v_mul_i32_i24 / v_mad_i32_i24 instead of the quarter-rate 32-bit multiply)
VRT_HD int mul24(int a, int b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    return a * b;
#endif
}

// ---- boxIntersection + DDA setup (frag:109-144) -------------------------------------------------------

struct DdaState {
    f3 p;                     // boxIntersection() result
    int mx, my, mz;           // mapPos
    float sdx, sdy, sdz;      // sideDist
    float dx, dy, dz;         // deltaDist
    int sx, sy, sz;           // rayStep
    uint32_t mask;            // rule A initial mask
    float ivx, ivy, ivz;      // 1 / dir (dda_entry -> dda_rest)
    float tspan;              // length of the ray inside the box, from where the march starts to where it leaves (0: never inside)
};

// boxIntersection (frag:109-125) and the first mapPos (frag:135): everything needed to know whether the march can
// leave the volume in iteration 0.
VRT_HD void dda_entry(const VolumeView& v, f3 start, f3 dir, DdaState& s)
{
    float ivx = 1.0f / dir.x, ivy = 1.0f / dir.y, ivz = 1.0f / dir.z;
    float t1x = (-start.x) * ivx, t2x = ((float)v.W - start.x) * ivx;
    float t1y = (-start.y) * ivy, t2y = ((float)v.H - start.y) * ivy;
    float t1z = (-start.z) * ivz, t2z = ((float)v.D - start.z) * ivz;
    float tnx = fminf(t1x, t2x), tny = fminf(t1y, t2y), tnz = fminf(t1z, t2z);
    float txx = fmaxf(t1x, t2x), txy = fmaxf(t1y, t2y), txz = fmaxf(t1z, t2z);
    float tmin = fmaxf(tnx, fmaxf(tny, tnz));
    float tmax = fminf(txx, fminf(txy, txz));
    s.p = start;
    s.mask = 0;
    if (tmin >= 0.0f && tmax >= tmin) {
        float t = tmin + 0.1f;
        s.p = mk3(start.x + t * dir.x, start.y + t * dir.y, start.z + t * dir.z);
        s.mask = (uint32_t)(tnx == tmin) | ((uint32_t)(tny == tmin) << 1) | ((uint32_t)(tnz == tmin) << 2);
    }
    s.mx = (int)floorf(s.p.x); s.my = (int)floorf(s.p.y); s.mz = (int)floorf(s.p.z);
    s.ivx = ivx; s.ivy = ivy; s.ivz = ivz;
    const float t0 = fmaxf(tmin, 0.0f);
    s.tspan = tmax >= t0 ? tmax - t0 : 0.0f;
}

// deltaDist, rayStep, sideDist (frag:136-144)
VRT_HD void dda_rest(f3 dir, DdaState& s)
{
    s.dx = fabsf(s.ivx); s.dy = fabsf(s.ivy); s.dz = fabsf(s.ivz);
    float gx = fsign(dir.x), gy = fsign(dir.y), gz = fsign(dir.z);
    s.sx = (int)gx; s.sy = (int)gy; s.sz = (int)gz;
    s.sdx = ((gx * ((float)s.mx - s.p.x) + gx * 0.5f) + 0.5f) * s.dx;
    s.sdy = ((gy * ((float)s.my - s.p.y) + gy * 0.5f) + 0.5f) * s.dy;
    s.sdz = ((gz * ((float)s.mz - s.p.z) + gz * 0.5f) + 0.5f) * s.dz;
}

VRT_HD void dda_setup(const VolumeView& v, f3 start, f3 dir, DdaState& s)
{
    dda_entry(v, start, dir, s);
    dda_rest(dir, s);
}

VRT_HD bool oob(const VolumeView& v, int mx, int my, int mz)
{
    return (uint32_t)mx >= (uint32_t)v.W || (uint32_t)my >= (uint32_t)v.H || (uint32_t)mz >= (uint32_t)v.D;
}

VRT_HD uint32_t voxel_at(const VolumeView& v, int mx, int my, int mz)
{
    return v.vox[(size_t)mx + ((size_t)my + (size_t)mz * (size_t)v.H) * (size_t)v.W];
}

VRT_HD uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { uint32_t m = a < b ? a : b; return m < c ? m : c; }

// One literal DDA iteration's advance (frag:164-170).  sideDist is never negative, so the order of the
// floats is the order of their bit patterns: mask_a = (side_a <= min(side_b, side_c)) == (bits_a == min3(bits)).
// (Integer min/compare need no NaN canonicalisation, which halves the instruction count of this block.)
#define VRT_DDA_STEP(S, MASK)                                                         \
    do {                                                                              \
        uint32_t bx_ = f2u((S).sdx), by_ = f2u((S).sdy), bz_ = f2u((S).sdz);          \
        uint32_t mn_ = umin3(bx_, by_, bz_);                                          \
        bool m0_ = bx_ == mn_, m1_ = by_ == mn_, m2_ = bz_ == mn_;                    \
        (MASK) = (uint32_t)m0_ | ((uint32_t)m1_ << 1) | ((uint32_t)m2_ << 2);         \
        (S).sdx = m0_ ? (S).sdx + (S).dx : (S).sdx; (S).mx += m0_ ? (S).sx : 0;       \
        (S).sdy = m1_ ? (S).sdy + (S).dy : (S).sdy; (S).my += m1_ ? (S).sy : 0;       \
        (S).sdz = m2_ ? (S).sdz + (S).dz : (S).sdz; (S).mz += m2_ ? (S).sz : 0;       \
    } while (0)

VRT_HD void finish(const DdaState& s, uint32_t material, uint32_t mask, uint32_t fetches, RayInt& r)
{
    r.pos = s.p; r.side = mk3(s.sdx, s.sdy, s.sdz); r.delta = mk3(s.dx, s.dy, s.dz);
    r.sx = s.sx; r.sy = s.sy; r.sz = s.sz; r.mx = s.mx; r.my = s.my; r.mz = s.mz;
    r.material = material; r.mask = mask; r.fetches = fetches; r.dbg0 = 0; r.dbg1 = 0;
}

// rint(x) as an int; NaN -> 0 (what v_cvt_i32_f32 does; spelled out for the host build)
VRT_HD int steps_taken(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)rintf(x);
#else
    return x == x ? (int)rintf(x) : 0;
#endif
}

// Signed number of steps an axis took while its sideDist grew by dside: floor(dside * g + 1/2), g = +-1/delta (or 0 for
// an axis that cannot step: its sideDist is +inf, inf - inf = NaN, and the DX9-rule multiply makes NaN * 0 = 0).
// Two VALU ops: v_mul_legacy_f32 + v_cvt_rpi_i32_f32 (round to nearest by floor(x + 0.5) in one instruction).
VRT_HD int steps_signed(float dside, float g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int n;
    float q;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(q) : "v"(dside), "v"(g));
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(n) : "v"(q));
    return n;
#else
    return g == 0.0f ? 0 : (int)floorf(dside * g + 0.5f);
#endif
}

// One DDA iteration that only advances sideDist (frag:164-170 without the mapPos / mask bookkeeping).
// Device: 7 VALU ops -- one three-way integer min, then per axis a v_cmpx that narrows EXEC to the lanes whose axis holds the
// minimum and a v_add_f32 that runs under it (the compiler's form is compare + select + add = 10).  EXEC is put back
// from a scalar copy after each axis; the scalar moves issue beside other waves' vector work.
VRT_HD void dda_advance(DdaState& s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t mn;
    uint64_t saved;
    asm volatile("v_min3_u32 %[mn], %[x], %[y], %[z]\n\t"
                 "s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_eq_u32 %[mn], %[x]\n\t"
                 "v_add_f32 %[x], %[x], %[dx]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_cmpx_eq_u32 %[mn], %[y]\n\t"
                 "v_add_f32 %[y], %[y], %[dy]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_cmpx_eq_u32 %[mn], %[z]\n\t"
                 "v_add_f32 %[z], %[z], %[dz]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [x] "+v"(s.sdx), [y] "+v"(s.sdy), [z] "+v"(s.sdz), [mn] "=&v"(mn), [sv] "=&s"(saved)
                 : [dx] "v"(s.dx), [dy] "v"(s.dy), [dz] "v"(s.dz)
                 : "vcc");
#else
    uint32_t bx = f2u(s.sdx), by = f2u(s.sdy), bz = f2u(s.sdz);
    uint32_t mn = umin3(bx, by, bz);
    s.sdx = bx == mn ? s.sdx + s.dx : s.sdx;
    s.sdy = by == mn ? s.sdy + s.dy : s.sdy;
    s.sdz = bz == mn ? s.sdz + s.dz : s.sdz;
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// The same iteration for use in wave-uniform control flow: only the lanes of `live` advance (EXEC is narrowed to
// live & "this axis holds the minimum" per axis and put back to its value on entry at the end).
__device__ __forceinline__ void dda_advance_live(DdaState& s, uint64_t live)
{
    uint32_t mn;
    uint64_t entry;
    asm volatile("v_min3_u32 %[mn], %[x], %[y], %[z]\n\t"
                 "s_mov_b64 %[en], exec\n\t"
                 "s_mov_b64 exec, %[lv]\n\t"
                 "v_cmpx_eq_u32 %[mn], %[x]\n\t"
                 "v_add_f32 %[x], %[x], %[dx]\n\t"
                 "s_mov_b64 exec, %[lv]\n\t"
                 "v_cmpx_eq_u32 %[mn], %[y]\n\t"
                 "v_add_f32 %[y], %[y], %[dy]\n\t"
                 "s_mov_b64 exec, %[lv]\n\t"
                 "v_cmpx_eq_u32 %[mn], %[z]\n\t"
                 "v_add_f32 %[z], %[z], %[dz]\n\t"
                 "s_mov_b64 exec, %[en]"
                 : [x] "+v"(s.sdx), [y] "+v"(s.sdy), [z] "+v"(s.sdz), [mn] "=&v"(mn), [en] "=&s"(entry)
                 : [dx] "v"(s.dx), [dy] "v"(s.dy), [dz] "v"(s.dz), [lv] "s"(live)
                 : "vcc");
}
// ... and handing out the EXEC mask each v_cmpx leaves behind: it IS that axis' mask bit for the live lanes.
__device__ __forceinline__ void dda_advance_live_masks(DdaState& s, uint64_t live, uint64_t& kx, uint64_t& ky, uint64_t& kz)
{
    uint32_t mn;
    uint64_t entry;
    asm volatile("v_min3_u32 %[mn], %[x], %[y], %[z]\n\t"
                 "s_mov_b64 %[en], exec\n\t"
                 "s_mov_b64 exec, %[lv]\n\t"
                 "v_cmpx_eq_u32 %[mn], %[x]\n\t"
                 "s_mov_b64 %[kx], exec\n\t"
                 "v_add_f32 %[x], %[x], %[dx]\n\t"
                 "s_mov_b64 exec, %[lv]\n\t"
                 "v_cmpx_eq_u32 %[mn], %[y]\n\t"
                 "s_mov_b64 %[ky], exec\n\t"
                 "v_add_f32 %[y], %[y], %[dy]\n\t"
                 "s_mov_b64 exec, %[lv]\n\t"
                 "v_cmpx_eq_u32 %[mn], %[z]\n\t"
                 "s_mov_b64 %[kz], exec\n\t"
                 "v_add_f32 %[z], %[z], %[dz]\n\t"
                 "s_mov_b64 exec, %[en]"
                 : [x] "+v"(s.sdx), [y] "+v"(s.sdy), [z] "+v"(s.sdz), [mn] "=&v"(mn), [en] "=&s"(entry),
                   [kx] "=&s"(kx), [ky] "=&s"(ky), [kz] "=&s"(kz)
                 : [dx] "v"(s.dx), [dy] "v"(s.dy), [dz] "v"(s.dz), [lv] "s"(live)
                 : "vcc");
}
// A whole run of kw >= 1 iterations for the lanes of `live` in one block: kw - 1 iterations whose masks nobody reads, then
// one that hands out its three EXEC masks.  EXEC is saved and put back once per run instead of once per iteration, and the
// counter lives in the block: 5 scalar instructions per iteration (three EXEC reloads, decrement, branch) instead of 8.
// The scalar unit matters: the kernel issues almost as many scalar as vector instructions.
__device__ __forceinline__ void dda_run_live_masks(DdaState& s, uint64_t live, uint32_t kw, uint64_t& kx, uint64_t& ky, uint64_t& kz,
                                                   float& ox, float& oy, float& oz)
{
    uint32_t mn, cnt;
    uint64_t entry;
#define VRT_DDA_ITER                                           \
                 "s_mov_b64 exec, %[lv]\n\t"                    \
                 "v_min3_u32 %[mn], %[x], %[y], %[z]\n\t"       \
                 "v_cmpx_eq_u32 %[mn], %[x]\n\t"                \
                 "v_add_f32 %[x], %[x], %[dx]\n\t"              \
                 "s_mov_b64 exec, %[lv]\n\t"                    \
                 "v_cmpx_eq_u32 %[mn], %[y]\n\t"                \
                 "v_add_f32 %[y], %[y], %[dy]\n\t"              \
                 "s_mov_b64 exec, %[lv]\n\t"                    \
                 "v_cmpx_eq_u32 %[mn], %[z]\n\t"                \
                 "v_add_f32 %[z], %[z], %[dz]\n\t"
    // half of all runs are a single iteration: they take the first branch and nothing else; longer runs do their plain
    // iterations four per loop trip (a taken branch stalls the wave's instruction stream), the odd one, two or three first
    // (the block also keeps sideDist as it was on entry, ox/oy/oz: the compiler's own copies around an in/out operand are
    // three before and two after)
    asm volatile("s_mov_b64 %[en], exec\n\t"
                 "v_mov_b32 %[ox], %[x]\n\t"
                 "v_mov_b32 %[oy], %[y]\n\t"
                 "v_mov_b32 %[oz], %[z]\n\t"
                 "s_cmp_eq_u32 %[kw], 1\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_sub_u32 %[cnt], %[kw], 1\n\t"            // plain iterations, >= 1
                 "s_bitcmp0_b32 %[cnt], 0\n\t"
                 "s_cbranch_scc1 3f\n\t"
                 VRT_DDA_ITER
                 "3:\n\t"
                 "s_bitcmp0_b32 %[cnt], 1\n\t"
                 "s_cbranch_scc1 4f\n\t"
                 VRT_DDA_ITER
                 VRT_DDA_ITER
                 "4:\n\t"
                 "s_lshr_b32 %[cnt], %[cnt], 2\n\t"          // quads; SCC = (quads != 0)
                 "s_cbranch_scc0 2f\n\t"
                 "s_sub_u32 %[cnt], %[cnt], 1\n\t"
                 "1:\n\t"
                 VRT_DDA_ITER
                 VRT_DDA_ITER
                 VRT_DDA_ITER
                 VRT_DDA_ITER
                 "s_sub_u32 %[cnt], %[cnt], 1\n\t"
                 "s_cbranch_scc0 1b\n\t"
                 "2:\n\t"
                 "s_mov_b64 exec, %[lv]\n\t"
                 "v_min3_u32 %[mn], %[x], %[y], %[z]\n\t"
                 "v_cmpx_eq_u32 %[mn], %[x]\n\t"
                 "s_mov_b64 %[kx], exec\n\t"
                 "v_add_f32 %[x], %[x], %[dx]\n\t"
                 "s_mov_b64 exec, %[lv]\n\t"
                 "v_cmpx_eq_u32 %[mn], %[y]\n\t"
                 "s_mov_b64 %[ky], exec\n\t"
                 "v_add_f32 %[y], %[y], %[dy]\n\t"
                 "s_mov_b64 exec, %[lv]\n\t"
                 "v_cmpx_eq_u32 %[mn], %[z]\n\t"
                 "s_mov_b64 %[kz], exec\n\t"
                 "v_add_f32 %[z], %[z], %[dz]\n\t"
                 "s_mov_b64 exec, %[en]"
                 : [x] "+v"(s.sdx), [y] "+v"(s.sdy), [z] "+v"(s.sdz), [mn] "=&v"(mn), [en] "=&s"(entry), [cnt] "=&s"(cnt),
                   [kx] "=&s"(kx), [ky] "=&s"(ky), [kz] "=&s"(kz), [ox] "=&v"(ox), [oy] "=&v"(oy), [oz] "=&v"(oz)
                 : [dx] "v"(s.dx), [dy] "v"(s.dy), [dz] "v"(s.dz), [lv] "s"(live), [kw] "s"(kw)
                 : "vcc", "scc");
#undef VRT_DDA_ITER
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// this lane's bits of three wave-uniform lane masks as 1 | 2 | 4: v_cndmask with the mask as its condition operand
__device__ __forceinline__ uint32_t lane_bits(uint64_t kx, uint64_t ky, uint64_t kz)
{
    uint32_t bx, by, bz;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(bx) : "s"(kx));
    asm("v_cndmask_b32_e64 %0, 0, 2, %1" : "=v"(by) : "s"(ky));
    asm("v_cndmask_b32_e64 %0, 0, 4, %1" : "=v"(bz) : "s"(kz));
    return bx | by | bz;
}
#endif

// ---- wavefront votes (device: the 64 lanes of a gfx950 wave; host tests: a single lane) ------------------

VRT_HD bool wave_all(bool p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __all(p) != 0;
#else
    return p;
#endif
}
VRT_HD bool wave_any(bool p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __any(p) != 0;
#else
    return p;
#endif
}
// min over the active lanes of k (k <= 63), by binary search over ballots: 6 votes, no cross-lane data movement.
VRT_HD uint32_t wave_min_u6(uint32_t k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (__ballot(true) == ~0ull) {
        // all 64 lanes live (the common case): DPP min-scan, total in lane 63.  row_shr:1,2,4,8 fold each row of 16,
        // row_bcast:15 / :31 fold the rows; lanes without a source keep `old` = 63, the identity.
        // v_min_u32 with the DPP modifier on its first source: one VALU op per stage (the builtin form costs three:
        // mov, mov_dpp, min).  A lane whose DPP source does not exist is disabled for that op and keeps its value.
        // s_nop 1 = the two wait states gfx9 needs between a VALU write of a VGPR and a DPP read of it.
        uint32_t v = k, total;
        asm volatile("s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_readlane_b32 %1, %0, 63\n\t"
                     "s_nop 3"
                     : "+v"(v), "=s"(total));
        return total;
    }
    uint32_t m = 0;                                           // partial waves: binary search over ballots
#pragma unroll
    for (uint32_t bit = 32u; bit != 0u; bit >>= 1)
        if (__ballot(k < (m | bit)) == 0ull) m |= bit;
    return m;
#else
    return k;
#endif
}

// One vote for "is every lane finished" and "how far may the wave run": finished lanes vote VRT_VOTE_DONE, live lanes
// their clearance (1..63); the minimum is VRT_VOTE_DONE exactly when nobody is live.
#define VRT_VOTE_DONE 0xFFFFu
VRT_HD uint32_t wave_min_vote(uint32_t k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // Half of all look-ups end with a clearance of 1 somewhere in the wave, another quarter with 2 or 3: one compare
    // each answers those before the 7-op reduction is needed (votes are >= 1; VRT_VOTE_DONE matches none of them).
    if (__ballot(k == 1u) != 0ull) return 1u;
    const bool full = __ballot(true) == ~0ull;
    if (!full) {                                               // secondary rays of a partly hit wave: the reduction below is
        if (__ballot(k == 2u) != 0ull) return 2u;              // the 6-vote binary search, worth two more shortcuts
        if (__ballot(k == 3u) != 0ull) return 3u;
    }
    if (full) {
        uint32_t v = k, total;
        asm volatile("s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_readlane_b32 %1, %0, 63\n\t"
                     "s_nop 3"
                     : "+v"(v), "=s"(total));
        return total;
    }
    if (__ballot(k != VRT_VOTE_DONE) == 0ull) return VRT_VOTE_DONE;
    return wave_min_u6(k < 63u ? k : 63u);
#else
    return k;
#endif
}

// ---- literal traversals ---------------------------------------------------------------------------------

// DF (clearance-field skip, wave-cooperative).  Profiling showed the per-iteration loops are bound by the
// vector-memory pipe and by instruction issue, not by arithmetic: one 64-lane byte gather per DDA iteration
// touches ~24 cache lines on the bench frame and sits on the critical path of every iteration.  Here a set of
// byte volumes holds, per empty voxel and per direction octant, the side k of the largest empty cube that has
// the voxel as its corner and extends towards the octant's signs.  A ray only ever moves towards the signs of
// its direction, at most one voxel per axis per DDA iteration, so its next k-1 iterations stay inside that cube:
// no memory test is needed for them (unlike an isotropic distance field, the clearance of a ray LEAVING a
// surface is large at once).  Voxels outside the volume count as solid, which bounds a run at the walls.  The lanes of a wave agree on the smallest clearance among them (wave_min_vote) and run
// that many iterations of pure ALU stepping -- the same fp32 additions as the shader, hence bit-identical
// results -- then look at memory again.  Neighbouring rays have similar clearances, so the wave-wide minimum
// costs little.  Finished lanes are masked off; the votes see live lanes only.
template <bool SMALL, class STATS, bool AHEAD>
VRT_HD void trace_df_impl(const VolumeView& v, f3 start, f3 dir, uint32_t maxSteps, RayInt& r, STATS& stats)
{
    DdaState s;
    dda_entry(v, start, dir, s);
    // A ray that starts outside the volume and misses it begins the loop at its own origin (frag:118-126) and leaves in
    // iteration 0.  When that is every lane of the wave -- the sky tiles of a frame -- the rest of the set-up, the
    // look-up and the vote have nothing to decide.
    if (wave_all(oob(v, s.mx, s.my, s.mz))) {
        s.dx = s.dy = s.dz = 0.0f; s.sdx = s.sdy = s.sdz = 0.0f; s.sx = s.sy = s.sz = 0;
        finish(s, 0u, s.mask, 0u, r);
        r.dbg0 = 1u; r.dbg1 = 0u;
        return;
    }
    dda_rest(dir, s);
#if defined(__HIP_DEVICE_COMPILE__)
    // deltaDist = |1/dir|: keep the three values in registers of their own (otherwise the |.| is rematerialised as
    // an extra VALU op in every iteration of the stepping loop, whose asm operands cannot take source modifiers)
    asm volatile("" : "+v"(s.dx), "+v"(s.dy), "+v"(s.dz));
#endif
    uint32_t material = 0, fetches = 0;
    // the mask of the latest iteration, one bool per axis: the compiler keeps each as a 64-bit lane mask in scalar
    // registers (the compare results themselves), so recording it costs no vector instructions
    // (device: three wave-wide lane masks, updated by the EXEC masks of the run's last iteration -- scalar moves only)
#if defined(__HIP_DEVICE_COMPILE__)
    // kx/ky/kz: the EXEC masks of the latest run's last iteration (the mask bits of the lanes that were live in it); a lane
    // copies its bits out when it finishes (lmask) -- five vector instructions in the hit path instead of six scalar ones
    // per look-up to keep three merged wave-wide masks up to date
    uint64_t kx = __ballot((s.mask & 1u) != 0u), ky = __ballot((s.mask & 2u) != 0u), kz = __ballot((s.mask & 4u) != 0u);
    uint32_t lmask = s.mask;
#else
    bool k0 = (s.mask & 1u) != 0u, k1 = (s.mask & 2u) != 0u, k2 = (s.mask & 4u) != 0u;
#endif
    // a ray that starts outside the volume and misses it begins the loop at its own origin (frag:118-126) and leaves
    // in iteration 0; every later position is inside the volume or in the one-voxel border of the fields
    bool done = oob(v, s.mx, s.my, s.mz);
    uint32_t clear = 63u;
    // clearance field of the octant the ray travels into (an axis the ray does not move along can use either sign)
    const uint32_t oct = (uint32_t)(s.sx > 0) | ((uint32_t)(s.sy > 0) << 1) | ((uint32_t)(s.sz > 0) << 2);
    // index into the padded field, kept incrementally; IDX is 32 bits while all eight fields stay below 4 GiB
    typedef typename IndexT<SMALL>::type IDX;
    typedef typename IndexT<SMALL>::stype SIDX;
    const int pw = v.W + 2;
    const long long pwh = (long long)pw * (long long)(v.H + 2);
    IDX idx = done ? (IDX)0 : (IDX)((size_t)oct * (size_t)v.df_stride + df_index(v, s.mx, s.my, s.mz));
    const float kInf = u2f(0x7F800000u);
    // signed steps per unit of sideDist: dir is 1/(+-delta) to within an ulp; 0 for an axis that cannot step
    const float gx = s.dx < kInf ? dir.x : 0.0f, gy = s.dy < kInf ? dir.y : 0.0f, gz = s.dz < kInf ? dir.z : 0.0f;
    uint32_t i = 0;                                            // wave-uniform: every live lane has done i iterations
    // look-ups and iterations in runs of >= 12: host instrumentation; on the device only in development builds
    // (make HIPFLAGS+=-DVRT_TRACE_COUNTERS for tools/exp_wavecost.py, exp_skipstats.py) -- four scalar instructions per
    // look-up in a kernel whose scalar unit is not idle
#if !defined(__HIP_DEVICE_COMPILE__) || defined(VRT_TRACE_COUNTERS)
#define VRT_DF_COUNT(x) x
#else
#define VRT_DF_COUNT(x)
#endif
    VRT_DF_COUNT(uint32_t n_outer = 0; uint32_t n_long = 0;)
#if defined(__HIP_DEVICE_COMPILE__)
    // the clearance of the next look-up is requested as soon as its index is known, at the end of the loop body, by every
    // lane (a finished lane re-reads the byte it stopped at): the mask merges, position updates and the loop's scalar
    // bookkeeping then run while the byte is on its way instead of before the request.  AHEAD: primary rays of the
    // primary-only kernel; in the megakernel the extra request per ray costs the short secondary rays more than it hides
    // (+15 % on the reference defaults).
    uint32_t ahead = AHEAD ? v.df[idx] : 0u;
    if (AHEAD && maxSteps == 0u) done = true;                  // (fetches = 0: no iteration runs)
#endif
    for (;;) {
        VRT_DF_COUNT(n_outer++;)
#if defined(__HIP_DEVICE_COMPILE__)
        // Flat form: the budget test is wave-uniform (i is), every lane takes the byte, and only the lanes that found 0
        // enter divergent code -- not a divergent `if (!done)` around everything, whose EXEC bookkeeping is a dozen scalar
        // instructions per look-up.
        // (Primary rays of the primary-only kernel, like the look-ahead request: the megakernel loses 10 % with it.)
        uint32_t vote;
        if (AHEAD) {                                           // i < maxSteps here: the budget is tested where i grows
            clear = ahead;
            st_lookup(stats);
            // the vote value first: a live lane that found 0 is the one lane whose vote is 0, so "did anybody hit" is one
            // compare and a wave-uniform branch that falls through when nobody did (no s_and_saveexec / taken branch /
            // s_or exec around a divergent region in the common case)
            vote = done ? VRT_VOTE_DONE : clear;
            asm volatile("" : "+v"(vote));
            if (__builtin_expect(__ballot(vote == 0u) != 0ull, 0)) {
                if (vote == 0u) {                              // solid, or the border: the ray has left the volume
                    if (oob(v, s.mx, s.my, s.mz)) fetches = i;
                    else {
                        material = SMALL ? v.vox[(uint32_t)s.mx + ((uint32_t)s.my + (uint32_t)s.mz * (uint32_t)v.H) * (uint32_t)v.W]
                                         : voxel_at(v, s.mx, s.my, s.mz);
                        fetches = i + 1u;
                    }
                    done = true;
                    lmask = lane_bits(kx, ky, kz);
                    vote = VRT_VOTE_DONE;
                }
            }
        } else if (!done) {
            if (i >= maxSteps) { done = true; fetches = i; lmask = lane_bits(kx, ky, kz); }
            else {
                clear = v.df[idx];
                st_lookup(stats);
                if (clear == 0u) {
                    if (oob(v, s.mx, s.my, s.mz)) fetches = i;
                    else {
                        material = SMALL ? v.vox[(uint32_t)s.mx + ((uint32_t)s.my + (uint32_t)s.mz * (uint32_t)v.H) * (uint32_t)v.W]
                                         : voxel_at(v, s.mx, s.my, s.mz);
                        fetches = i + 1u;
                    }
                    done = true;
                    lmask = lane_bits(kx, ky, kz);
                }
            }
        }
#else
        if (!done) {
            if (i >= maxSteps) { done = true; fetches = i; }
            else {
                clear = v.df[idx];
                st_lookup(stats);
                if (clear == 0u) {                             // solid, or the border: the ray has left the volume
                    if (oob(v, s.mx, s.my, s.mz)) fetches = i;
                    else {
                        material = SMALL ? v.vox[(uint32_t)s.mx + ((uint32_t)s.my + (uint32_t)s.mz * (uint32_t)v.H) * (uint32_t)v.W]
                                         : voxel_at(v, s.mx, s.my, s.mz);
                        fetches = i + 1u;
                    }
                    done = true;
                }
            }
        }
#endif
        // iterations the wave can take blind: the smallest clearance among its live lanes (the fields count the
        // outside of the volume as solid, so a run cannot carry a lane further than one voxel past a wall); the same
        // vote says whether anybody is still live
#if defined(__HIP_DEVICE_COMPILE__)
        if (!AHEAD) vote = done ? VRT_VOTE_DONE : clear;               // live lanes stand on empty in-bounds voxels: >= 1
        asm volatile("" : "+v"(vote));                                 // (keeps the compare below a compare: see `live`)
#else
        uint32_t vote = done ? VRT_VOTE_DONE : clear;
#endif
        uint32_t kw = wave_min_vote(vote);
        if (kw == VRT_VOTE_DONE) break;
        uint32_t left = maxSteps - i;                          // i < maxSteps for every live lane
        kw = kw < left ? kw : left;
        st_jump(stats, kw > 4u ? 2 : 1);
        VRT_DF_COUNT(if (kw >= 12u) n_long += kw;)
#if defined(__HIP_DEVICE_COMPILE__)
        {
            // The run is wave-uniform control flow: every lane executes it, but the stepping asm narrows EXEC to the live
            // lanes itself, so a finished lane's sideDist stands still, its differences below are 0 and its mapPos /
            // index do not move.  (Inside a divergent `if (!done)` the three lane masks would be per-lane values: six
            // VGPRs and twelve VALU ops per look-up to merge them.)
            const uint64_t live = __ballot(vote != VRT_VOTE_DONE);     // one compare; __ballot(!done) of the bool is two ops + a scalar one
            // Only sideDist is advanced inside the run; mapPos is recovered afterwards: an axis that took n steps has
            // grown by n (+) additions of delta, so n = round((side - side_before) / delta) -- n <= 63 per run and the
            // accumulated rounding error is orders of magnitude below 1/2.
            float ox, oy, oz;
            // kw - 1 iterations whose mask nobody will read, then one whose EXEC masks are the mask bits
            dda_run_live_masks(s, live, kw, kx, ky, kz, ox, oy, oz);
            const int nx = steps_signed(s.sdx - ox, gx), ny = steps_signed(s.sdy - oy, gy), nz = steps_signed(s.sdz - oz, gz);
            idx += SMALL ? (IDX)(nx + mul24(ny, pw) + mul24(nz, (int)pwh))
                         : (IDX)((SIDX)nx + (SIDX)ny * (SIDX)pw + (SIDX)nz * (SIDX)pwh);
            if (AHEAD) ahead = v.df[idx];
            s.mx += nx; s.my += ny; s.mz += nz;
        }
#else
        if (!done) {
            const float ox = s.sdx, oy = s.sdy, oz = s.sdz;
            for (uint32_t j = 1; j < kw; j++) dda_advance(s);
            {
                uint32_t bx = f2u(s.sdx), by = f2u(s.sdy), bz = f2u(s.sdz);
                uint32_t mn = umin3(bx, by, bz);
                k0 = bx == mn; k1 = by == mn; k2 = bz == mn;
                s.sdx = k0 ? s.sdx + s.dx : s.sdx;
                s.sdy = k1 ? s.sdy + s.dy : s.sdy;
                s.sdz = k2 ? s.sdz + s.dz : s.sdz;
            }
            const int nx = steps_signed(s.sdx - ox, gx), ny = steps_signed(s.sdy - oy, gy), nz = steps_signed(s.sdz - oz, gz);
            s.mx += nx; s.my += ny; s.mz += nz;
            idx += SMALL ? (IDX)(nx + mul24(ny, pw) + mul24(nz, (int)pwh))
                         : (IDX)((SIDX)nx + (SIDX)ny * (SIDX)pw + (SIDX)nz * (SIDX)pwh);
        }
#endif
        i += kw;
#if defined(__HIP_DEVICE_COMPILE__)
        if (AHEAD && i >= maxSteps) {                          // wave-uniform: the budget ends for every live lane at once
            if (!done) { fetches = i; lmask = lane_bits(kx, ky, kz); }
            break;
        }
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    finish(s, material, lmask, fetches, r);
#else
    finish(s, material, (uint32_t)k0 | ((uint32_t)k1 << 1) | ((uint32_t)k2 << 2), fetches, r);
#endif
    VRT_DF_COUNT(r.dbg0 = n_outer; r.dbg1 = n_long;)
#undef VRT_DF_COUNT
}

// an AO ray as df_ao_pool_loop takes it up, and the LDS bytes of a wave's pool (11 dwords x 64 columns) + its two rows of counters
struct AoRay { float x, y, z, dx, dy, dz, gx, gy, gz; uint32_t idx0, voxoff; };
#define VRT_AO_SLOT 3840   // (dense scenes: 12 rows of 256 B + two rows of counters; brick scenes: 13 + 2)

#if defined(__HIP_DEVICE_COMPILE__)
// ---- DF, hand-written look-up loop (primary rays of the primary-only kernel) ----------------------------------------------
// Same march, same results as trace_df_impl<true, ., true>; what differs is what it costs to get from one look-up to the
// next.  Measured issue costs on gfx950 at 8 waves per SIMD (tools/ubench/issue_cost.hip; unit: one v_add_f32, about 3
// cycles of a SIMD): two-source vector ops 0.85-1.0, v_min3 / compares into a scalar pair / v_cndmask with a scalar mask /
// conversions / 24-bit mads about 1.5, a DPP step 1.2 (+0.4 for its s_nop), v_cmpx + v_add as a pair 1.7, scalar
// instructions issued between vector ones almost nothing.  So the loop is written for few, cheap VECTOR instructions:
//  * a finished lane does not leave EXEC: its deltas are zeroed (x + 0 = x: its sideDist stands still), its index points
//    at a byte that holds 0xFF, so the byte it reads IS its vote -- no select, no "done" mask, no live mask: every
//    iteration runs under the mask the loop was entered with;
//  * iterations are the EXEC-narrowing form (v_min3, then per axis v_cmpx + v_add under it: 6.9 units; the compare-
//    select-add form without EXEC writes measured 10.0);
//  * one compare (byte < 2) answers both "did a lane hit" and "is the run a single iteration" (half of all look-ups);
//    the DPP minimum runs only when every live lane has two or more iterations to go;
//  * where a ray stands is recovered from how far its sideDist has come since the START of the ray, n = rint(side * g + c)
//    per axis (c = -side0 * g): no copies of sideDist per run, no mapPos; the voxel id at a hit is read from a copy of
//    the volume in the fields' own layout at the same index.  |error of n| <= (n^2 + 3n) 2^-24 (n additions, each rounded to
//    the running sum's ulp), far below 1/2 for the budgets this path accepts (maxSteps <= 1024: 0.063).  An axis the ray
//    cannot step along has side = +inf and g = 0: the product is v_mul_legacy_f32's (0 * anything = 0; a NaN would
//    convert to INT_MIN, tools/ubench/nan_cvt.hip).
// Preconditions (checked by the host): v.df_fast, 1 <= maxSteps <= 1024.  The loop runs under the EXEC mask it is entered with.
// Hazards kept by hand inside the block (gfx950): a DPP source needs two wait states after a vector write, a DPP five
// after a write of EXEC; a scalar pair written by a vector compare needs two before a vector instruction reads it as mask.
// Counting twins of the three look-up loops (template CNT; vrt_render_geometry with VRT_FLAG_MARCHED_COUNTS / _LOOKUP_COUNTS): the same
// instructions plus a per-lane count of the clearance bytes a LIVE lane asks for (the index it loads from is not the 0xFF byte's)
// and of the voxel ids read at the end of a ray -- the bytes the march really requests, which bench.py reports next to the
// iterations it performs.  The product kernels are the CNT = false instantiations: not an instruction more.
#define VRT_CNT_LOOK "v_cmp_ne_u32_e32 vcc, %[sent], v53\n\t" "v_addc_co_u32_e32 %[lk], vcc, 0, %[lk], vcc\n\t"
#define VRT_CNT_FIND "v_add_u32 %[lk], 1, %[lk]\n\t"
#define VRT_CNT_OPND , [lk] "+v"(looks)
#define VRT_CNT_NONE
#define VRT_CNT_NOOP
template <bool CNT>
__device__ __forceinline__ void df_fast_loop(const uint8_t* base, uint32_t maxSteps, int pw, int pwh, uint32_t sentinel,
                                             float& x, float& y, float& z, float dx, float dy, float dz,
                                             float gx, float gy, float gz, float cx, float cy, float cz,
                                             uint32_t idx0, uint32_t voxoff, uint32_t& lmask, uint32_t& material, uint32_t& fetches,
                                             uint64_t kx, uint64_t ky, uint64_t kz, int incx, int incy, int incz, uint32_t anyhit, uint32_t pf, uint32_t& looks)
{
    // the same iteration that also moves the index of the lane's voxel: one more vector instruction per axis under the
    // EXEC mask that is there anyway -- cheaper than recovering the position afterwards for runs of up to four iterations
    // (three quarters of all runs)
#define VRT_F_EITER_IDX                                          \
        "s_mov_b64 exec, s[68:69]\n\t"                                  \
        "v_min3_u32 v48, %[x], %[y], %[z]\n\t"                    \
        "v_cmpx_eq_u32 v48, %[x]\n\t"                             \
        "v_add_f32 %[x], %[x], %[dx]\n\t"                         \
        "v_add_u32 v53, v53, %[ix]\n\t"                           \
        "s_mov_b64 exec, s[68:69]\n\t"                                  \
        "v_cmpx_eq_u32 v48, %[y]\n\t"                             \
        "v_add_f32 %[y], %[y], %[dy]\n\t"                         \
        "v_add_u32 v53, v53, %[iy]\n\t"                           \
        "s_mov_b64 exec, s[68:69]\n\t"                                  \
        "v_cmpx_eq_u32 v48, %[z]\n\t"                             \
        "v_add_f32 %[z], %[z], %[dz]\n\t"                         \
        "v_add_u32 v53, v53, %[iz]\n\t"
#define VRT_F_EITER                                              \
        "s_mov_b64 exec, s[68:69]\n\t"                                  \
        "v_min3_u32 v48, %[x], %[y], %[z]\n\t"                    \
        "v_cmpx_eq_u32 v48, %[x]\n\t"                             \
        "v_add_f32 %[x], %[x], %[dx]\n\t"                         \
        "s_mov_b64 exec, s[68:69]\n\t"                                  \
        "v_cmpx_eq_u32 v48, %[y]\n\t"                             \
        "v_add_f32 %[y], %[y], %[dy]\n\t"                         \
        "s_mov_b64 exec, s[68:69]\n\t"                                  \
        "v_cmpx_eq_u32 v48, %[z]\n\t"                             \
        "v_add_f32 %[z], %[z], %[dz]\n\t"
    // Secondary rays (pf != 0): with every look-up the bytes one step further along y and along z are asked for as well and
    // thrown away -- a ray creeping along a surface looks at memory after every iteration, each look a 64-lane gather that
    // comes from beyond the L2, and with a fifth of the pixels hitting there are no other waves to hide that behind; the next
    // cell of a single-iteration run is one of this cell's three neighbours (the one along x shares its cache line), so the
    // next look finds its line in the vector L1.  (Up to one row / slice past the fields' one-voxel border: the fields are
    // allocated with that much room around them and indexed from pwh bytes in front of field 0.)  The look-up's own load is
    // the oldest of the three: label 10 waits for all but the two youngest.
#define VRT_F_PREFETCH                                           \
        "s_cmp_eq_u32 %[pf], 0\n\t"                               \
        "s_cbranch_scc1 7f\n\t"                                   \
        "v_add_u32 v54, v53, %[iy]\n\t"                           \
        "global_load_ubyte v55, v54, %[base]\n\t"                 \
        "v_add_u32 v54, v53, %[iz]\n\t"                           \
        "global_load_ubyte v56, v54, %[base]\n\t"                 \
        "7:\n\t"
    // scalars of the block: s60 = i (iterations done by every live lane), s61 = kw, s62 = left / iterations still to do,
    // s63 = 0xFF, s[66:67] = saved EXEC; vectors: v48..v50 temporaries, v52 = the byte read = the lane's vote,
    // v53 = index of the byte to read next
    // the block starts on an instruction-cache line: where its loops fall relative to the lines is then a property of the block,
    // not of the code the compiler happens to put in front of it (6 % between two builds that differed elsewhere)
#ifndef VRT_LOOP_PAD_N
#define VRT_LOOP_PAD_N 0             // s_nop words after the alignment (development: scan of the block's phase)
#endif
#define VRT_STR2(x) #x
#define VRT_STR(x) VRT_STR2(x)
#define VRT_F_LOOP(CNT_LOOK, CNT_FIND, CNT_OPND) \
    asm volatile( \
        ".p2align 6\n\t" \
        ".fill " VRT_STR(VRT_LOOP_PAD_N) ", 4, 0xBF800000\n\t" \
        "s_mov_b32 s60, 0\n\t" \
        "s_movk_i32 s63, 0xff\n\t" \
 /* the loop runs under the EXEC mask it is entered with (all 64 lanes for primary rays; the hit lanes of a wave for its */ \
 /* secondary rays): no instruction here may write a lane outside it -- the compiler lets the registers of this block */ \
 /* hold live values of the OTHER lanes of a divergent branch (SIOptimizeVGPRLiveRange) */ \
        "s_mov_b64 s[68:69], exec\n\t" \
        "v_mov_b32 v53, %[idx0]\n\t" \
        "global_load_ubyte v52, v53, %[base]\n\t" \
        VRT_F_PREFETCH \
        "10:\n\t" /* ---- look-up: every lane's byte is here ---- */ \
        CNT_LOOK \
        "s_cmp_eq_u32 %[pf], 0\n\t" \
        "s_cbranch_scc1 8f\n\t" \
        "s_waitcnt vmcnt(2)\n\t" \
        "s_branch 11f\n\t" \
        "8:\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        "11:\n\t" \
 /* any-hit rays (AO, shadow): a lane whose clearance covers what is left of its budget will test nothing but empty voxels */ \
 /* until the budget ends -- it is a miss with fetches = maxSteps, decided here without stepping (a 64-iteration AO ray */ \
 /* that starts in the open: by its first look-up; the fields hold clearances up to 127) */ \
        "s_cmp_eq_u32 %[any], 0\n\t" \
        "s_cbranch_scc1 111f\n\t" \
        "s_sub_u32 s62, %[maxs], s60\n\t" \
        "v_cmp_le_u32_e32 vcc, s62, v52\n\t" /* left <= clearance ... */ \
        "v_cmp_ne_u32_e64 s[64:65], s63, v52\n\t" /* ... of a lane that is still live */ \
        "s_and_b64 vcc, vcc, s[64:65]\n\t" \
        "s_cbranch_vccz 111f\n\t" \
        "s_and_saveexec_b64 s[66:67], vcc\n\t" \
        "s_cmp_eq_u32 %[any], 2\n\t" /* (any == 2: report the iterations taken, VolumeView::count_marched) */ \
        "s_cselect_b32 s62, s60, %[maxs]\n\t" \
        "v_mov_b32 %[fet], s62\n\t" \
        "v_mov_b32 %[dx], 0\n\t" \
        "v_mov_b32 %[dy], 0\n\t" \
        "v_mov_b32 %[dz], 0\n\t" \
        "v_mov_b32 %[gx], 0\n\t" \
        "v_mov_b32 %[gy], 0\n\t" \
        "v_mov_b32 %[gz], 0\n\t" \
        "v_mov_b32 %[cx], 0\n\t" \
        "v_mov_b32 %[cy], 0\n\t" \
        "v_mov_b32 %[cz], 0\n\t" \
        "v_mov_b32 %[ix], 0\n\t" \
        "v_mov_b32 %[iy], 0\n\t" \
        "v_mov_b32 %[iz], 0\n\t" \
        "v_mov_b32 %[idx0], %[sent]\n\t" \
        "v_mov_b32 v53, %[sent]\n\t" \
        "v_mov_b32 v52, s63\n\t" \
        "s_mov_b64 exec, s[66:67]\n\t" \
        "s_nop 4\n\t" \
        "111:\n\t" \
        "v_cmp_gt_u32_e32 vcc, 2, v52\n\t" /* 0 (solid / border) or 1 (single iteration) somewhere? */ \
        "s_cbranch_vccnz 15f\n\t" \
        "v_cmp_gt_u32_e32 vcc, 4, v52\n\t" /* 2 or 3 somewhere (and nothing below)? */ \
        "s_cbranch_vccnz 16f\n\t" \
        "s_cmp_eq_u64 s[68:69], -1\n\t" \
        "s_cbranch_scc0 14f\n\t" /* a partly filled wave: no DPP reduction (lane 63 may be off) */ \
        "v_mov_b32 v48, v52\n\t" /* wave minimum of the votes: every live lane has >= 4 */ \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v48, v48, v48 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v48, v48, v48 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v48, v48, v48 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v48, v48, v48 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v48, v48, v48 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v48, v48, v48 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
        "s_sub_u32 s62, %[maxs], s60\n\t" /* left >= 1 */ \
        "s_nop 0\n\t" \
        "v_readlane_b32 s61, v48, 63\n\t" \
        "s_nop 1\n\t" \
        "s_cmp_eq_u32 s61, s63\n\t" \
        "s_cbranch_scc1 40f\n\t" /* nobody is live: done */ \
        "13:\n\t" \
        "s_min_u32 s61, s61, s62\n\t" \
        "s_cmp_le_u32 s61, 4\n\t" \
        "s_cbranch_scc1 18f\n\t" \
 /* ---- a run of kw = s61 >= 5 iterations: plain iterations, then the positions from the sideDist travelled ---- */ \
        "s_add_u32 s60, s60, s61\n\t" /* i += kw */ \
        "s_sub_u32 s62, s61, 1\n\t" /* plain iterations before the one whose masks are kept */ \
        "s_cmp_eq_u32 s62, 0\n\t" \
        "s_cbranch_scc1 34f\n\t" \
        "s_bitcmp0_b32 s62, 0\n\t" \
        "s_cbranch_scc1 31f\n\t" \
        VRT_F_EITER \
        "31:\n\t" \
        "s_bitcmp0_b32 s62, 1\n\t" \
        "s_cbranch_scc1 32f\n\t" \
        VRT_F_EITER \
        VRT_F_EITER \
        "32:\n\t" \
        "s_lshr_b32 s62, s62, 2\n\t" /* quads; SCC = (quads != 0) */ \
        "s_cbranch_scc0 34f\n\t" \
        "s_sub_u32 s62, s62, 1\n\t" \
        "33:\n\t" \
        VRT_F_EITER \
        VRT_F_EITER \
        VRT_F_EITER \
        VRT_F_EITER \
        "s_sub_u32 s62, s62, 1\n\t" /* SCC = borrow: that was the last quad */ \
        "s_cbranch_scc0 33b\n\t" \
        "34:\n\t" /* the run's last iteration: its EXEC masks are the mask bits */ \
        "s_mov_b64 exec, s[68:69]\n\t" \
        "v_min3_u32 v48, %[x], %[y], %[z]\n\t" \
        "v_cmpx_eq_u32 v48, %[x]\n\t" \
        "s_mov_b64 %[kx], exec\n\t" \
        "v_add_f32 %[x], %[x], %[dx]\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        "v_cmpx_eq_u32 v48, %[y]\n\t" \
        "s_mov_b64 %[ky], exec\n\t" \
        "v_add_f32 %[y], %[y], %[dy]\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        "v_cmpx_eq_u32 v48, %[z]\n\t" \
        "s_mov_b64 %[kz], exec\n\t" \
        "v_add_f32 %[z], %[z], %[dz]\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
 /* ---- where is every lane now?  request its next byte ---- */ \
        "v_mul_legacy_f32 v48, %[x], %[gx]\n\t" /* (legacy: inf * 0 = 0, an axis the ray cannot step along) */ \
        "v_mul_legacy_f32 v49, %[y], %[gy]\n\t" \
        "v_mul_legacy_f32 v50, %[z], %[gz]\n\t" \
        "v_add_f32 v48, v48, %[cx]\n\t" \
        "v_add_f32 v49, v49, %[cy]\n\t" \
        "v_add_f32 v50, v50, %[cz]\n\t" \
        "v_cvt_rpi_i32_f32 v48, v48\n\t" \
        "v_cvt_rpi_i32_f32 v49, v49\n\t" \
        "v_cvt_rpi_i32_f32 v50, v50\n\t" \
        "v_mad_i32_i24 v48, v49, %[pw], v48\n\t" \
        "v_mad_i32_i24 v48, v50, %[pwh], v48\n\t" \
        "v_add_u32 v53, %[idx0], v48\n\t" \
        "global_load_ubyte v52, v53, %[base]\n\t" \
        VRT_F_PREFETCH \
        "s_cmp_lt_u32 s60, %[maxs]\n\t" \
        "s_cbranch_scc1 10b\n\t" \
 /* ---- the budget is spent: the lanes that are still live (their start index is not the 0xFF byte's) stop here ---- */ \
        "19:\n\t" \
        "v_cmp_ne_u32_e32 vcc, %[sent], %[idx0]\n\t" \
        "s_and_saveexec_b64 s[66:67], vcc\n\t" \
        "v_cndmask_b32_e64 v48, 0, 1, %[kx]\n\t" \
        "v_cndmask_b32_e64 v49, 0, 2, %[ky]\n\t" \
        "v_cndmask_b32_e64 v50, 0, 4, %[kz]\n\t" \
        "v_or3_b32 %[lm], v48, v49, v50\n\t" \
        "v_mov_b32 %[fet], s60\n\t" \
        "s_mov_b64 exec, s[66:67]\n\t" \
        "s_branch 40f\n\t" \
        "14:\n\t" /* ---- minimum of a partly filled wave: binary search by votes ---- */ \
        "v_cmp_ne_u32_e32 vcc, s63, v52\n\t" \
        "s_cbranch_vccz 40f\n\t" /* nobody is live: done */ \
        "v_min_u32 v48, 63, v52\n\t" /* (a finished lane: 63, never below a live one) */ \
        "s_sub_u32 s62, %[maxs], s60\n\t" \
        "s_mov_b32 s61, 0\n\t" \
        "s_or_b32 s66, s61, 32\n\t" \
        "v_cmp_gt_u32_e32 vcc, s66, v48\n\t" \
        "s_cmp_eq_u64 vcc, 0\n\t" \
        "s_cselect_b32 s61, s66, s61\n\t" \
        "s_or_b32 s66, s61, 16\n\t" \
        "v_cmp_gt_u32_e32 vcc, s66, v48\n\t" \
        "s_cmp_eq_u64 vcc, 0\n\t" \
        "s_cselect_b32 s61, s66, s61\n\t" \
        "s_or_b32 s66, s61, 8\n\t" \
        "v_cmp_gt_u32_e32 vcc, s66, v48\n\t" \
        "s_cmp_eq_u64 vcc, 0\n\t" \
        "s_cselect_b32 s61, s66, s61\n\t" \
        "s_or_b32 s66, s61, 4\n\t" \
        "v_cmp_gt_u32_e32 vcc, s66, v48\n\t" \
        "s_cmp_eq_u64 vcc, 0\n\t" \
        "s_cselect_b32 s61, s66, s61\n\t" \
        "s_or_b32 s66, s61, 2\n\t" \
        "v_cmp_gt_u32_e32 vcc, s66, v48\n\t" \
        "s_cmp_eq_u64 vcc, 0\n\t" \
        "s_cselect_b32 s61, s66, s61\n\t" \
        "s_or_b32 s66, s61, 1\n\t" \
        "v_cmp_gt_u32_e32 vcc, s66, v48\n\t" \
        "s_cmp_eq_u64 vcc, 0\n\t" \
        "s_cselect_b32 s61, s66, s61\n\t" \
        "s_branch 13b\n\t" \
        "16:\n\t" /* ---- the smallest vote is 2 or 3 ---- */ \
        "v_cmp_eq_u32_e32 vcc, 2, v52\n\t" \
        "s_sub_u32 s62, %[maxs], s60\n\t" \
        "s_mov_b32 s61, 3\n\t" \
        "s_cbranch_vccz 17f\n\t" \
        "s_mov_b32 s61, 2\n\t" \
        "17:\n\t" \
        "s_min_u32 s61, s61, s62\n\t" \
        "18:\n\t" /* ---- a run of kw = s61 in 1..4 iterations, index moved along ---- */ \
        "s_add_u32 s60, s60, s61\n\t" \
        "s_cmp_eq_u32 s61, 1\n\t" \
        "s_cbranch_scc1 184f\n\t" \
        "s_cmp_eq_u32 s61, 2\n\t" \
        "s_cbranch_scc1 183f\n\t" \
        "s_cmp_eq_u32 s61, 3\n\t" \
        "s_cbranch_scc1 182f\n\t" \
        VRT_F_EITER_IDX \
        "182:\n\t" \
        VRT_F_EITER_IDX \
        "183:\n\t" \
        VRT_F_EITER_IDX \
        "184:\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        "v_min3_u32 v48, %[x], %[y], %[z]\n\t" \
        "v_cmpx_eq_u32 v48, %[x]\n\t" \
        "s_mov_b64 %[kx], exec\n\t" \
        "v_add_f32 %[x], %[x], %[dx]\n\t" \
        "v_add_u32 v53, v53, %[ix]\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        "v_cmpx_eq_u32 v48, %[y]\n\t" \
        "s_mov_b64 %[ky], exec\n\t" \
        "v_add_f32 %[y], %[y], %[dy]\n\t" \
        "v_add_u32 v53, v53, %[iy]\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        "v_cmpx_eq_u32 v48, %[z]\n\t" \
        "s_mov_b64 %[kz], exec\n\t" \
        "v_add_f32 %[z], %[z], %[dz]\n\t" \
        "v_add_u32 v53, v53, %[iz]\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        "global_load_ubyte v52, v53, %[base]\n\t" \
        VRT_F_PREFETCH \
        "s_cmp_lt_u32 s60, %[maxs]\n\t" \
        "s_cbranch_scc1 10b\n\t" \
        "s_branch 19b\n\t" \
        "15:\n\t" /* ---- some lane read 0 or 1 ---- */ \
        "v_cmp_eq_u32_e32 vcc, 0, v52\n\t" \
        "s_mov_b32 s61, 1\n\t" \
        "s_cbranch_vccz 18b\n\t" /* only 1s: a single-iteration run */ \
        "s_and_saveexec_b64 s[66:67], vcc\n\t" /* lanes that read 0: a solid voxel, or the border */ \
        CNT_FIND \
        "v_add_u32 v48, v53, %[voxoff]\n\t" \
        "global_load_ubyte %[mat], v48, %[base]\n\t" /* the voxel id (0 in the border: the ray has left the volume) */ \
        "v_cndmask_b32_e64 v48, 0, 1, %[kx]\n\t" \
        "v_cndmask_b32_e64 v49, 0, 2, %[ky]\n\t" \
        "v_cndmask_b32_e64 v50, 0, 4, %[kz]\n\t" \
        "v_or3_b32 %[lm], v48, v49, v50\n\t" \
        "v_mov_b32 %[fet], s60\n\t" \
        "v_mov_b32 %[dx], 0\n\t" \
        "v_mov_b32 %[dy], 0\n\t" \
        "v_mov_b32 %[dz], 0\n\t" \
        "v_mov_b32 %[gx], 0\n\t" \
        "v_mov_b32 %[gy], 0\n\t" \
        "v_mov_b32 %[gz], 0\n\t" \
        "v_mov_b32 %[cx], 0\n\t" \
        "v_mov_b32 %[cy], 0\n\t" \
        "v_mov_b32 %[cz], 0\n\t" \
        "v_mov_b32 %[ix], 0\n\t" \
        "v_mov_b32 %[iy], 0\n\t" \
        "v_mov_b32 %[iz], 0\n\t" \
        "v_mov_b32 %[idx0], %[sent]\n\t" \
        "v_mov_b32 v53, %[sent]\n\t" \
        "v_mov_b32 v52, s63\n\t" \
        "s_mov_b64 exec, s[66:67]\n\t" \
        "s_nop 4\n\t" /* EXEC written -> DPP: five wait states */ \
        "s_branch 11b\n\t" /* the other lanes' bytes are still to be looked at */ \
        "40:\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [dx] "+v"(dx), [dy] "+v"(dy), [dz] "+v"(dz), \
          [gx] "+v"(gx), [gy] "+v"(gy), [gz] "+v"(gz), [cx] "+v"(cx), [cy] "+v"(cy), [cz] "+v"(cz), \
          [idx0] "+v"(idx0), [lm] "+v"(lmask), [mat] "+v"(material), [fet] "+v"(fetches), \
          [kx] "+s"(kx), [ky] "+s"(ky), [kz] "+s"(kz), [ix] "+v"(incx), [iy] "+v"(incy), [iz] "+v"(incz) CNT_OPND \
        : [voxoff] "v"(voxoff), [base] "s"(base), [maxs] "s"(maxSteps), [pw] "s"(pw), [pwh] "s"(pwh), [sent] "s"(sentinel), [any] "s"(anyhit), [pf] "s"(pf) \
        : "vcc", "scc", "memory", "v48", "v49", "v50", "v52", "v53", "v54", "v55", "v56", \
          "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69")
    if (CNT) { VRT_F_LOOP(VRT_CNT_LOOK, VRT_CNT_FIND, VRT_CNT_OPND); }
    else { VRT_F_LOOP(VRT_CNT_NONE, VRT_CNT_NONE, VRT_CNT_NOOP); }
#undef VRT_F_EITER
#undef VRT_F_EITER_IDX
#undef VRT_F_PREFETCH
}

// ---- the same march for PRIMARY rays: long runs by threshold, every lane its own clearance ------------------------------------
// sideDist of an axis only ever grows by that axis' own additions, x (+) dx (+) dx ..., whatever the other two axes do: the three
// axes are three independent sequences, and the shader's loop is their MERGE -- every iteration takes the smallest value (equal
// values together, frag:164).  So "the state after every event with a value <= T has been processed" is a state the loop passes
// through, for ANY T, and it can be reached without merging: step each axis on its own while its sideDist is <= T.  One
// compare and one addition per STEP (v_cmpx narrows EXEC monotonically: a lane that has passed T stays out, nothing to put back
// between steps) where the merged iteration spends seven vector and three scalar instructions on the same addition.  T is the
// lane's own: with clearance c every position at most c - 1 steps away along each axis lies in the empty cube, and
//     T = min over axes of (side + (c - 1) delta), times (1 - 2^-16),
// is below the value at which any axis would take its c-th step (c - 1 <= 126 roundings of 2^-24 each cannot bridge 2^-16), so a
// lane spends ITS OWN clearance and the wave needs no vote on the run length at all.  The position afterwards follows from the
// sideDist travelled, as in df_fast_loop's long runs; the cell a lane lands on lies inside the cube, i.e. it is EMPTY: a solid
// voxel is only ever found after one of the short runs (a vote of 1..4 somewhere in the wave: up to four merged iterations for
// everybody, as before), which leave the mask bits of their last iteration behind.
// What this loop does not know is how many ITERATIONS a lane has taken (two axes that hold the same value step in one): it has
// no budget.  So it is entered only by waves none of whose rays can reach the budget at all (trace_df_fast): a ray takes at most
// one step per integer plane it crosses, i.e. no more than (its length inside the box) x (|dir.x| + |dir.y| + |dir.z|) + 3
// iterations whatever it meets; where that bound stays below maxSteps the budget is dead code, and every other wave takes
// df_fast_loop, which counts.  A ray that finds nothing leaves through the border or an open cell as it always did.  Launches
// that report iteration counts do not come here.
// Hazards as in df_fast_loop.  Runs under the EXEC mask it is entered with.
template <bool CNT>
__device__ __forceinline__ void df_prim_loop(const uint8_t* base, int pw, int pwh, uint32_t sentinel,
                                             float& x, float& y, float& z, float dx, float dy, float dz,
                                             float gx, float gy, float gz, float cx, float cy, float cz,
                                             uint32_t idx0, uint32_t voxoff, uint32_t& lmask, uint32_t& material,
                                             uint64_t kx, uint64_t ky, uint64_t kz, int incx, int incy, int incz, uint32_t& looks)
{
#define VRT_P_EITER_IDX                                          \
        "s_mov_b64 exec, s[68:69]\n\t"                            \
        "v_min3_u32 v48, %[x], %[y], %[z]\n\t"                    \
        "v_cmpx_eq_u32 v48, %[x]\n\t"                             \
        "v_add_f32 %[x], %[x], %[dx]\n\t"                         \
        "v_add_u32 v53, v53, %[ix]\n\t"                           \
        "s_mov_b64 exec, s[68:69]\n\t"                            \
        "v_cmpx_eq_u32 v48, %[y]\n\t"                             \
        "v_add_f32 %[y], %[y], %[dy]\n\t"                         \
        "v_add_u32 v53, v53, %[iy]\n\t"                           \
        "s_mov_b64 exec, s[68:69]\n\t"                            \
        "v_cmpx_eq_u32 v48, %[z]\n\t"                             \
        "v_add_f32 %[z], %[z], %[dz]\n\t"                         \
        "v_add_u32 v53, v53, %[iz]\n\t"
    // one axis of a threshold run: up to 128 steps, four per trip; EXEC only ever narrows, so the trip that empties it ends the axis
#define VRT_P_AXIS(A, DA, L)                                     \
        "s_movk_i32 s61, 31\n\t"                                  \
        L "0:\n\t"                                                \
        "v_cmpx_ge_f32 v54, " A "\n\t"                        \
        "v_add_f32 " A ", " A ", " DA "\n\t"                      \
        "v_cmpx_ge_f32 v54, " A "\n\t"                        \
        "v_add_f32 " A ", " A ", " DA "\n\t"                      \
        "v_cmpx_ge_f32 v54, " A "\n\t"                        \
        "v_add_f32 " A ", " A ", " DA "\n\t"                      \
        "v_cmpx_ge_f32 v54, " A "\n\t"                        \
        "v_add_f32 " A ", " A ", " DA "\n\t"                      \
        "s_cbranch_execz " L "1f\n\t"                             \
        "s_sub_u32 s61, s61, 1\n\t"                               \
        "s_cbranch_scc0 " L "0b\n\t"                              \
        L "1:\n\t"                                                \
        "s_mov_b64 exec, s[68:69]\n\t"
    // scalars of the block: s61 = run length / trip counter, s63 = 0xFF, s[66:67] = saved EXEC, s[68:69] = EXEC on entry;
    // vectors: v48..v50 temporaries, v52 = the byte read (the lane's clearance; 0xFF: finished), v53 = index of the byte to
    // read next, v54 = the lane's threshold
#define VRT_P_LOOP(CNT_LOOK, CNT_FIND, CNT_OPND) \
    asm volatile( \
        ".p2align 6\n\t" \
        "s_movk_i32 s63, 0xff\n\t" \
        "s_mov_b64 s[68:69], exec\n\t" \
        "v_mov_b32 v53, %[idx0]\n\t" \
        "s_mov_b32 s62, 0\n\t" \
        "global_load_ubyte v52, v53, %[base]\n\t" \
        "10:\n\t" /* ---- look-up: every lane's byte is here ---- */ \
        CNT_LOOK \
 /* (every look-up is followed by at least one step of every live lane, and no ray has more than 3 x 1024 steps in it: the */ \
 /* count below only ends a wave whose rays cannot step at all -- direction (0, 0, 0): the shader's loop spins to its budget */ \
 /* and misses, and so does a lane that is still live here) */ \
        "s_add_u32 s62, s62, 1\n\t" \
        "s_cmp_gt_u32 s62, 0x1000\n\t" \
        "s_cbranch_scc1 40f\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        "11:\n\t" \
        "v_cmp_gt_u32_e32 vcc, 2, v52\n\t" /* 0 (solid / border / open) or 1 somewhere? */ \
        "s_cbranch_vccnz 15f\n\t" \
        "v_cmp_gt_u32_e32 vcc, 5, v52\n\t" /* 2, 3 or 4 somewhere (and nothing below)? */ \
        "s_cbranch_vccnz 16f\n\t" \
        "v_cmp_ne_u32_e32 vcc, s63, v52\n\t" /* everybody has >= 5, or is finished -- anybody live at all? */ \
        "s_cbranch_vccz 40f\n\t" \
 /* ---- a threshold run: T = min(side + (c - 1) delta) (1 - 2^-16); a finished lane (delta 0, c - 1 = 254) gets T below */ \
 /* every side and takes no step; an axis that cannot step has side = delta = inf and never holds the minimum ---- */ \
        "v_add_u32 v48, -1, v52\n\t" \
        "v_cvt_f32_u32_e32 v48, v48\n\t" \
        "v_fma_f32 v49, v48, %[dx], %[x]\n\t" \
        "v_fma_f32 v50, v48, %[dy], %[y]\n\t" \
        "v_fma_f32 v48, v48, %[dz], %[z]\n\t" \
        "v_min3_f32 v49, v49, v50, v48\n\t" \
        "v_mul_f32 v54, 0x3f7fff00, v49\n\t" \
        VRT_P_AXIS("%[x]", "%[dx]", "2") \
        VRT_P_AXIS("%[y]", "%[dy]", "3") \
        VRT_P_AXIS("%[z]", "%[dz]", "5") \
 /* ---- where is every lane now?  request its next byte ---- */ \
        "v_mul_legacy_f32 v48, %[x], %[gx]\n\t" /* (legacy: inf * 0 = 0, an axis the ray cannot step along) */ \
        "v_mul_legacy_f32 v49, %[y], %[gy]\n\t" \
        "v_mul_legacy_f32 v50, %[z], %[gz]\n\t" \
        "v_add_f32 v48, v48, %[cx]\n\t" \
        "v_add_f32 v49, v49, %[cy]\n\t" \
        "v_add_f32 v50, v50, %[cz]\n\t" \
        "v_cvt_rpi_i32_f32 v48, v48\n\t" \
        "v_cvt_rpi_i32_f32 v49, v49\n\t" \
        "v_cvt_rpi_i32_f32 v50, v50\n\t" \
        "v_mad_i32_i24 v48, v49, %[pw], v48\n\t" \
        "v_mad_i32_i24 v48, v50, %[pwh], v48\n\t" \
        "v_add_u32 v53, %[idx0], v48\n\t" \
        "global_load_ubyte v52, v53, %[base]\n\t" \
        "s_branch 10b\n\t" \
        "16:\n\t" /* ---- the smallest vote is 2, 3 or 4 ---- */ \
        "v_cmp_eq_u32_e32 vcc, 2, v52\n\t" \
        "s_mov_b32 s61, 2\n\t" \
        "s_cbranch_vccnz 18f\n\t" \
        "v_cmp_eq_u32_e32 vcc, 3, v52\n\t" \
        "s_mov_b32 s61, 3\n\t" \
        "s_cbranch_vccnz 18f\n\t" \
        "s_mov_b32 s61, 4\n\t" \
        "18:\n\t" /* ---- a run of s61 in 1..4 merged iterations, index moved along ---- */ \
        "s_cmp_eq_u32 s61, 1\n\t" \
        "s_cbranch_scc1 184f\n\t" \
        "s_cmp_eq_u32 s61, 2\n\t" \
        "s_cbranch_scc1 183f\n\t" \
        "s_cmp_eq_u32 s61, 3\n\t" \
        "s_cbranch_scc1 182f\n\t" \
        VRT_P_EITER_IDX \
        "182:\n\t" \
        VRT_P_EITER_IDX \
        "183:\n\t" \
        VRT_P_EITER_IDX \
        "184:\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        "v_min3_u32 v48, %[x], %[y], %[z]\n\t" \
        "v_cmpx_eq_u32 v48, %[x]\n\t" \
        "s_mov_b64 %[kx], exec\n\t" \
        "v_add_f32 %[x], %[x], %[dx]\n\t" \
        "v_add_u32 v53, v53, %[ix]\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        "v_cmpx_eq_u32 v48, %[y]\n\t" \
        "s_mov_b64 %[ky], exec\n\t" \
        "v_add_f32 %[y], %[y], %[dy]\n\t" \
        "v_add_u32 v53, v53, %[iy]\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        "v_cmpx_eq_u32 v48, %[z]\n\t" \
        "s_mov_b64 %[kz], exec\n\t" \
        "v_add_f32 %[z], %[z], %[dz]\n\t" \
        "v_add_u32 v53, v53, %[iz]\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        "global_load_ubyte v52, v53, %[base]\n\t" \
        "s_branch 10b\n\t" \
        "15:\n\t" /* ---- some lane read 0 or 1 ---- */ \
        "v_cmp_eq_u32_e32 vcc, 0, v52\n\t" \
        "s_mov_b32 s61, 1\n\t" \
        "s_cbranch_vccz 18b\n\t" /* only 1s: a single-iteration run */ \
        "s_and_saveexec_b64 s[66:67], vcc\n\t" /* lanes that read 0: a solid voxel, the border, or an open cell */ \
        CNT_FIND \
        "v_add_u32 v48, v53, %[voxoff]\n\t" \
        "global_load_ubyte %[mat], v48, %[base]\n\t" /* the voxel id (0 in the border and in an open cell: a miss) */ \
        "v_cndmask_b32_e64 v48, 0, 1, %[kx]\n\t" \
        "v_cndmask_b32_e64 v49, 0, 2, %[ky]\n\t" \
        "v_cndmask_b32_e64 v50, 0, 4, %[kz]\n\t" \
        "v_or3_b32 %[lm], v48, v49, v50\n\t" \
        "v_mov_b32 %[dx], 0\n\t" \
        "v_mov_b32 %[dy], 0\n\t" \
        "v_mov_b32 %[dz], 0\n\t" \
        "v_mov_b32 %[gx], 0\n\t" \
        "v_mov_b32 %[gy], 0\n\t" \
        "v_mov_b32 %[gz], 0\n\t" \
        "v_mov_b32 %[cx], 0\n\t" \
        "v_mov_b32 %[cy], 0\n\t" \
        "v_mov_b32 %[cz], 0\n\t" \
        "v_mov_b32 %[ix], 0\n\t" \
        "v_mov_b32 %[iy], 0\n\t" \
        "v_mov_b32 %[iz], 0\n\t" \
        "v_mov_b32 %[idx0], %[sent]\n\t" \
        "v_mov_b32 v53, %[sent]\n\t" \
        "v_mov_b32 v52, s63\n\t" \
        "s_mov_b64 exec, s[66:67]\n\t" \
        "s_branch 11b\n\t" /* the other lanes' bytes are still to be looked at */ \
        "40:\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [dx] "+v"(dx), [dy] "+v"(dy), [dz] "+v"(dz), \
          [gx] "+v"(gx), [gy] "+v"(gy), [gz] "+v"(gz), [cx] "+v"(cx), [cy] "+v"(cy), [cz] "+v"(cz), \
          [idx0] "+v"(idx0), [lm] "+v"(lmask), [mat] "+v"(material), \
          [kx] "+s"(kx), [ky] "+s"(ky), [kz] "+s"(kz), [ix] "+v"(incx), [iy] "+v"(incy), [iz] "+v"(incz) CNT_OPND \
        : [voxoff] "v"(voxoff), [base] "s"(base), [pw] "s"(pw), [pwh] "s"(pwh), [sent] "s"(sentinel) \
        : "vcc", "scc", "memory", "v48", "v49", "v50", "v52", "v53", "v54", \
          "s61", "s62", "s63", "s66", "s67", "s68", "s69")
    if (CNT) { VRT_P_LOOP(VRT_CNT_LOOK, VRT_CNT_FIND, VRT_CNT_OPND); }
    else { VRT_P_LOOP(VRT_CNT_NONE, VRT_CNT_NONE, VRT_CNT_NOOP); }
#undef VRT_P_AXIS
#undef VRT_P_EITER_IDX
}

// ---- the same march for rays that point every way (AO): every lane spends ITS OWN clearance -------------------------------
// The lanes of a primary or a shadow wave are neighbours going the same way, and the wave-wide minimum of their clearances
// costs little.  The AO rays of a wave point every way from a surface: somebody's cube is always tiny, the minimum is 1 or 2, and
// the wave looks at memory -- a gather of 64 unrelated cache lines -- after every iteration or two.  Here a lane that read
// clearance c takes c iterations of its own while the others take theirs (an iteration runs under the mask of the lanes that
// still have some left: one compare and one subtraction more than the 7 instructions of the plain one), and the wave looks
// again only when everybody has used his up: the number of looks is the number the NEEDIEST lane needs, not the sum over the
// minima.  Positions from the sideDist travelled, as in the long runs of df_fast_loop.  Any-hit rays only: no mask bits, no
// per-lane iteration of a hit to remember -- a lane's own iteration count (fetches) is its i.
// Same fp32 additions per lane, in the same order: bit-identical results.  Runs under the EXEC mask it is entered with.
#ifndef VRT_OWN_CAP
#define VRT_OWN_CAP 6                // iterations of its own a lane may take per look.  Round 3 (df_any_loop, a ray per lane): 4.  With the wave's pool
                                     // (df_ao_pool_loop) a lane that is done does not wait for the others, so longer stretches pay: reference defaults /
                                     // Mandelbulb 4K at 3: 138 us / 1.09 ms, 4: 128 / 1.06, 6: 121 / 1.025, 8: 118 / 1.047, 12: 120 / 1.065, 16: 126 / 1.10
#endif
#ifndef VRT_THRESH_SPREAD
#define VRT_THRESH_SPREAD 1          // brick_march_thresh: a lane's threshold run uses at most this many times the wave's smallest clearance (measured on config 5: 1 -> 3.38 ms, 2 -> 3.46, 4 -> 3.48, no cap -> 13 ms: the others wait for the lane that goes furthest)
#endif
#ifndef VRT_OWN_CAP_BRICK
#define VRT_OWN_CAP_BRICK 8          // ... in the brick march, whose look-ups cost more (trace_brick_own: 5.07 ms at 4, 5.00 at 8)
#endif
template <bool CNT>
__device__ __forceinline__ void df_any_loop(const uint8_t* base, uint32_t maxSteps, int pw, int pwh, uint32_t sentinel,
                                            float& x, float& y, float& z, float dx, float dy, float dz,
                                            float gx, float gy, float gz, float cx, float cy, float cz,
                                            uint32_t idx0, uint32_t voxoff, uint32_t& material, uint32_t& fetches, uint32_t marched, uint32_t& looks)
{
    // vectors of the block: v48..v50 temporaries, v52 the byte read, v53 its index, v54 = i (iterations this lane has taken),
    // v55 = iterations this lane may still take before it has to look again; scalars: s63 = 0xFF, s[64:65] lanes with some left,
    // s[66:67] saved EXEC, s[68:69] EXEC on entry
#define VRT_A_ITER                                               \
        "v_cmp_lt_u32_e32 vcc, 0, v55\n\t"                        \
        "s_cbranch_vccz 30f\n\t"                                  \
        "s_mov_b64 s[64:65], vcc\n\t"                             \
        "s_mov_b64 exec, vcc\n\t"                                 \
        "v_subrev_u32 v55, 1, v55\n\t"                            \
        "v_min3_u32 v48, %[x], %[y], %[z]\n\t"                    \
        "v_cmpx_eq_u32 v48, %[x]\n\t"                             \
        "v_add_f32 %[x], %[x], %[dx]\n\t"                         \
        "s_mov_b64 exec, s[64:65]\n\t"                            \
        "v_cmpx_eq_u32 v48, %[y]\n\t"                             \
        "v_add_f32 %[y], %[y], %[dy]\n\t"                         \
        "s_mov_b64 exec, s[64:65]\n\t"                            \
        "v_cmpx_eq_u32 v48, %[z]\n\t"                             \
        "v_add_f32 %[z], %[z], %[dz]\n\t"                         \
        "s_mov_b64 exec, s[68:69]\n\t"
#define VRT_A_FINISH                                             \
        "v_mov_b32 %[dx], 0\n\t"                                  \
        "v_mov_b32 %[dy], 0\n\t"                                  \
        "v_mov_b32 %[dz], 0\n\t"                                  \
        "v_mov_b32 %[gx], 0\n\t"                                  \
        "v_mov_b32 %[gy], 0\n\t"                                  \
        "v_mov_b32 %[gz], 0\n\t"                                  \
        "v_mov_b32 %[cx], 0\n\t"                                  \
        "v_mov_b32 %[cy], 0\n\t"                                  \
        "v_mov_b32 %[cz], 0\n\t"                                  \
        "v_mov_b32 %[idx0], %[sent]\n\t"                          \
        "v_mov_b32 v53, %[sent]\n\t"                              \
        "v_mov_b32 v52, s63\n\t"
#define VRT_A_LOOP(CNT_LOOK, CNT_FIND, CNT_OPND) \
    asm volatile( \
        ".p2align 6\n\t" \
        "s_movk_i32 s63, 0xff\n\t" \
        "s_mov_b64 s[68:69], exec\n\t" \
        "v_mov_b32 v53, %[idx0]\n\t" \
        "v_mov_b32 v54, 0\n\t" \
        "global_load_ubyte v52, v53, %[base]\n\t" \
        "10:\n\t" /* ---- every lane's byte is here ---- */ \
        CNT_LOOK \
        "s_waitcnt vmcnt(0)\n\t" \
        "v_cmp_eq_u32_e32 vcc, 0, v52\n\t" /* solid, border or open cell: the lane ends here */ \
        "s_cbranch_vccz 12f\n\t" \
        "s_and_saveexec_b64 s[66:67], vcc\n\t" \
        "v_add_u32 v48, v53, %[voxoff]\n\t" \
        CNT_FIND \
        "global_load_ubyte %[mat], v48, %[base]\n\t" /* the voxel id says which */ \
        "v_mov_b32 %[fet], v54\n\t" \
        VRT_A_FINISH \
        "s_mov_b64 exec, s[66:67]\n\t" \
        "12:\n\t" \
        "v_cmp_ne_u32_e64 s[64:65], s63, v52\n\t" /* live lanes ... */ \
        "v_sub_u32 v48, %[maxs], v54\n\t" /* ... what is left of their budget ... */ \
        "s_nop 0\n\t" \
        "v_cmp_le_u32_e32 vcc, v48, v52\n\t" /* ... and whether the clearance covers it: a miss at the budget */ \
        "s_and_b64 vcc, vcc, s[64:65]\n\t" \
        "s_cbranch_vccz 13f\n\t" \
        "s_and_saveexec_b64 s[66:67], vcc\n\t" \
        "v_mov_b32 %[fet], %[maxs]\n\t" \
        "s_cmp_eq_u32 %[mar], 0\n\t" /* (VolumeView::count_marched: the iterations this lane took) */ \
        "s_cbranch_scc1 131f\n\t" \
        "v_mov_b32 %[fet], v54\n\t" \
        "131:\n\t" \
        VRT_A_FINISH \
        "s_mov_b64 exec, s[66:67]\n\t" \
        "13:\n\t" \
        "v_cmp_ne_u32_e32 vcc, s63, v52\n\t" /* who is live now */ \
        "s_cbranch_vccz 40f\n\t" /* nobody: done */ \
        "v_cndmask_b32_e32 v55, 0, v52, vcc\n\t" /* iterations the lane may take: its clearance (0 for a finished lane) */ \
        "v_min_u32 v55, " VRT_STR(VRT_OWN_CAP) ", v55\n\t" /* ... but no more than a few: the others wait for the longest */ \
        "v_add_u32 v54, v54, v55\n\t" /* it will take them all before the next look */ \
        "20:\n\t" /* ---- iterations for the lanes that have some left (four per trip) ---- */ \
        VRT_A_ITER VRT_A_ITER VRT_A_ITER VRT_A_ITER \
        "s_branch 20b\n\t" \
        "30:\n\t" /* ---- where is every lane now?  request its next byte ---- */ \
        "v_mul_legacy_f32 v48, %[x], %[gx]\n\t" \
        "v_mul_legacy_f32 v49, %[y], %[gy]\n\t" \
        "v_mul_legacy_f32 v50, %[z], %[gz]\n\t" \
        "v_add_f32 v48, v48, %[cx]\n\t" \
        "v_add_f32 v49, v49, %[cy]\n\t" \
        "v_add_f32 v50, v50, %[cz]\n\t" \
        "v_cvt_rpi_i32_f32 v48, v48\n\t" \
        "v_cvt_rpi_i32_f32 v49, v49\n\t" \
        "v_cvt_rpi_i32_f32 v50, v50\n\t" \
        "v_mad_i32_i24 v48, v49, %[pw], v48\n\t" \
        "v_mad_i32_i24 v48, v50, %[pwh], v48\n\t" \
        "v_add_u32 v53, %[idx0], v48\n\t" \
        "global_load_ubyte v52, v53, %[base]\n\t" \
        "s_branch 10b\n\t" \
        "40:\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [dx] "+v"(dx), [dy] "+v"(dy), [dz] "+v"(dz), \
          [gx] "+v"(gx), [gy] "+v"(gy), [gz] "+v"(gz), [cx] "+v"(cx), [cy] "+v"(cy), [cz] "+v"(cz), \
          [idx0] "+v"(idx0), [mat] "+v"(material), [fet] "+v"(fetches) CNT_OPND \
        : [voxoff] "v"(voxoff), [base] "s"(base), [maxs] "s"(maxSteps), [pw] "s"(pw), [pwh] "s"(pwh), [sent] "s"(sentinel), [mar] "s"(marched) \
        : "vcc", "scc", "memory", "v48", "v49", "v50", "v52", "v53", "v54", "v55", \
          "s63", "s64", "s65", "s66", "s67", "s68", "s69")
    if (CNT) { VRT_A_LOOP(VRT_CNT_LOOK, VRT_CNT_FIND, VRT_CNT_OPND); }
    else { VRT_A_LOOP(VRT_CNT_NONE, VRT_CNT_NONE, VRT_CNT_NOOP); }
#undef VRT_A_FINISH
#undef VRT_A_ITER
}

// ---- AO rays from a pool the whole wave draws on ---------------------------------------------------------------------------------
// What an AO trace costs is its ROUNDS: every look-up is a gather the next run depends on, and a wave goes round until its
// neediest lane is done.  The AO rays of a wave point every way and need very different numbers of looks -- on the Mandelbulb 7.8
// on average and 30 for the neediest of 64 (a quarter of the lanes of an AO look-up are live: tools/exp_r4_ao_util.py) -- and the
// long rays belong to the same pixels sample after sample (a pixel in a crevice stays in its crevice): letting a lane go on to
// ITS OWN next ray the moment it is done (first attempt of round 4) measured no gain at all.  So the rays are not the lanes':
// every lane writes the ray of its pixel's next AO sample into a pool in LDS (11 dwords: the state trace_df_fast sets up), and a
// lane that is done takes the NEXT RAY OF THE POOL, whoever's it is -- the slot is the wave's counter + the lane's rank among the
// lanes asking in this round (v_mbcnt: no atomic, the wave runs in lock step) -- and reports what it finds to the owner's counter
// in LDS.  When the pool is empty and a lane rests, the loop returns, the lanes write their next sample's rays and the loop goes
// on where it was: the work of a pixel's four AO rays is spread over the wave, and a wave takes about
// max(all looks / lanes, longest single ray) rounds instead of the sum of four maxima.
// Otherwise df_any_loop: every lane spends its ray's own clearance (at most VRT_OWN_CAP iterations per look), the same fp32
// additions per ray in the same order, the iteration count of a ray its own -- which ray a lane works on changes no result.
// A ray that never enters the volume is handed over with zero deltas and an index whose byte is 0 (the first byte of a field:
// its border): it ends at its first look, as a miss, like any other ray ends.
// LDS of a wave (VRT_AO_SLOT bytes at ldsw): the pool, dword q of the ray in slot k at q * 256 + k * 4 (x y z dx dy dz gx gy gz idx0
// voxoff tag; tag = the column of the ray's pixel | the clearance at its first voxel << 8); at 3072 one counter per COLUMN: rays of
// that pixel that found a solid voxel; at 3328 (CNT) what the count planes report for them.  Column = rank of the pixel's lane
// among the lanes the AO phase runs for; slot = where the ray waits (the owners put the rays that will creep in front).
template <bool CNT>
__device__ __forceinline__ void df_ao_pool_loop(const uint8_t* base, uint32_t maxSteps, int pw, int pwh, uint32_t sentinel,
                                                uint32_t ldsw, uint32_t ldsh, uint32_t count, uint32_t more, uint32_t& next,
                                                float& x, float& y, float& z, float& dx, float& dy, float& dz,
                                                float& gx, float& gy, float& gz, float& cx, float& cy, float& cz,
                                                uint32_t& idx0, uint32_t& voxoff, uint32_t& state, uint32_t& owner,
                                                uint32_t marched, uint32_t& looks)
{
    // vectors of the block: v48..v50 temporaries, v52 the byte read (0xFD: the ray ended in this round, 0xFE: a ray just taken up,
    // 0xFF: the lane rests), v53 its index, v54 = i (iterations of the lane's current ray), v55 = iterations it may take before it
    // looks again, v56 = voxel id on its way, v57 = LDS address of the counter of the ray that id belongs to; scalars: s61 temporary,
    // s62 = 0x80, s63 = 0xFF, s[64:65] lanes with some left, s[66:67] / s[70:71] saved EXEC, s[68:69] EXEC on entry
#define VRT_A_ITER                                               \
        "v_cmp_lt_u32_e32 vcc, 0, v55\n\t"                        \
        "s_cbranch_vccz 30f\n\t"                                  \
        "s_mov_b64 s[64:65], vcc\n\t"                             \
        "s_mov_b64 exec, vcc\n\t"                                 \
        "v_subrev_u32 v55, 1, v55\n\t"                            \
        "v_min3_u32 v48, %[x], %[y], %[z]\n\t"                    \
        "v_cmpx_eq_u32 v48, %[x]\n\t"                             \
        "v_add_f32 %[x], %[x], %[dx]\n\t"                         \
        "s_mov_b64 exec, s[64:65]\n\t"                            \
        "v_cmpx_eq_u32 v48, %[y]\n\t"                             \
        "v_add_f32 %[y], %[y], %[dy]\n\t"                         \
        "s_mov_b64 exec, s[64:65]\n\t"                            \
        "v_cmpx_eq_u32 v48, %[z]\n\t"                             \
        "v_add_f32 %[z], %[z], %[dz]\n\t"                         \
        "s_mov_b64 exec, s[68:69]\n\t"
    // the ids asked for when rays read 0 have arrived: a ray that found a solid voxel counts for its owner (L: a label of its own)
#define VRT_Q_COUNT_ID(L, CNT_SOLID)                             \
        "v_cmp_ne_u32_e32 vcc, 0, v56\n\t"                        \
        "s_cbranch_vccz " L "f\n\t"                               \
        "s_and_saveexec_b64 s[66:67], vcc\n\t"                    \
        "v_mov_b32 v48, 1\n\t"                                    \
        "ds_add_u32 v57, v48\n\t"                                 \
        CNT_SOLID                                                 \
        "s_mov_b64 exec, s[66:67]\n\t"                            \
        L ":\n\t"                                                 \
        "v_mov_b32 v56, 0\n\t"
#define VRT_Q_LOOP(CNT_LOOK, CNT_FIND, CNT_MISS, CNT_SOLID, CNT_OPND) \
    asm volatile( \
        ".p2align 6\n\t" \
        "s_movk_i32 s63, 0xff\n\t" \
        "s_movk_i32 s62, 0x80\n\t" \
        "s_mov_b64 s[68:69], exec\n\t" \
        "v_and_b32 v52, 0xff, %[st]\n\t" /* (between calls: the byte in bits 0..7, the iterations of the lane's ray above) */ \
        "v_lshrrev_b32 v54, 8, %[st]\n\t" \
        "v_mov_b32 v53, %[idx0]\n\t" \
        "v_mov_b32 v56, 0\n\t" \
        "s_branch 125f\n\t" /* (the lanes that rest take their rays first) */ \
        "10:\n\t" /* ---- every lane's byte is here, and the ids asked for in the round before ---- */ \
        CNT_LOOK \
        "s_waitcnt vmcnt(0)\n\t" \
        VRT_Q_COUNT_ID("101", CNT_SOLID) \
        "v_cmp_eq_u32_e32 vcc, 0, v52\n\t" /* solid, border or open cell: the ray ends here; the voxel id says which */ \
        "s_cbranch_vccz 12f\n\t" \
        "s_and_saveexec_b64 s[66:67], vcc\n\t" \
        "v_add_u32 v48, v53, %[voxoff]\n\t" \
        "global_load_ubyte v56, v48, %[base]\n\t" \
        "v_lshl_add_u32 v57, %[own], 2, %[ldsh]\n\t" /* the counter of the pixel this ray belongs to */ \
        CNT_FIND \
        "v_mov_b32 v52, 0xfd\n\t" \
        "s_mov_b64 exec, s[66:67]\n\t" \
        "12:\n\t" \
        "v_cmp_gt_u32_e64 s[64:65], s62, v52\n\t" /* lanes with a clearance (1 .. 127) ... */ \
        "v_sub_u32 v48, %[maxs], v54\n\t" /* ... what is left of their ray's budget ... */ \
        "s_nop 0\n\t" \
        "v_cmp_le_u32_e32 vcc, v48, v52\n\t" /* ... and whether the clearance covers it: a miss at the budget */ \
        "s_and_b64 vcc, vcc, s[64:65]\n\t" \
        "s_cbranch_vccz 125f\n\t" \
        "s_and_saveexec_b64 s[66:67], vcc\n\t" \
        CNT_MISS \
        "v_mov_b32 v52, 0xfd\n\t" \
        "s_mov_b64 exec, s[66:67]\n\t" \
        "125:\n\t" /* ---- lanes whose ray ended in this round, and lanes that rest: the next rays of the pool ---- */ \
        "v_cmp_eq_u32_e32 vcc, 0xfd, v52\n\t" \
        "v_cmp_eq_u32_e64 s[64:65], s63, v52\n\t" \
        "s_nop 1\n\t" \
        "s_or_b64 vcc, vcc, s[64:65]\n\t" \
        "s_cbranch_vccz 13f\n\t" \
        "s_mov_b64 s[70:71], vcc\n\t" \
        "s_and_saveexec_b64 s[66:67], vcc\n\t" \
        "v_mbcnt_lo_u32_b32 v48, s70, 0\n\t" \
        "v_mbcnt_hi_u32_b32 v48, s71, v48\n\t" /* the lane's rank among those asking ... */ \
        "v_add_u32 v48, %[nx], v48\n\t" /* ... + the rays taken so far = its ray's column */ \
        "s_bcnt1_i32_b64 s61, s[70:71]\n\t" \
        "v_cmp_gt_u32_e32 vcc, %[cnt], v48\n\t" /* the pool has that many */ \
        "s_add_u32 %[nx], %[nx], s61\n\t" \
        "s_min_u32 %[nx], %[nx], %[cnt]\n\t" \
        "s_and_saveexec_b64 s[70:71], vcc\n\t" /* EXEC = asking and served; s[70:71] = asking */ \
        "s_cbranch_execz 126f\n\t" \
        "v_lshl_add_u32 v49, v48, 2, %[ldsw]\n\t" \
        "ds_read_b32 v50, v49 offset:2816\n\t" /* the ray's pixel (its counters' column) and the clearance its owner read at its start */ \
        "ds_read_b32 %[x], v49\n\t" \
        "ds_read_b32 %[y], v49 offset:256\n\t" \
        "ds_read_b32 %[z], v49 offset:512\n\t" \
        "ds_read_b32 %[dx], v49 offset:768\n\t" \
        "ds_read_b32 %[dy], v49 offset:1024\n\t" \
        "ds_read_b32 %[dz], v49 offset:1280\n\t" \
        "ds_read_b32 %[gx], v49 offset:1536\n\t" \
        "ds_read_b32 %[gy], v49 offset:1792\n\t" \
        "ds_read_b32 %[gz], v49 offset:2048\n\t" \
        "ds_read_b32 %[idx0], v49 offset:2304\n\t" \
        "ds_read_b32 %[voxoff], v49 offset:2560\n\t" \
        "v_mov_b32 v54, 0\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_and_b32 %[own], 0xff, v50\n\t" \
        "v_lshrrev_b32 v52, 8, v50\n\t" /* (1 .. maxSteps - 1: the lane goes on to its iterations in this very round) */ \
        "v_mov_b32 v53, %[idx0]\n\t" \
        "v_mul_legacy_f32 %[cx], %[x], %[gx]\n\t" /* c = -(side0 * g): where the ray stands is rint(side * g + c) steps from its start */ \
        "v_mul_legacy_f32 %[cy], %[y], %[gy]\n\t" \
        "v_mul_legacy_f32 %[cz], %[z], %[gz]\n\t" \
        "v_xor_b32 %[cx], 0x80000000, %[cx]\n\t" \
        "v_xor_b32 %[cy], 0x80000000, %[cy]\n\t" \
        "v_xor_b32 %[cz], 0x80000000, %[cz]\n\t" \
        "126:\n\t" \
        "s_andn2_b64 exec, s[70:71], vcc\n\t" /* asking and not served: the lane rests (zero deltas, the 0xFF byte) */ \
        "s_cbranch_execz 127f\n\t" \
        "v_mov_b32 %[dx], 0\n\t" \
        "v_mov_b32 %[dy], 0\n\t" \
        "v_mov_b32 %[dz], 0\n\t" \
        "v_mov_b32 %[gx], 0\n\t" \
        "v_mov_b32 %[gy], 0\n\t" \
        "v_mov_b32 %[gz], 0\n\t" \
        "v_mov_b32 %[cx], 0\n\t" \
        "v_mov_b32 %[cy], 0\n\t" \
        "v_mov_b32 %[cz], 0\n\t" \
        "v_mov_b32 %[idx0], %[sent]\n\t" \
        "v_mov_b32 v53, %[sent]\n\t" \
        "v_mov_b32 v52, s63\n\t" \
        "127:\n\t" \
        "s_mov_b64 exec, s[66:67]\n\t" \
        "s_cmp_lt_u32 %[nx], %[cnt]\n\t" /* the pool is empty, the pixels have more samples, and a lane rests: back to have it refilled */ \
        "s_cbranch_scc1 13f\n\t" \
        "s_cmp_eq_u32 %[more], 0\n\t" \
        "s_cbranch_scc1 13f\n\t" \
        "v_cmp_eq_u32_e32 vcc, s63, v52\n\t" \
        "s_cbranch_vccnz 40f\n\t" \
        "13:\n\t" \
        "v_cmp_ne_u32_e32 vcc, s63, v52\n\t" /* who is live now */ \
        "s_cbranch_vccz 40f\n\t" /* nobody: done, or the pool wants refilling */ \
        "v_cmp_gt_u32_e32 vcc, s62, v52\n\t" /* iterations a lane may take: its clearance (none for a ray just taken up, or a resting lane) */ \
        "v_cndmask_b32_e32 v55, 0, v52, vcc\n\t" \
        "v_min_u32 v55, " VRT_STR(VRT_OWN_CAP) ", v55\n\t" /* ... but no more than a few: the others wait for the longest */ \
        "v_add_u32 v54, v54, v55\n\t" /* it will take them all before the next look */ \
        "20:\n\t" /* ---- iterations for the lanes that have some left (four per trip) ---- */ \
        VRT_A_ITER VRT_A_ITER VRT_A_ITER VRT_A_ITER \
        "s_branch 20b\n\t" \
        "30:\n\t" /* ---- where is every lane now?  request its next byte ---- */ \
        "v_mul_legacy_f32 v48, %[x], %[gx]\n\t" \
        "v_mul_legacy_f32 v49, %[y], %[gy]\n\t" \
        "v_mul_legacy_f32 v50, %[z], %[gz]\n\t" \
        "v_add_f32 v48, v48, %[cx]\n\t" \
        "v_add_f32 v49, v49, %[cy]\n\t" \
        "v_add_f32 v50, v50, %[cz]\n\t" \
        "v_cvt_rpi_i32_f32 v48, v48\n\t" \
        "v_cvt_rpi_i32_f32 v49, v49\n\t" \
        "v_cvt_rpi_i32_f32 v50, v50\n\t" \
        "v_mad_i32_i24 v48, v49, %[pw], v48\n\t" \
        "v_mad_i32_i24 v48, v50, %[pwh], v48\n\t" \
        "v_add_u32 v53, %[idx0], v48\n\t" \
        "global_load_ubyte v52, v53, %[base]\n\t" \
        "s_branch 10b\n\t" \
        "40:\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        VRT_Q_COUNT_ID("401", CNT_SOLID) \
        "v_lshl_or_b32 %[st], v54, 8, v52\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "s_mov_b64 exec, s[68:69]\n\t" \
        : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [dx] "+v"(dx), [dy] "+v"(dy), [dz] "+v"(dz), \
          [gx] "+v"(gx), [gy] "+v"(gy), [gz] "+v"(gz), [cx] "+v"(cx), [cy] "+v"(cy), [cz] "+v"(cz), \
          [idx0] "+v"(idx0), [voxoff] "+v"(voxoff), [st] "+v"(state), [own] "+v"(owner), [nx] "+s"(next) CNT_OPND \
        : [base] "s"(base), [maxs] "s"(maxSteps), [pw] "s"(pw), [pwh] "s"(pwh), [sent] "s"(sentinel), [mar] "s"(marched), \
          [ldsw] "s"(ldsw), [ldsh] "s"(ldsh), [cnt] "s"(count), [more] "s"(more) \
        : "vcc", "scc", "memory", "v48", "v49", "v50", "v52", "v53", "v54", "v55", "v56", "v57", \
          "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71")
    if (CNT) {
        VRT_Q_LOOP(VRT_CNT_LOOK,
                   "v_add_u32 %[lk], 1, %[lk]\n\t" "ds_add_u32 v57, v54 offset:256\n\t",
                   "v_lshl_add_u32 v49, %[own], 2, %[ldsh]\n\t" "v_mov_b32 v50, %[maxs]\n\t" "s_cmp_eq_u32 %[mar], 0\n\t" "s_cbranch_scc1 141f\n\t" "v_mov_b32 v50, v54\n\t" "141:\n\t" "ds_add_u32 v49, v50 offset:256\n\t",
                   "ds_add_u32 v57, v48 offset:256\n\t",
                   VRT_CNT_OPND);
    } else {
        VRT_Q_LOOP(VRT_CNT_NONE, VRT_CNT_NONE, VRT_CNT_NONE, VRT_CNT_NONE, VRT_CNT_NOOP);
    }
#undef VRT_A_ITER
}

// PF: the look-ups ask for the two neighbouring rows as well (secondary rays: VRT_F_PREFETCH)
// OWN: every lane spends its own clearance (df_any_loop; any-hit rays)
// CNT: the counting twins of the loops (VRT_TRAVERSAL_DF_FAST_CNT): r.fetches = what the march DID for this ray -- the iterations it took
//      (a threshold run's per-axis steps, which is the same number unless two axes tie), or with VolumeView::count_lookups the bytes it
//      asked for (clearance bytes of live lanes, times three where the look-ups prefetch, + the voxel id at the end)
template <class STATS, bool ANYHIT = false, bool PF = false, bool OWN = false, bool CNT = false>
__device__ __forceinline__ void trace_df_fast(const VolumeView& v, f3 start, f3 dir, uint32_t maxSteps, RayInt& r, STATS& stats)
{
    DdaState s;
    dda_entry(v, start, dir, s);
    if (wave_all(oob(v, s.mx, s.my, s.mz))) {
        s.dx = s.dy = s.dz = 0.0f; s.sdx = s.sdy = s.sdz = 0.0f; s.sx = s.sy = s.sz = 0;
        finish(s, 0u, s.mask, 0u, r);
        return;
    }
    dda_rest(dir, s);
    const bool done0 = oob(v, s.mx, s.my, s.mz);
    const uint32_t oct = (uint32_t)(s.sx > 0) | ((uint32_t)(s.sy > 0) << 1) | ((uint32_t)(s.sz > 0) << 2);
    const uint32_t stride = (uint32_t)v.df_stride;
    const int pw = v.W + 2, pwh = pw * (v.H + 2);
    // offsets count from pwh bytes in front of field 0 (room for a prefetch one slice before it: vrt_api.hip df_guard)
    const uint32_t bias = (uint32_t)pwh, octoff = bias + oct * stride, sentinel = bias + 9u * stride;
    const float kInf = u2f(0x7F800000u);
    // a lane that never enters the volume is finished from the start: zero deltas, the 0xFF byte
    float dx = done0 ? 0.0f : s.dx, dy = done0 ? 0.0f : s.dy, dz = done0 ? 0.0f : s.dz;
    float gx = (!done0 && s.dx < kInf) ? dir.x : 0.0f, gy = (!done0 && s.dy < kInf) ? dir.y : 0.0f, gz = (!done0 && s.dz < kInf) ? dir.z : 0.0f;
    float cx = gx != 0.0f ? -(s.sdx * gx) : 0.0f, cy = gy != 0.0f ? -(s.sdy * gy) : 0.0f, cz = gz != 0.0f ? -(s.sdz * gz) : 0.0f;
    const float gx0 = gx, gy0 = gy, gz0 = gz, cx0 = cx, cy0 = cy, cz0 = cz;     // (CNT: the loops zero a finished lane's copies)
    const uint32_t idx0 = done0 ? sentinel : octoff + (uint32_t)df_index(v, s.mx, s.my, s.mz);
    const uint32_t voxoff = 8u * stride - (octoff - bias);
    uint32_t lmask = s.mask, material = 0u, fetches = 0u, looks = 0u;
    const uint64_t kx = __ballot((s.mask & 1u) != 0u), ky = __ballot((s.mask & 2u) != 0u), kz = __ballot((s.mask & 4u) != 0u);
    float x = s.sdx, y = s.sdy, z = s.sdz;
    // what one step along an axis adds to the index (0 for a lane that never enters the volume)
    const int incx = done0 ? 0 : s.sx, incy = done0 ? 0 : s.sy * pw, incz = done0 ? 0 : s.sz * pwh;
    // the block's scalar operands must BE in scalar registers: values that are uniform but were computed under divergent
    // control flow (secondary rays) may live in vector registers
    const uint64_t b64 = (uint64_t)v.df - (uint64_t)bias;
    const uint8_t* base = (const uint8_t*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b64 >> 32)) << 32) |
                                           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64));      // (the builtin returns int)
    // the loop without a budget (df_prim_loop) for a wave none of whose rays can take maxSteps iterations: a ray
    // steps once per integer plane it crosses, at most tspan * (|dir.x| + |dir.y| + |dir.z|) + 3 times (the margin below covers the
    // rounding of the three factors and the march's own deviation from the ideal line)
    bool thresh = false;
    if (!OWN && __builtin_amdgcn_readfirstlane((int)v.df_thresh) != 0) {      // (primary, bounce and shadow rays; the AO rays' 64 iterations ARE their budget)
        const float bound = s.tspan * ((fabsf(dir.x) + fabsf(dir.y)) + fabsf(dir.z)) * 1.001f + 8.0f;
        thresh = __ballot(!done0 && !(bound < (float)maxSteps)) == 0ull;
    }
    bool prefetching = false;
    if (thresh) {
        df_prim_loop<CNT>(base, __builtin_amdgcn_readfirstlane(pw), __builtin_amdgcn_readfirstlane(pwh), (uint32_t)__builtin_amdgcn_readfirstlane((int)sentinel),
                          x, y, z, dx, dy, dz, gx, gy, gz, cx, cy, cz, idx0, voxoff, lmask, material, kx, ky, kz, incx, incy, incz, looks);
        if (CNT) {
            // the iterations of a march that does not count them: the steps each axis has taken, from the sideDist travelled (the
            // position recovery's own formula); two axes that tie step in ONE iteration of the shader's loop and count twice here
            float qx, qy, qz;
            asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(qx) : "v"(x), "v"(gx0));
            asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(qy) : "v"(y), "v"(gy0));
            asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(qz) : "v"(z), "v"(gz0));
            int nx, ny, nz;
            asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(nx) : "v"(qx + cx0));
            asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(ny) : "v"(qy + cy0));
            asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(nz) : "v"(qz + cz0));
            fetches = (uint32_t)((nx < 0 ? -nx : nx) + (ny < 0 ? -ny : ny) + (nz < 0 ? -nz : nz));
        }
    } else if (ANYHIT && OWN && __builtin_amdgcn_readfirstlane((int)v.df_own) != 0) {
        df_any_loop<CNT>(base, (uint32_t)__builtin_amdgcn_readfirstlane((int)maxSteps), __builtin_amdgcn_readfirstlane(pw), __builtin_amdgcn_readfirstlane(pwh),
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)sentinel), x, y, z, dx, dy, dz, gx, gy, gz, cx, cy, cz, idx0, voxoff, material, fetches,
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)v.count_marched), looks);
    } else {
        const uint32_t pf = PF ? (uint32_t)__builtin_amdgcn_readfirstlane((int)v.df_prefetch) : 0u;
        prefetching = pf != 0u;
        df_fast_loop<CNT>(base, (uint32_t)__builtin_amdgcn_readfirstlane((int)maxSteps), __builtin_amdgcn_readfirstlane(pw), __builtin_amdgcn_readfirstlane(pwh),
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)sentinel), x, y, z, dx, dy, dz, gx, gy, gz, cx, cy, cz, idx0, voxoff, lmask, material, fetches, kx, ky, kz,
                          incx, incy, incz, ANYHIT ? 1u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(v.count_marched != 0u)) : 0u, pf, looks);
    }
    s.sdx = x; s.sdy = y; s.sdz = z;
    uint32_t reported = fetches + (material != 0u ? 1u : 0u);
    if (CNT && v.count_lookups != 0u) {
        // bytes asked for: one clearance byte per look-up of a live lane and the voxel id where a lane read 0 (the loops count both
        // in one register); where the look-ups prefetch the two neighbouring rows, three bytes each (the id then counts thrice too:
        // an upper bound, two bytes per ray above the truth)
        reported = done0 ? 0u : (prefetching ? 3u * looks : looks);
    }
    finish(s, material, lmask, reported, r);
    (void)stats; (void)gx0; (void)gy0; (void)gz0; (void)cx0; (void)cy0; (void)cz0;
}

// ---- AO rays through the wave's pool (df_ao_pool_loop) ---------------------------------------------------------------------------
// trace_df_fast's set-up of one ray (frag:109-144 + the index into its octant's clearance field)
__device__ __forceinline__ void ao_ray_setup(const VolumeView& v, f3 start, f3 dir, AoRay& a)
{
    DdaState s;
    dda_entry(v, start, dir, s);
    dda_rest(dir, s);
    const bool done0 = oob(v, s.mx, s.my, s.mz);
    const uint32_t oct = (uint32_t)(s.sx > 0) | ((uint32_t)(s.sy > 0) << 1) | ((uint32_t)(s.sz > 0) << 2);
    const uint32_t stride = (uint32_t)v.df_stride;
    const int pw = v.W + 2, pwh = pw * (v.H + 2);
    const uint32_t bias = (uint32_t)pwh, octoff = bias + oct * stride;
    const float kInf = u2f(0x7F800000u);
    a.x = s.sdx; a.y = s.sdy; a.z = s.sdz;
    a.dx = done0 ? 0.0f : s.dx; a.dy = done0 ? 0.0f : s.dy; a.dz = done0 ? 0.0f : s.dz;
    a.gx = (!done0 && s.dx < kInf) ? dir.x : 0.0f; a.gy = (!done0 && s.dy < kInf) ? dir.y : 0.0f; a.gz = (!done0 && s.dz < kInf) ? dir.z : 0.0f;
    // a ray that never enters the volume: the first byte of its field -- the border, 0 -- ends it at its first look, as a miss
    a.idx0 = done0 ? octoff : octoff + (uint32_t)df_index(v, s.mx, s.my, s.mz);
    a.voxoff = 8u * stride - (octoff - bias);
}

// a ray into column `col` of the wave's pool (ldsw: the byte address of the wave's LDS); tag: the column of its pixel's counters | the
// clearance its owner read at its first voxel << 8
__device__ __forceinline__ void ao_ray_store(uint32_t ldsw, uint32_t col, const AoRay& a, uint32_t tag)
{
    __attribute__((address_space(3))) uint32_t* p = (__attribute__((address_space(3))) uint32_t*)(ldsw + col * 4u);
    p[0 * 64] = f2u(a.x); p[1 * 64] = f2u(a.y); p[2 * 64] = f2u(a.z);
    p[3 * 64] = f2u(a.dx); p[4 * 64] = f2u(a.dy); p[5 * 64] = f2u(a.dz);
    p[6 * 64] = f2u(a.gx); p[7 * 64] = f2u(a.gy); p[8 * 64] = f2u(a.gz);
    p[9 * 64] = a.idx0; p[10 * 64] = a.voxoff; p[11 * 64] = tag;
}

// What a lane of the pool loop carries from one call to the next: the ray it is on (or the resting state it starts in)
struct AoLane { float x, y, z, dx, dy, dz, gx, gy, gz, cx, cy, cz; uint32_t idx0, voxoff, state, owner; };
__device__ __forceinline__ void ao_lane_rest(const VolumeView& v, AoLane& l)
{
    const uint32_t stride = (uint32_t)v.df_stride;
    const int pw = v.W + 2, pwh = pw * (v.H + 2);
    l.x = l.y = l.z = l.dx = l.dy = l.dz = l.gx = l.gy = l.gz = l.cx = l.cy = l.cz = 0.0f;
    l.idx0 = (uint32_t)pwh + 9u * stride; l.voxoff = 0u; l.state = 0xFFu; l.owner = 0u;
}

// One call of the pool loop: `count` rays wait in the pool, `next` of them are taken (in / out), `more`: the pixels have further
// samples, so come back when the pool is empty and a lane rests.  looks (CNT): clearance bytes + ids this LANE asked for.
template <bool CNT>
__device__ __forceinline__ void trace_ao_pool(const VolumeView& v, AoLane& l, uint32_t ldsw, uint32_t count, uint32_t more, uint32_t& next,
                                              uint32_t maxSteps, uint32_t& looks)
{
    const uint32_t stride = (uint32_t)v.df_stride;
    const int pw = v.W + 2, pwh = pw * (v.H + 2);
    const uint32_t bias = (uint32_t)pwh, sentinel = bias + 9u * stride;
    const uint64_t b64 = (uint64_t)v.df - (uint64_t)bias;
    const uint8_t* base = (const uint8_t*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b64 >> 32)) << 32) |
                                           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64));
    const uint32_t lw = (uint32_t)__builtin_amdgcn_readfirstlane((int)ldsw);
    uint32_t nx = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
    df_ao_pool_loop<CNT>(base, (uint32_t)__builtin_amdgcn_readfirstlane((int)maxSteps), __builtin_amdgcn_readfirstlane(pw), __builtin_amdgcn_readfirstlane(pwh),
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)sentinel), lw, lw + 3072u, (uint32_t)__builtin_amdgcn_readfirstlane((int)count),
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)more), nx,
                         l.x, l.y, l.z, l.dx, l.dy, l.dz, l.gx, l.gy, l.gz, l.cx, l.cy, l.cz, l.idx0, l.voxoff, l.state, l.owner,
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)v.count_marched), looks);
    next = nx;
}
#else
// host pass of a .hip file / the host build of the unit tests: parsed, never run (the loop is gfx950 assembly)
template <class STATS, bool ANYHIT = false, bool PF = false, bool OWN = false, bool CNT = false>
VRT_HD void trace_df_fast(const VolumeView&, f3, f3, uint32_t, RayInt&, STATS&) {}
VRT_HD void ao_ray_setup(const VolumeView&, f3, f3, AoRay&) {}
VRT_HD void ao_ray_store(uint32_t, uint32_t, const AoRay&, uint32_t) {}
struct AoLane { float x, y, z, dx, dy, dz, gx, gy, gz, cx, cy, cz; uint32_t idx0, voxoff, state, owner; };
VRT_HD void ao_lane_rest(const VolumeView&, AoLane&) {}
template <bool CNT> VRT_HD void trace_ao_pool(const VolumeView&, AoLane&, uint32_t, uint32_t, uint32_t, uint32_t&, uint32_t, uint32_t&) {}
#endif

template <class STATS, bool AHEAD = false>
VRT_HD void trace_df(const VolumeView& v, f3 start, f3 dir, uint32_t maxSteps, RayInt& r, STATS& stats)
{
    // wave-uniform choice; the 64-bit variant is only reached by volumes whose eight fields total 4 GiB or more (or
    // whose padded z-slice does not fit the 24-bit multiply of the incremental index)
    if (df_small(v)) trace_df_impl<true, STATS, AHEAD>(v, start, dir, maxSteps, r, stats);
    else trace_df_impl<false, STATS, AHEAD>(v, start, dir, maxSteps, r, stats);
}

// ---- BRICK: the DF march over a sparse volume -------------------------------------------------------------------------
// Same loop as DF -- the wave agrees on a number of iterations nobody needs a memory test for, runs them with the shader's
// own fp32 additions, looks again -- with the clearance read from two levels: in an EMPTY brick one byte per brick and
// octant says how many bricks are clear (the voxel's own clearance follows from where it sits in its brick), in an OCCUPIED
// brick one byte per voxel and octant (looking through the neighbouring bricks, up to 16 voxels).  Any lower bound of the
// true clearance gives the same hit (a run only ever skips voxels that are certainly empty), so the result is the dense
// DF's and the oracle's bit for bit; empty space costs one byte per BRICK in memory and traffic.
VRT_HD uint64_t brick_entry_pack(uint32_t ptr, const uint8_t coarse[8])
{
    // ptr: a padded-grid entry (0 empty, 0xFFFFFFFF border, else 1 + pool index < 0xFFFFFF); coarse[o]: the octant's coarse byte
    // (low 7 bits: clearance in bricks, 0 = occupied or border; bit 7: open).  A clearance above 15 is stored as 15: any
    // lower bound of the true clearance gives the same march.
    uint32_t lo = ptr == 0xFFFFFFFFu ? 0xFFFFFFu : (ptr & 0xFFFFFFu), hi = 0u;
    for (int o = 0; o < 8; o++) {
        const uint32_t c = coarse[o] & 0x7Fu;
        hi |= (c > 15u ? 15u : c) << (4 * o);
        if (coarse[o] & 0x80u) lo |= 1u << (24 + o);
    }
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// One look-up of the brick march: the clearance at voxel (mx, my, mz) for a ray of octant `oct` (sx, sy, sz its steps), and the
// voxel's id where that is 0 inside an occupied brick.  One 8-byte load answers for an empty brick, the border and an open
// brick; only a lane inside an OCCUPIED brick goes on to its per-voxel byte.  The arithmetic for the empty-brick answer runs
// for every lane (a dozen instructions, no branch); 32-bit offsets while the padded grid is below 2^29 bricks.
#if defined(VRT_TRACE_COUNTERS) && defined(__HIPCC__)
// development build only: look-ups of the brick march by kind, summed over the lanes of all rays (vrt_debug_counters)
__device__ unsigned long long g_vrt_brick_counts[4];   // lanes looking up, ... in an occupied brick, ... that found a solid voxel, ... in the border / an open brick
#endif
VRT_HD uint32_t brick_clear(const VolumeView& v, int mx, int my, int mz, uint32_t oct, int sx, int sy, int sz, uint32_t& material, uint32_t* bytes = nullptr)
{
    const uint32_t bx = (uint32_t)((mx >> 3) + 1), by = (uint32_t)((my >> 3) + 1), bz = (uint32_t)((mz >> 3) + 1);   // (-1 >> 3 = -1: the border brick)
    const uint32_t bi = bx + (uint32_t)mul24((int)by, v.pbx) + (uint32_t)mul24((int)bz, v.pbx * v.pby);
    const uint64_t e = v.bentry[bi];
    const uint32_t lo = (uint32_t)e, c = ((uint32_t)(e >> 32) >> (oct * 4u)) & 15u, ptr = lo & 0xFFFFFFu;
    const uint32_t lx = (uint32_t)mx & 7u, ly = (uint32_t)my & 7u, lz = (uint32_t)mz & 7u;
    // an empty brick with c - 1 empty bricks behind it on every axis: the room to the brick's far face is (l ^ 7) + 1 looking
    // up an axis, l + 1 looking down
    const uint32_t rx = (lx ^ (sx > 0 ? 7u : 0u)) + 1u, ry = (ly ^ (sy > 0 ? 7u : 0u)) + 1u, rz = (lz ^ (sz > 0 ? 7u : 0u)) + 1u;
    uint32_t k = (c - 1u) * 8u + umin3(rx, ry, rz);
    k = k < 127u ? k : 127u;
    const bool open = v.brick_open != 0u && ((lo >> (24u + oct)) & 1u) != 0u;     // an open brick: the march ends here as a miss
    uint32_t clear = (c != 0u && !open) ? k : 0u;                                  // (c == 0 and no pool brick: the border -- the ray has left)
    if (c == 0u && ptr - 1u < 0xFFFFFEu) {
        const uint32_t l = lx | (ly << 3) | (lz << 6);
        clear = v.bfine[(size_t)(ptr - 1u) * 4096u + (size_t)(oct * 512u + l)];
        if (clear == 0u) material = v.bpool[(size_t)(ptr - 1u) * 512u + (size_t)l];
        if (bytes) *bytes += clear == 0u ? 2u : 1u;
    }
    if (bytes) *bytes += 8u;
#if defined(VRT_TRACE_COUNTERS) && defined(__HIP_DEVICE_COMPILE__)
    {
        const bool occb = c == 0u && ptr - 1u < 0xFFFFFEu;
        const unsigned long long n0 = __builtin_popcountll(__ballot(true)), n1 = __builtin_popcountll(__ballot(occb)),
                                 n2 = __builtin_popcountll(__ballot(occb && clear == 0u)), n3 = __builtin_popcountll(__ballot(!occb && clear == 0u));
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u || __ballot(true) != ~0ull) {
            // one lane of the active set adds for all (the first active lane)
            const unsigned long long act = __ballot(true);
            if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(act)) {
                atomicAdd(&g_vrt_brick_counts[0], n0); atomicAdd(&g_vrt_brick_counts[1], n1);
                atomicAdd(&g_vrt_brick_counts[2], n2); atomicAdd(&g_vrt_brick_counts[3], n3);
            }
        }
    }
#endif
    return clear;
}

#if defined(__HIP_DEVICE_COMPILE__)
// One axis of a threshold run under the EXEC mask it is entered with: x (+)= dx while x <= T, at most 128 times (v_cmpx narrows
// EXEC monotonically -- a lane that has passed T stays out -- so nothing is put back between steps; four steps per trip)
__device__ __forceinline__ void axis_run(float& x, float dx, float T)
{
    uint64_t saved;
    uint32_t cnt;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_movk_i32 %[cnt], 31\n\t"
                 "1:\n\t"
                 "v_cmpx_ge_f32 %[T], %[x]\n\t"
                 "v_add_f32 %[x], %[x], %[dx]\n\t"
                 "v_cmpx_ge_f32 %[T], %[x]\n\t"
                 "v_add_f32 %[x], %[x], %[dx]\n\t"
                 "v_cmpx_ge_f32 %[T], %[x]\n\t"
                 "v_add_f32 %[x], %[x], %[dx]\n\t"
                 "v_cmpx_ge_f32 %[T], %[x]\n\t"
                 "v_add_f32 %[x], %[x], %[dx]\n\t"
                 "s_cbranch_execz 2f\n\t"
                 "s_sub_u32 %[cnt], %[cnt], 1\n\t"
                 "s_cbranch_scc0 1b\n\t"
                 "2:\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [x] "+v"(x), [sv] "=&s"(saved), [cnt] "=&s"(cnt)
                 : [dx] "v"(dx), [T] "v"(T)
                 : "vcc", "scc");
}

// The brick march without a budget, its long runs by threshold (df_prim_loop's scheme in the generic loop): a run of up to four
// iterations for everybody when some lane's clearance is that small (they leave the mask bits behind that a hit needs), else
// every lane steps each axis on its own while its sideDist is <= the lane's threshold min(side + (c - 1) delta) (1 - 2^-16) --
// one compare and one addition per step, every lane its own clearance.  Entered only by waves none of whose rays can take
// maxSteps iterations (trace_brick); primary, bounce and shadow rays.
template <class STATS, bool CNT = false>
__device__ __forceinline__ void brick_march_thresh(const VolumeView& v, DdaState& s, f3 dir, RayInt& r, STATS& stats)
{
    asm volatile("" : "+v"(s.dx), "+v"(s.dy), "+v"(s.dz));
    uint64_t kx = __ballot((s.mask & 1u) != 0u), ky = __ballot((s.mask & 2u) != 0u), kz = __ballot((s.mask & 4u) != 0u);
    uint32_t lmask = s.mask, material = 0u, clear = 63u;
    // (CNT, VRT_TRAVERSAL_BRICK_CNT: the steps the axes take, which is the iteration count unless two of them tie, and the bytes the
    // look-ups ask for; the product kernels are the CNT = false instantiations)
    constexpr bool cnt = CNT;
    uint32_t steps = 0u, bytes = 0u;
    bool done = oob(v, s.mx, s.my, s.mz);
    const uint32_t oct = (uint32_t)(s.sx > 0) | ((uint32_t)(s.sy > 0) << 1) | ((uint32_t)(s.sz > 0) << 2);
    const float kInf = u2f(0x7F800000u);
    const float gx = s.dx < kInf ? dir.x : 0.0f, gy = s.dy < kInf ? dir.y : 0.0f, gz = s.dz < kInf ? dir.z : 0.0f;
    // (every look-up is followed by at least one step of every live lane: the count only ends a wave whose rays cannot step --
    // direction (0, 0, 0): the shader's loop spins to its budget and misses, and so does a lane still live here)
    for (uint32_t guard = 0; guard < 16384u; guard++) {
        if (!done) {
            uint32_t m = 0u;
            clear = brick_clear(v, s.mx, s.my, s.mz, oct, s.sx, s.sy, s.sz, m, cnt ? &bytes : nullptr);
            st_lookup(stats);
            if (clear == 0u) {                                 // solid, the border, or an open brick
                if (!oob(v, s.mx, s.my, s.mz)) material = m;
                done = true;
                lmask = lane_bits(kx, ky, kz);
            }
        }
        uint32_t vote = done ? VRT_VOTE_DONE : clear;
        asm volatile("" : "+v"(vote));
        if (__ballot(vote != VRT_VOTE_DONE) == 0ull) break;
        const float ox = s.sdx, oy = s.sdy, oz = s.sdz;
        if (__ballot(vote < 5u) != 0ull) {
            // a short run for everybody: the smallest clearance, 1..4 iterations, the last one's EXEC masks are the mask bits
            const uint32_t kw = wave_min_vote(vote);
            const uint64_t live = __ballot(vote != VRT_VOTE_DONE);
            float qx, qy, qz;
            dda_run_live_masks(s, live, kw, kx, ky, kz, qx, qy, qz);
        } else {
            // (a lane's own clearance, but no more than VRT_THRESH_SPREAD times the smallest of the wave: the others wait for the
            // lane that goes furthest; a finished lane has T = -1: it takes no step)
            const uint32_t kmin = wave_min_vote(vote), cap = kmin * (uint32_t)VRT_THRESH_SPREAD;
            const float cm1 = (float)((clear < cap ? clear : cap) - 1u);
            float T = fminf(fminf(__builtin_fmaf(cm1, s.dx, s.sdx), __builtin_fmaf(cm1, s.dy, s.sdy)), __builtin_fmaf(cm1, s.dz, s.sdz)) * 0.99998474f;
            if (done) T = -1.0f;
            axis_run(s.sdx, s.dx, T);
            axis_run(s.sdy, s.dy, T);
            axis_run(s.sdz, s.dz, T);
        }
        const int nx = steps_signed(s.sdx - ox, gx), ny = steps_signed(s.sdy - oy, gy), nz = steps_signed(s.sdz - oz, gz);
        s.mx += nx; s.my += ny; s.mz += nz;
        if (cnt) steps += (uint32_t)((nx < 0 ? -nx : nx) + (ny < 0 ? -ny : ny) + (nz < 0 ? -nz : nz));
    }
    finish(s, material, lmask, cnt ? (v.count_lookups != 0u ? bytes : steps + (material != 0u ? 1u : 0u)) : 0u, r);
}
#endif

template <class STATS, bool ANYHIT = false, bool CNT = false>
VRT_HD void trace_brick(const VolumeView& v, f3 start, f3 dir, uint32_t maxSteps, RayInt& r, STATS& stats)
{
    DdaState s;
    dda_entry(v, start, dir, s);
    if (wave_all(oob(v, s.mx, s.my, s.mz))) {
        s.dx = s.dy = s.dz = 0.0f; s.sdx = s.sdy = s.sdz = 0.0f; s.sx = s.sy = s.sz = 0;
        finish(s, 0u, s.mask, 0u, r);
        r.dbg0 = 1u; r.dbg1 = 0u;
        return;
    }
    dda_rest(dir, s);
#if defined(__HIP_DEVICE_COMPILE__)
    // the march without a budget for a wave none of whose rays can take maxSteps iterations (the bound of trace_df_fast)
    if (__builtin_amdgcn_readfirstlane((int)v.df_thresh) != 0) {
        const float bound = s.tspan * ((fabsf(dir.x) + fabsf(dir.y)) + fabsf(dir.z)) * 1.001f + 8.0f;
        if (__ballot(!oob(v, s.mx, s.my, s.mz) && !(bound < (float)maxSteps)) == 0ull) {
            brick_march_thresh<STATS, CNT>(v, s, dir, r, stats);
            return;
        }
    }
    asm volatile("" : "+v"(s.dx), "+v"(s.dy), "+v"(s.dz));
    uint64_t kx = __ballot((s.mask & 1u) != 0u), ky = __ballot((s.mask & 2u) != 0u), kz = __ballot((s.mask & 4u) != 0u);
    uint32_t lmask = s.mask;
#else
    bool k0 = (s.mask & 1u) != 0u, k1 = (s.mask & 2u) != 0u, k2 = (s.mask & 4u) != 0u;
#endif
    uint32_t material = 0, fetches = 0, lk_bytes = 0;
    bool done = oob(v, s.mx, s.my, s.mz);
    uint32_t clear = 63u;
    const uint32_t oct = (uint32_t)(s.sx > 0) | ((uint32_t)(s.sy > 0) << 1) | ((uint32_t)(s.sz > 0) << 2);
    const float kInf = u2f(0x7F800000u);
    const float gx = s.dx < kInf ? dir.x : 0.0f, gy = s.dy < kInf ? dir.y : 0.0f, gz = s.dz < kInf ? dir.z : 0.0f;
    uint32_t i = 0;
#if defined(VRT_TRACE_COUNTERS)
    uint32_t n_look = 0, n_occ = 0;                            // development: look-ups of this lane, and those inside occupied bricks
#endif
    for (;;) {
        if (!done) {
            if (i >= maxSteps) {
                done = true; fetches = i;
#if defined(__HIP_DEVICE_COMPILE__)
                lmask = lane_bits(kx, ky, kz);
#endif
            } else {
                uint32_t m = 0u;
                clear = brick_clear(v, s.mx, s.my, s.mz, oct, s.sx, s.sy, s.sz, m, CNT ? &lk_bytes : nullptr);
                st_lookup(stats);
#if defined(VRT_TRACE_COUNTERS)
                n_look++;
                {
                    const uint32_t bi_ = (uint32_t)((s.mx >> 3) + 1) + (uint32_t)(((s.my >> 3) + 1) * v.pbx) + (uint32_t)(((s.mz >> 3) + 1) * v.pbx * v.pby);
                    const uint64_t e_ = v.bentry[bi_];
                    if ((((uint32_t)(e_ >> 32) >> (oct * 4u)) & 15u) == 0u) n_occ++;
                }
#endif
                if (clear == 0u) {                             // solid, or the border: the ray has left the volume
                    if (oob(v, s.mx, s.my, s.mz)) fetches = i;
                    else { material = m; fetches = i + 1u; }
                    done = true;
#if defined(__HIP_DEVICE_COMPILE__)
                    lmask = lane_bits(kx, ky, kz);
#endif
                } else if (ANYHIT && clear >= maxSteps - i) {
                    // any-hit ray whose clearance covers the rest of its budget: a miss with fetches = maxSteps, no stepping
                    done = true; fetches = v.count_marched ? i : maxSteps;
                }
            }
        }
        uint32_t vote = done ? VRT_VOTE_DONE : clear;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(vote));
#endif
        uint32_t kw = wave_min_vote(vote);
        if (kw == VRT_VOTE_DONE) break;
        uint32_t left = maxSteps - i;
        kw = kw < left ? kw : left;
        st_jump(stats, kw > 4u ? 2 : 1);
#if defined(__HIP_DEVICE_COMPILE__)
        {
            const uint64_t live = __ballot(vote != VRT_VOTE_DONE);
            float ox, oy, oz;
            dda_run_live_masks(s, live, kw, kx, ky, kz, ox, oy, oz);
            s.mx += steps_signed(s.sdx - ox, gx); s.my += steps_signed(s.sdy - oy, gy); s.mz += steps_signed(s.sdz - oz, gz);
        }
#else
        if (!done) {
            const float ox = s.sdx, oy = s.sdy, oz = s.sdz;
            for (uint32_t j = 1; j < kw; j++) dda_advance(s);
            {
                uint32_t bx = f2u(s.sdx), by = f2u(s.sdy), bz = f2u(s.sdz);
                uint32_t mn = umin3(bx, by, bz);
                k0 = bx == mn; k1 = by == mn; k2 = bz == mn;
                s.sdx = k0 ? s.sdx + s.dx : s.sdx;
                s.sdy = k1 ? s.sdy + s.dy : s.sdy;
                s.sdz = k2 ? s.sdz + s.dz : s.sdz;
            }
            s.mx += steps_signed(s.sdx - ox, gx); s.my += steps_signed(s.sdy - oy, gy); s.mz += steps_signed(s.sdz - oz, gz);
        }
#endif
        i += kw;
    }
    if (CNT && v.count_lookups != 0u) fetches = lk_bytes;       // (VRT_FLAG_LOOKUP_COUNTS: the bytes the look-ups asked for)
#if defined(__HIP_DEVICE_COMPILE__)
    finish(s, material, lmask, fetches, r);
#if defined(VRT_TRACE_COUNTERS)
    r.dbg0 = n_look; r.dbg1 = n_occ;
#endif
#else
    finish(s, material, (uint32_t)k0 | ((uint32_t)k1 << 1) | ((uint32_t)k2 << 2), fetches, r);
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// The brick march for rays that point every way (AO): every lane spends its own clearance, up to VRT_OWN_CAP iterations per
// look (df_any_loop's scheme in the generic loop: an iteration runs under the ballot of the lanes that have some left).
template <class STATS, bool CNT = false>
__device__ __forceinline__ void trace_brick_own(const VolumeView& v, f3 start, f3 dir, uint32_t maxSteps, RayInt& r, STATS& stats)
{
    DdaState s;
    dda_entry(v, start, dir, s);
    if (wave_all(oob(v, s.mx, s.my, s.mz))) {
        s.dx = s.dy = s.dz = 0.0f; s.sdx = s.sdy = s.sdz = 0.0f; s.sx = s.sy = s.sz = 0;
        finish(s, 0u, s.mask, 0u, r);
        return;
    }
    dda_rest(dir, s);
    asm volatile("" : "+v"(s.dx), "+v"(s.dy), "+v"(s.dz));
    uint32_t material = 0u, fetches = 0u, i = 0u, lk_bytes = 0u;             // i: iterations THIS lane has taken
    bool done = oob(v, s.mx, s.my, s.mz);
    const uint32_t oct = (uint32_t)(s.sx > 0) | ((uint32_t)(s.sy > 0) << 1) | ((uint32_t)(s.sz > 0) << 2);
    const float kInf = u2f(0x7F800000u);
    const float gx = s.dx < kInf ? dir.x : 0.0f, gy = s.dy < kInf ? dir.y : 0.0f, gz = s.dz < kInf ? dir.z : 0.0f;
    for (;;) {
        uint32_t own = 0u;
        if (!done) {
            if (i >= maxSteps) { done = true; fetches = i; }
            else {
                uint32_t m = 0u;
                const uint32_t clear = brick_clear(v, s.mx, s.my, s.mz, oct, s.sx, s.sy, s.sz, m, CNT ? &lk_bytes : nullptr);
                st_lookup(stats);
                if (clear == 0u) {
                    if (oob(v, s.mx, s.my, s.mz)) fetches = i;
                    else { material = m; fetches = i + 1u; }
                    done = true;
                } else if (clear >= maxSteps - i) { done = true; fetches = v.count_marched ? i : maxSteps; }
                else own = clear < (uint32_t)VRT_OWN_CAP_BRICK ? clear : (uint32_t)VRT_OWN_CAP_BRICK;
            }
        }
        if (__ballot(own != 0u) == 0ull) break;
        const float ox = s.sdx, oy = s.sdy, oz = s.sdz;
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)VRT_OWN_CAP_BRICK; j++) {
            const uint64_t mk = __ballot(own > j);
            if (mk == 0ull) break;
            dda_advance_live(s, mk);
        }
        s.mx += steps_signed(s.sdx - ox, gx); s.my += steps_signed(s.sdy - oy, gy); s.mz += steps_signed(s.sdz - oz, gz);
        i += own;
    }
    finish(s, material, s.mask, (CNT && v.count_lookups != 0u) ? lk_bytes : fetches, r);
}

// ---- the same pool for brick scenes (df_ao_pool_loop's scheme in the generic loop) ---------------------------------------------
// A ray of the pool: 13 dwords (sideDist, deltaDist, 1 / delta with its sign, mapPos, tag = column of its pixel | first clearance << 8), slot k
// at row q: q * 256 + k * 4; the two rows of counters behind them (3328: rays of the column's pixel that found a solid voxel; 3584, CNT:
// what the count planes report).
struct BrickAoLane { DdaState s; float gx, gy, gz; uint32_t i, owner; bool live; };
__device__ __forceinline__ void brick_ao_rest(BrickAoLane& l)
{
    l.s.sdx = l.s.sdy = l.s.sdz = 0.0f; l.s.dx = l.s.dy = l.s.dz = 0.0f; l.s.mx = l.s.my = l.s.mz = 0; l.s.sx = l.s.sy = l.s.sz = 0;
    l.gx = l.gy = l.gz = 0.0f; l.i = 0u; l.owner = 0u; l.live = false;
}
// the set-up of the lane's pixel's next AO ray (frag:109-144) ...
__device__ __forceinline__ void brick_ao_setup(const VolumeView& v, f3 start, f3 dir, DdaState& s, float& gx, float& gy, float& gz)
{
    dda_entry(v, start, dir, s);
    dda_rest(dir, s);
    const float kInf = u2f(0x7F800000u);
    gx = s.dx < kInf ? dir.x : 0.0f; gy = s.dy < kInf ? dir.y : 0.0f; gz = s.dz < kInf ? dir.z : 0.0f;
}
// ... and the ray into slot `slot` of the pool; tag: the column of its pixel's counters | the clearance its owner read at its first voxel << 8
__device__ __forceinline__ void brick_ao_store(uint32_t ldsw, uint32_t slot, const DdaState& s, float gx, float gy, float gz, uint32_t tag)
{
    __attribute__((address_space(3))) uint32_t* p = (__attribute__((address_space(3))) uint32_t*)(uintptr_t)(ldsw + slot * 4u);
    p[0 * 64] = f2u(s.sdx); p[1 * 64] = f2u(s.sdy); p[2 * 64] = f2u(s.sdz);
    p[3 * 64] = f2u(s.dx); p[4 * 64] = f2u(s.dy); p[5 * 64] = f2u(s.dz);
    p[6 * 64] = f2u(gx); p[7 * 64] = f2u(gy); p[8 * 64] = f2u(gz);
    p[9 * 64] = (uint32_t)s.mx; p[10 * 64] = (uint32_t)s.my; p[11 * 64] = (uint32_t)s.mz; p[12 * 64] = tag;
}
// One call: `count` rays wait in the pool, `next` of them are taken (in / out); more: come back when the pool is empty and a lane
// rests.  Every lane spends its ray's own clearance, at most VRT_OWN_CAP_BRICK iterations per look (trace_brick_own).
template <bool CNT>
__device__ __forceinline__ void brick_ao_pool(const VolumeView& v, BrickAoLane& l, uint32_t ldsw, uint32_t count, bool more, uint32_t& next,
                                              uint32_t maxSteps, uint32_t& looks)
{
    __attribute__((address_space(3))) uint32_t* pool = (__attribute__((address_space(3))) uint32_t*)(uintptr_t)ldsw;
    for (;;) {
        uint32_t own = 0u;
        if (l.live) {
            bool ended = false, solid = false;
            uint32_t fet = 0u;
            if (l.i >= maxSteps) { ended = true; fet = l.i; }
            else {
                const uint32_t oct = (uint32_t)(l.s.sx > 0) | ((uint32_t)(l.s.sy > 0) << 1) | ((uint32_t)(l.s.sz > 0) << 2);
                uint32_t m = 0u;
                const uint32_t clear = brick_clear(v, l.s.mx, l.s.my, l.s.mz, oct, l.s.sx, l.s.sy, l.s.sz, m, CNT ? &looks : nullptr);
                if (clear == 0u) {
                    ended = true;
                    solid = !oob(v, l.s.mx, l.s.my, l.s.mz) && m != 0u;
                    fet = solid ? l.i + 1u : l.i;
                } else if (clear >= maxSteps - l.i) { ended = true; fet = v.count_marched ? l.i : maxSteps; }
                else own = clear < (uint32_t)VRT_OWN_CAP_BRICK ? clear : (uint32_t)VRT_OWN_CAP_BRICK;
            }
            if (ended) {
                if (solid) __hip_atomic_fetch_add(pool + 13 * 64 + l.owner, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (CNT) __hip_atomic_fetch_add(pool + 14 * 64 + l.owner, fet, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                l.live = false;
            }
        }
        // lanes without a ray take the next ones of the pool: the wave's counter + the lane's rank among those asking
        const uint64_t asking = __ballot(!l.live);
        if (!l.live) {
            const uint32_t slot = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(asking >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)asking, 0u));
            if (slot < count) {
                __attribute__((address_space(3))) uint32_t* p = pool + slot;
                l.s.sdx = u2f(p[0 * 64]); l.s.sdy = u2f(p[1 * 64]); l.s.sdz = u2f(p[2 * 64]);
                l.s.dx = u2f(p[3 * 64]); l.s.dy = u2f(p[4 * 64]); l.s.dz = u2f(p[5 * 64]);
                l.gx = u2f(p[6 * 64]); l.gy = u2f(p[7 * 64]); l.gz = u2f(p[8 * 64]);
                l.s.mx = (int)p[9 * 64]; l.s.my = (int)p[10 * 64]; l.s.mz = (int)p[11 * 64];
                // rayStep from the direction's signs (an axis the ray cannot step along has g = 0: the step is never taken)
                l.s.sx = l.gx > 0.0f ? 1 : (l.gx < 0.0f ? -1 : 0); l.s.sy = l.gy > 0.0f ? 1 : (l.gy < 0.0f ? -1 : 0); l.s.sz = l.gz > 0.0f ? 1 : (l.gz < 0.0f ? -1 : 0);
                const uint32_t tag = p[12 * 64];
                l.i = 0u; l.owner = tag & 0xFFu; l.live = true;
                // (its owner read the clearance at its first voxel: 1 .. maxSteps - 1; the lane marches it in this very round)
                const uint32_t c0 = tag >> 8;
                own = c0 < (uint32_t)VRT_OWN_CAP_BRICK ? c0 : (uint32_t)VRT_OWN_CAP_BRICK;
            }
        }
        const uint32_t taken = next + (uint32_t)__builtin_popcountll(asking);
        next = taken < count ? taken : count;
        const uint64_t live = __ballot(l.live);
        if (next >= count && more && __ballot(!l.live) != 0ull) return;      // the pool wants refilling
        if (live == 0ull) return;
        if (__ballot(own != 0u) == 0ull) continue;                          // (only rays just taken up: their first look)
        const float ox = l.s.sdx, oy = l.s.sdy, oz = l.s.sdz;
        asm volatile("" : "+v"(l.s.dx), "+v"(l.s.dy), "+v"(l.s.dz));
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)VRT_OWN_CAP_BRICK; j++) {
            const uint64_t mk = __ballot(own > j);
            if (mk == 0ull) break;
            dda_advance_live(l.s, mk);
        }
        l.s.mx += steps_signed(l.s.sdx - ox, l.gx); l.s.my += steps_signed(l.s.sdy - oy, l.gy); l.s.mz += steps_signed(l.s.sdz - oz, l.gz);
        l.i += own;
    }
}
#else
template <class STATS, bool CNT = false>
VRT_HD void trace_brick_own(const VolumeView&, f3, f3, uint32_t, RayInt&, STATS&) {}
struct BrickAoLane { DdaState s; float gx, gy, gz; uint32_t i, owner; bool live; };
VRT_HD void brick_ao_rest(BrickAoLane&) {}
VRT_HD void brick_ao_setup(const VolumeView&, f3, f3, DdaState&, float&, float&, float&) {}
VRT_HD void brick_ao_store(uint32_t, uint32_t, const DdaState&, float, float, float, uint32_t) {}
template <bool CNT> VRT_HD void brick_ao_pool(const VolumeView&, BrickAoLane&, uint32_t, uint32_t, bool, uint32_t&, uint32_t, uint32_t&) {}
#endif

// DENSE: one R8 fetch per iteration.  Straight-line body with a single exit (out of budget, out of bounds or
// solid), the voxel index maintained incrementally in IDX (uint32_t for volumes below 4 GiB).
template <class IDX>
VRT_HD void trace_dense(const VolumeView& v, f3 start, f3 dir, uint32_t maxSteps, RayInt& r)
{
    DdaState s;
    dda_setup(v, start, dir, s);
    uint32_t mask = s.mask, material = 0, i = 0;
    IDX idx = (IDX)s.mx + (IDX)s.my * (IDX)v.W + (IDX)s.mz * (IDX)v.W * (IDX)v.H;   // meaningless (and unused) while out of bounds
    const IDX incx = (IDX)(int64_t)s.sx, incy = (IDX)((int64_t)s.sy * (int64_t)v.W),
              incz = (IDX)((int64_t)s.sz * (int64_t)v.W * (int64_t)v.H);
    for (;;) {
        bool stop = i >= maxSteps || oob(v, s.mx, s.my, s.mz);
        uint32_t m = v.vox[stop ? (IDX)0 : idx];
        if (stop || m != 0u) { material = stop ? 0u : m; break; }
        ++i;
        uint32_t bx = f2u(s.sdx), by = f2u(s.sdy), bz = f2u(s.sdz);
        uint32_t mn = umin3(bx, by, bz);
        bool m0 = bx == mn, m1 = by == mn, m2 = bz == mn;
        mask = (uint32_t)m0 | ((uint32_t)m1 << 1) | ((uint32_t)m2 << 2);
        s.sdx = m0 ? s.sdx + s.dx : s.sdx; s.mx += m0 ? s.sx : 0;
        s.sdy = m1 ? s.sdy + s.dy : s.sdy; s.my += m1 ? s.sy : 0;
        s.sdz = m2 ? s.sdz + s.dz : s.sdz; s.mz += m2 ? s.sz : 0;
        idx += (m0 ? incx : (IDX)0) + (m1 ? incy : (IDX)0) + (m2 ? incz : (IDX)0);
    }
    finish(s, material, mask, material ? i + 1u : i, r);
}

// DENSE, latency-hiding form.  The DDA advance does not depend on the fetched voxel -- only the exit test
// does -- so the march runs B iterations ahead with B fetches in flight, then looks for the first iteration
// that should have stopped; if there is one, the saved state is replayed up to it with the same arithmetic.
// One fetch latency (an L2 / Infinity-Cache hit, 200-550 cycles) is paid per B iterations instead of per
// iteration; this is what shortens the critical path of the longest ray in a frame.
template <int B>
VRT_HD void trace_dense_blocked(const VolumeView& v, f3 start, f3 dir, uint32_t maxSteps, RayInt& r)
{
    DdaState s;
    dda_setup(v, start, dir, s);
    uint32_t mask = s.mask, material = 0, i = 0;
    uint32_t idx = (uint32_t)s.mx + (uint32_t)s.my * (uint32_t)v.W + (uint32_t)s.mz * (uint32_t)v.W * (uint32_t)v.H;
    const uint32_t incx = (uint32_t)s.sx, incy = (uint32_t)(s.sy * v.W), incz = (uint32_t)(s.sz * v.W * v.H);
    for (;;) {
        // snapshot
        const float s0x = s.sdx, s0y = s.sdy, s0z = s.sdz;
        const int m0x = s.mx, m0y = s.my, m0z = s.mz;
        const uint32_t mask0 = mask;
        uint32_t vox[B];
        uint32_t stopbits = 0;
#pragma unroll
        for (int b = 0; b < B; b++) {
            bool stop = (i + (uint32_t)b) >= maxSteps || oob(v, s.mx, s.my, s.mz);
            stopbits |= (uint32_t)stop << b;
            vox[b] = v.vox[stop ? 0u : idx];
            uint32_t bx = f2u(s.sdx), by = f2u(s.sdy), bz = f2u(s.sdz);
            uint32_t mn = umin3(bx, by, bz);
            bool k0 = bx == mn, k1 = by == mn, k2 = bz == mn;
            mask = (uint32_t)k0 | ((uint32_t)k1 << 1) | ((uint32_t)k2 << 2);
            s.sdx = k0 ? s.sdx + s.dx : s.sdx; s.mx += k0 ? s.sx : 0;
            s.sdy = k1 ? s.sdy + s.dy : s.sdy; s.my += k1 ? s.sy : 0;
            s.sdz = k2 ? s.sdz + s.dz : s.sdz; s.mz += k2 ? s.sz : 0;
            idx += (k0 ? incx : 0u) + (k1 ? incy : 0u) + (k2 ? incz : 0u);
        }
        uint32_t exitbits = stopbits;
#pragma unroll
        for (int b = 0; b < B; b++) exitbits |= (uint32_t)(vox[b] != 0u) << b;
        if (exitbits == 0u) { i += (uint32_t)B; continue; }
        // first iteration of the block that stops: replay the snapshot up to it
        int first = 0;
#pragma unroll
        for (int b = B - 1; b >= 0; b--) if ((exitbits >> b) & 1u) first = b;
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < B; b++) if (b == first) m = vox[b];
        material = ((stopbits >> first) & 1u) ? 0u : m;
        s.sdx = s0x; s.sdy = s0y; s.sdz = s0z; s.mx = m0x; s.my = m0y; s.mz = m0z; mask = mask0;
        for (int b = 0; b < first; b++) VRT_DDA_STEP(s, mask);
        i += (uint32_t)first;
        break;
    }
    finish(s, material, mask, material ? i + 1u : i, r);
}

template <int TRAV, class OP>
VRT_HD void trace_literal(const VolumeView& v, OP o2, f3 start, f3 dir, uint32_t maxSteps, RayInt& r)
{
    if (TRAV == VRT_TRAVERSAL_DENSE) {
        trace_dense_blocked<4>(v, start, dir, maxSteps, r);    // the API rejects DENSE for volumes of 4 GiB and more
        return;
    }
    DdaState s;
    dda_setup(v, start, dir, s);
    uint32_t mask = s.mask, material = 0, fetches = 0;
    uint32_t ckey = 0xFFFFFFFFu;
    uint64_t word = 0;
    uint32_t i = 0;
    for (; i < maxSteps; i++) {
        if (oob(v, s.mx, s.my, s.mz)) break;
        uint32_t key = (uint32_t)(s.mx >> 2) | ((uint32_t)(s.my >> 2) << 10) | ((uint32_t)(s.mz >> 2) << 20);
        if (key != ckey) { ckey = key; word = fetch_cell(v, o2, s.mx >> 2, s.my >> 2, s.mz >> 2); }
        if ((word >> cell_bit(s.mx, s.my, s.mz)) & 1ull) {
            material = voxel_at(v, s.mx, s.my, s.mz);
            fetches = i + 1;
            break;
        }
        VRT_DDA_STEP(s, mask);
    }
    if (material == 0) fetches = i;
    finish(s, material, mask, fetches, r);
}

// ---- exact jumps ----------------------------------------------------------------------------------------

// Increment of the bit pattern per addition of d while the sum stays in the binade of `bits`, and how many
// further additions are guaranteed to stay there (conservative).  false: no closed form here (zero/denormal/
// inf sideDist, d not below the binade of s, or a half-ulp tie on an odd mantissa) -> step that axis literally.
VRT_HD bool axis_increment(uint32_t bits, float d, uint32_t& Q, uint32_t& jmax)
{
    uint32_t db = f2u(d);
    uint32_t E = bits >> 23, Ed = db >> 23;
    int shift = (int)E - (int)Ed;
    if (E == 0u || E >= 255u || shift < 1 || shift > 20) return false;
    uint32_t Md = (db & 0x7FFFFFu) | 0x800000u;
    uint32_t q = Md >> shift;
    uint32_t rem = Md & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half) q += 1u;
    else if (rem == half) { if (bits & 1u) return false; q += (q & 1u); }
    uint32_t room = 0xFFFFFFu - ((bits & 0x7FFFFFu) | 0x800000u);       // mantissa head-room in this binade
    uint32_t je = (uint32_t)((float)room * rcp_approx((float)q));
    jmax = je > 0u ? je - 1u : 0u;                                       // conservative: never over-estimates
    Q = q;
    return true;
}

struct JumpAxis { uint32_t bits, Q, T; int n; };

VRT_HD void jump_axis_prepare(float side, float d, int step, int n_region, JumpAxis& a)
{
    a.bits = f2u(side);
    a.Q = 0u;
    if (step == 0) { a.n = 0; a.T = 0xFFFFFFFFu; return; }
    uint32_t jmax;
    if (axis_increment(a.bits, d, a.Q, jmax)) {
        int lim = (int)jmax + 1;
        a.n = n_region < lim ? n_region : lim;
    } else {
        a.n = 1; a.Q = 0u;
    }
    a.T = a.bits + (uint32_t)(a.n - 1) * a.Q;
}

// Advance axis by the steps it takes up to and including time T*.  Returns the number of steps; `last` tells
// whether its final step happened exactly at T* (it then belongs to the final iteration's mask).
VRT_HD int jump_axis_apply(const JumpAxis& a, uint32_t Tstar, float d, float& side, bool& last)
{
    last = false;
    if (a.n == 0) return 0;
    if (a.T == Tstar) {                                   // reaches its n-th step at T*: last addition in fp32
        side = u2f(a.T) + d;
        last = true;
        return a.n;
    }
    if (a.Q == 0u || Tstar < a.bits) return 0;
    uint32_t num = Tstar - a.bits;
    uint32_t c = (uint32_t)((float)num * rcp_approx((float)a.Q));
    if (c * a.Q > num) c -= 1u;
    else if ((c + 1u) * a.Q <= num) c += 1u;
    last = (c * a.Q == num);
    c += 1u;                                              // steps j = 0..c-1 have S(j) <= T*
    side = u2f(a.bits + c * a.Q);
    return (int)c;
}

VRT_HD int iabs(int x) { return x < 0 ? -x : x; }

template <class STATS, class OP>
VRT_HD void trace_jump(const VolumeView& v, OP o2, OP o3, f3 start, f3 dir,
                       uint32_t maxSteps, RayInt& r, STATS& stats)
{
    DdaState s;
    dda_setup(v, start, dir, s);
    const int mx0 = s.mx, my0 = s.my, mz0 = s.mz;
    uint32_t mask = s.mask, material = 0;
    uint32_t it_up = 0;                                   // upper bound of DDA iterations done (sum of axis steps)
    uint32_t ckey = 0xFFFFFFFFu;
    uint64_t word = 0;
    int lvl = 0;
    bool hit = false;
    for (;;) {
        if (oob(v, s.mx, s.my, s.mz)) break;
        uint32_t key = (uint32_t)(s.mx >> 2) | ((uint32_t)(s.my >> 2) << 10) | ((uint32_t)(s.mz >> 2) << 20);
        if (key != ckey) { ckey = key; lvl = lookup_level(v, o2, o3, s.mx, s.my, s.mz, word); st_lookup(stats); }
        if (lvl == 0) {
            if ((word >> cell_bit(s.mx, s.my, s.mz)) & 1ull) { hit = true; break; }
            VRT_DDA_STEP(s, mask);
            it_up += (mask & 1u) + ((mask >> 1) & 1u) + ((mask >> 2) & 1u);
            st_literal(stats);
        } else {
            // empty aligned cell of edge sz around mapPos, clipped to the volume
            int sh = 2 * lvl, sz = 1 << sh;
            int lox = (s.mx >> sh) << sh, loy = (s.my >> sh) << sh, loz = (s.mz >> sh) << sh;
            int hix = lox + sz < v.W ? lox + sz : v.W;
            int hiy = loy + sz < v.H ? loy + sz : v.H;
            int hiz = loz + sz < v.D ? loz + sz : v.D;
            JumpAxis ax, ay, az;
            jump_axis_prepare(s.sdx, s.dx, s.sx, s.sx > 0 ? hix - s.mx : s.mx - lox + 1, ax);
            jump_axis_prepare(s.sdy, s.dy, s.sy, s.sy > 0 ? hiy - s.my : s.my - loy + 1, ay);
            jump_axis_prepare(s.sdz, s.dz, s.sz, s.sz > 0 ? hiz - s.mz : s.mz - loz + 1, az);
            uint32_t Tstar = ax.T < ay.T ? ax.T : ay.T;
            Tstar = Tstar < az.T ? Tstar : az.T;
            if (Tstar == 0xFFFFFFFFu) break;              // direction (0,0,0): the literal loop spins to the budget -> miss
            bool l0, l1, l2;
            int c0 = jump_axis_apply(ax, Tstar, s.dx, s.sdx, l0);
            int c1 = jump_axis_apply(ay, Tstar, s.dy, s.sdy, l1);
            int c2 = jump_axis_apply(az, Tstar, s.dz, s.sdz, l2);
            s.mx += c0 * s.sx; s.my += c1 * s.sy; s.mz += c2 * s.sz;
            mask = (uint32_t)l0 | ((uint32_t)l1 << 1) | ((uint32_t)l2 << 2);
            it_up += (uint32_t)(c0 + c1 + c2);
            st_jump(stats, lvl);
            // every iteration steps each axis at most once: once one axis alone has taken maxSteps steps the
            // literal loop is certainly exhausted
            int klo = iabs(s.mx - mx0); int k1 = iabs(s.my - my0); int k2 = iabs(s.mz - mz0);
            klo = klo > k1 ? klo : k1; klo = klo > k2 ? klo : k2;
            if ((uint32_t)klo >= maxSteps) break;
        }
    }
    if (hit) {
        int klo = iabs(s.mx - mx0); int k1 = iabs(s.my - my0); int k2 = iabs(s.mz - mz0);
        klo = klo > k1 ? klo : k1; klo = klo > k2 ? klo : k2;
        if (it_up < maxSteps) {
            material = voxel_at(v, s.mx, s.my, s.mz);     // the literal loop reaches this fetch within its budget
        } else if ((uint32_t)klo >= maxSteps) {
            hit = false;                                  // certainly exhausted before reaching this voxel
        } else {
            st_retrace(stats);                            // ties make the count ambiguous: decide literally
            trace_literal<VRT_TRAVERSAL_BITMASK>(v, o2, start, dir, maxSteps, r);
            return;
        }
    }
    finish(s, material, mask, hit ? it_up + 1u : it_up, r);
}

// DFJ: DF with the long runs done in closed form.  A lane whose clearance is >= kJumpMin does not walk its run: the
// cube its clearance guarantees empty is exactly the kind of region trace_jump() leaves in one step, so it jumps to the
// cube's exit (or as far as its sideDist binades allow) with the integer closed form above, independently of the other
// lanes.  Lanes with small clearances walk wave-cooperative runs as in DF.  Iteration counts become bounds
// (lo <= iterations <= lo + slack); hits that straddle the budget are re-traced literally, as in JUMP.
template <class STATS>
VRT_HD void trace_dfj(const VolumeView& v, f3 start, f3 dir, uint32_t maxSteps, RayInt& r, STATS& stats)
{
    constexpr uint32_t kJumpMin = 12u;
    DdaState s;
    dda_setup(v, start, dir, s);
    uint32_t mask = s.mask, material = 0;
    bool done = false, hit = false;
    uint32_t clear = 63u;
    const uint32_t oct = (uint32_t)(s.sx > 0) | ((uint32_t)(s.sy > 0) << 1) | ((uint32_t)(s.sz > 0) << 2);
    const size_t octant = (size_t)oct * (size_t)v.df_stride;
    const float kInf = u2f(0x7F800000u);
    const float gx = s.dx < kInf ? fabsf(dir.x) : 0.0f, gy = s.dy < kInf ? fabsf(dir.y) : 0.0f, gz = s.dz < kInf ? fabsf(dir.z) : 0.0f;
    uint32_t lo = 0, slack = 0;                                // lo <= DDA iterations done so far <= lo + slack
    for (;;) {
        if (!done) {
            if (lo >= maxSteps || oob(v, s.mx, s.my, s.mz)) done = true;       // certainly out of budget, or out of the volume
            else {
                clear = v.df[octant + df_index(v, s.mx, s.my, s.mz)];
                st_lookup(stats);
                if (clear == 0u) { hit = true; done = true; }
            }
        }
        if (wave_all(done)) break;
        const bool jumper = !done && clear >= kJumpMin;
        if (jumper) {
            JumpAxis ax, ay, az;
            jump_axis_prepare(s.sdx, s.dx, s.sx, (int)clear, ax);
            jump_axis_prepare(s.sdy, s.dy, s.sy, (int)clear, ay);
            jump_axis_prepare(s.sdz, s.dz, s.sz, (int)clear, az);
            uint32_t Tstar = ax.T < ay.T ? ax.T : ay.T;
            Tstar = Tstar < az.T ? Tstar : az.T;
            if (Tstar == 0xFFFFFFFFu) { done = true; }        // direction (0,0,0): the literal loop spins to its budget -> miss
            else {
                bool l0, l1, l2;
                int c0 = jump_axis_apply(ax, Tstar, s.dx, s.sdx, l0);
                int c1 = jump_axis_apply(ay, Tstar, s.dy, s.sdy, l1);
                int c2 = jump_axis_apply(az, Tstar, s.dz, s.sdz, l2);
                s.mx += c0 * s.sx; s.my += c1 * s.sy; s.mz += c2 * s.sz;
                mask = (uint32_t)l0 | ((uint32_t)l1 << 1) | ((uint32_t)l2 << 2);
                int cm = c0 > c1 ? c0 : c1; cm = cm > c2 ? cm : c2;
                lo += (uint32_t)cm; slack += (uint32_t)(c0 + c1 + c2 - cm);
                st_jump(stats, 3);
            }
        }
        const bool walker = !done && !jumper;
        if (wave_any(walker)) {
            uint32_t kw = wave_min_u6(walker ? clear : 63u);
            st_jump(stats, 1);
            if (walker) {
                const float ox = s.sdx, oy = s.sdy, oz = s.sdz;
                for (uint32_t j = 1; j < kw; j++) {
                    uint32_t bx = f2u(s.sdx), by = f2u(s.sdy), bz = f2u(s.sdz);
                    uint32_t mn = umin3(bx, by, bz);
                    s.sdx = bx == mn ? s.sdx + s.dx : s.sdx;
                    s.sdy = by == mn ? s.sdy + s.dy : s.sdy;
                    s.sdz = bz == mn ? s.sdz + s.dz : s.sdz;
                }
                {
                    uint32_t bx = f2u(s.sdx), by = f2u(s.sdy), bz = f2u(s.sdz);
                    uint32_t mn = umin3(bx, by, bz);
                    bool k0 = bx == mn, k1 = by == mn, k2 = bz == mn;
                    mask = (uint32_t)k0 | ((uint32_t)k1 << 1) | ((uint32_t)k2 << 2);
                    s.sdx = k0 ? s.sdx + s.dx : s.sdx;
                    s.sdy = k1 ? s.sdy + s.dy : s.sdy;
                    s.sdz = k2 ? s.sdz + s.dz : s.sdz;
                }
                int nx = steps_taken((s.sdx - ox) * gx), ny = steps_taken((s.sdy - oy) * gy), nz = steps_taken((s.sdz - oz) * gz);
                s.mx += s.sx < 0 ? -nx : nx;
                s.my += s.sy < 0 ? -ny : ny;
                s.mz += s.sz < 0 ? -nz : nz;
                lo += kw;
            }
        }
    }
    if (hit) {
        if (lo + slack < maxSteps) material = voxel_at(v, s.mx, s.my, s.mz);   // the literal loop reaches this fetch within its budget
        else if (lo >= maxSteps) hit = false;                                   // certainly exhausted before it
        else {
            st_retrace(stats);
            trace_df(v, start, dir, maxSteps, r, stats);                         // ambiguous (ties near the budget): decide literally
            return;
        }
    }
    finish(s, material, mask, hit ? lo + slack + 1u : lo + slack, r);
}

// Dispatcher used by the kernels.
// ANYHIT: the caller only uses r.material and r.fetches (traceRayHit, frag:198-202): a traversal may then stop stepping a ray
// that is certain to exhaust its budget in empty space (DF_FAST does)
template <int TRAV, class OP, bool AHEAD = false, bool ANYHIT = false, bool PF = false, bool OWN = false>
VRT_HD void trace_int(const VolumeView& v, OP o2, OP o3, f3 start, f3 dir,
                      uint32_t maxSteps, RayInt& r)
{
    if (TRAV == VRT_TRAVERSAL_JUMP) {
        NoStats ns;
        trace_jump(v, o2, o3, start, dir, maxSteps, r, ns);
    } else if (TRAV == VRT_TRAVERSAL_BRICK) {
        NoStats ns;
        if (ANYHIT && OWN && v.df_own) trace_brick_own(v, start, dir, maxSteps, r, ns);
        else trace_brick<NoStats, ANYHIT>(v, start, dir, maxSteps, r, ns);
    } else if (TRAV == VRT_TRAVERSAL_BRICK_CNT) {
        NoStats ns;
        if (ANYHIT && OWN && v.df_own) trace_brick_own<NoStats, true>(v, start, dir, maxSteps, r, ns);
        else trace_brick<NoStats, ANYHIT, true>(v, start, dir, maxSteps, r, ns);
    } else if (TRAV == VRT_TRAVERSAL_DF_FAST) {
        NoStats ns;
        trace_df_fast<NoStats, ANYHIT, PF, OWN>(v, start, dir, maxSteps, r, ns);
    } else if (TRAV == VRT_TRAVERSAL_DF_FAST_CNT) {
        NoStats ns;
        trace_df_fast<NoStats, ANYHIT, PF, OWN, true>(v, start, dir, maxSteps, r, ns);
    } else if (TRAV == VRT_TRAVERSAL_DF) {
        NoStats ns;
        trace_df<NoStats, AHEAD>(v, start, dir, maxSteps, r, ns);
    } else if (TRAV == VRT_TRAVERSAL_DFJ) {
        NoStats ns;
        trace_dfj(v, start, dir, maxSteps, r, ns);
    } else {
        trace_literal<TRAV>(v, o2, start, dir, maxSteps, r);
    }
}

} // namespace vrt
