// vrt_device.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4; 64-wide wavefronts).
//
//   k_build_occ1/2    occupancy pyramid over the dense R8 volume (scene build)
//   k_primary<...>    K1: per-pixel ray generation + Amanatides-Woo DDA + G-buffer write
//                     (voxel_volume.frag:309-346, :109-196 of the reference)
//   k_shade<...>      K2: AO / shadow / mirror-bounce rays + shading (voxel_volume.frag:205-307)
//   k_denoise(_lds)   K3: one a-trous cross-bilateral pass (denoiser.frag:38-73)
//   k_rows(_batch)    strip pack / unpack for the multi-GPU gather and halo exchange
//   k_blit, k_accumulate, k_resolve   presentation / temporal rows (blit.frag, the FSR2 stand-in)
//
// A wave owns an 8x8 pixel block so that its 64 rays stay spatially coherent; the default (clearance-field)
// traversal runs one wave per workgroup, the LDS-staged ones 16x16 tiles of four waves.  Rays are generated in-kernel
// from the 96-byte push-constant block (no ray buffers); a launch covers up to 8 frames.  No MFMA: nothing here is a
// dense contraction.
//
// All arithmetic follows vrt_spec.h (fp32, -ffp-contract=off); the DDA state (sideDist, mapPos, mask)
// is advanced with exactly the additions of voxel_volume.frag:164-170 in every traversal mode, so hit
// voxel, mask, t and step budget are independent of the mode.
#include "vrt_internal.h"
#include "vrt_spec.h"

namespace vrt {

// ---------------------------------------------------------------------------------------------
// occupancy pyramid build
// ---------------------------------------------------------------------------------------------

__global__ void k_build_occ1(const uint8_t* __restrict__ vox, int W, int H, int D,
                             uint64_t* __restrict__ occ1, int n1x, int n1y, int n1z)
{
    int cx = blockIdx.x * blockDim.x + threadIdx.x;
    int cy = blockIdx.y, cz = blockIdx.z;
    if (cx >= n1x) return;
    uint64_t w = 0;
    for (int z = 0; z < 4; z++) {
        int vz = cz * 4 + z;
        if (vz >= D) break;
        for (int y = 0; y < 4; y++) {
            int vy = cy * 4 + y;
            if (vy >= H) break;
            size_t base = (size_t)cx * 4 + ((size_t)vy + (size_t)vz * H) * W;
            for (int x = 0; x < 4; x++) {
                int vx = cx * 4 + x;
                if (vx < W && vox[base + x] != 0) w |= 1ull << (x | (y << 2) | (z << 4));
            }
        }
    }
    occ1[(size_t)cx + ((size_t)cy + (size_t)cz * n1y) * n1x] = w;
}

// level k+1 from level k: bit set <=> child word != 0
__global__ void k_build_occ_up(const uint64_t* __restrict__ lo, int lx, int ly, int lz,
                               uint64_t* __restrict__ hi, int hx, int hy, int hz)
{
    int cx = blockIdx.x * blockDim.x + threadIdx.x;
    int cy = blockIdx.y, cz = blockIdx.z;
    if (cx >= hx) return;
    uint64_t w = 0;
    for (int z = 0; z < 4; z++) {
        int vz = cz * 4 + z;
        if (vz >= lz) break;
        for (int y = 0; y < 4; y++) {
            int vy = cy * 4 + y;
            if (vy >= ly) break;
            for (int x = 0; x < 4; x++) {
                int vx = cx * 4 + x;
                if (vx < lx && lo[(size_t)vx + ((size_t)vy + (size_t)vz * ly) * lx] != 0)
                    w |= 1ull << (x | (y << 2) | (z << 4));
            }
        }
    }
    hi[(size_t)cx + ((size_t)cy + (size_t)cz * hy) * hx] = w;
}

hipError_t launch_build_pyramid(const uint8_t* vox, int W, int H, int D, uint64_t* occ1, uint64_t* occ2,
                                uint64_t* occ3, hipStream_t s)
{
    int n1x = (W + 3) / 4, n1y = (H + 3) / 4, n1z = (D + 3) / 4;
    int n2x = (n1x + 3) / 4, n2y = (n1y + 3) / 4, n2z = (n1z + 3) / 4;
    int n3x = (n2x + 3) / 4, n3y = (n2y + 3) / 4, n3z = (n2z + 3) / 4;
    hipLaunchKernelGGL(k_build_occ1, dim3((n1x + 63) / 64, n1y, n1z), dim3(64), 0, s, vox, W, H, D, occ1, n1x, n1y, n1z);
    hipLaunchKernelGGL(k_build_occ_up, dim3((n2x + 63) / 64, n2y, n2z), dim3(64), 0, s, occ1, n1x, n1y, n1z, occ2, n2x, n2y, n2z);
    hipLaunchKernelGGL(k_build_occ_up, dim3((n3x + 63) / 64, n3y, n3z), dim3(64), 0, s, occ2, n2x, n2y, n2z, occ3, n3x, n3y, n3z);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// clearance fields (scene build).  For octant o = (sx, sy, sz) in {-1,+1}^3, c_o(p) = side of the largest empty
// cube with corner p extending towards (sx, sy, sz), 0 for a solid voxel, capped at 63:
//   c(p) = min_{c>=0} max(c, min_{b>=0} max(b, min_{a>=0} max(a, solid(p + (a sx, b sy, c sz)) ? 0 : INF)))
// i.e. three one-sided 1-D min-max passes.  Outside the volume counts as solid, so a run never carries a ray more
// than one voxel past a wall.
// ---------------------------------------------------------------------------------------------

#define VRT_DF_CAP 127      // >= 64: a 64-iteration AO ray that starts in the open is decided by its first look-up (trace_df_fast, any-hit)

// src == nullptr: first pass, the field is (vox != 0 ? 0 : INF).
__global__ __launch_bounds__(256) void k_df_pass(const uint8_t* __restrict__ vox, const uint8_t* __restrict__ src,
                                                 uint8_t* __restrict__ dst, int W, int H, int D, int axis, int dir, int padded, int cap)
{
    size_t n = (size_t)W * H * D;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x = (int)(i % (size_t)W), y = (int)((i / (size_t)W) % (size_t)H), z = (int)(i / ((size_t)W * H));
    int pos = axis == 0 ? x : (axis == 1 ? y : z);
    int dim = axis == 0 ? W : (axis == 1 ? H : D);
    long long stride = (axis == 0 ? 1 : (axis == 1 ? (long long)W : (long long)W * H)) * dir;
    int best = src ? (int)src[i] : (vox[i] != 0 ? 0 : cap + 1);
    for (int t = 1; t < best; t++) {
        int q = pos + t * dir;
        int val = (q < 0 || q >= dim) ? 0
                                      : (src ? (int)src[(long long)i + t * stride] : (vox[(long long)i + t * stride] != 0 ? 0 : cap + 1));
        int m = val > t ? val : t;
        best = best < m ? best : m;
    }
    size_t o = i;
    if (padded) {                                            // final pass: into the zero-bordered field (vrt_traverse.h df_index)
        o = (size_t)(x + 1) + ((size_t)(y + 1) + (size_t)(z + 1) * ((size_t)H + 2u)) * ((size_t)W + 2u);
    }
    dst[o] = (uint8_t)(best > cap ? cap : best);
}

// df: 8 * stride bytes (stride = df_field_bytes: one zero-bordered field); tmp0/tmp1: W*H*D bytes each
hipError_t launch_build_df(const uint8_t* vox, int W, int H, int D, uint8_t* df, size_t stride, uint8_t* tmp0, uint8_t* tmp1, hipStream_t s, int cap)
{
    if (cap <= 0) cap = VRT_DF_CAP;
    size_t n = (size_t)W * H * D;
    unsigned blocks = (unsigned)((n + 255) / 256);
    for (int o = 0; o < 8; o++) {
        int sx = (o & 1) ? 1 : -1, sy = (o & 2) ? 1 : -1, sz = (o & 4) ? 1 : -1;
        hipLaunchKernelGGL(k_df_pass, dim3(blocks), dim3(256), 0, s, vox, (const uint8_t*)nullptr, tmp0, W, H, D, 0, sx, 0, cap);
        hipLaunchKernelGGL(k_df_pass, dim3(blocks), dim3(256), 0, s, vox, (const uint8_t*)tmp0, tmp1, W, H, D, 1, sy, 0, cap);
        hipLaunchKernelGGL(k_df_pass, dim3(blocks), dim3(256), 0, s, vox, (const uint8_t*)tmp1, df + (size_t)o * stride, W, H, D, 2, sz, 1, cap);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// open cells.  A ray only ever moves towards the signs of its direction, so from voxel p it can only meet voxels of the box
// between p and the volume's corner in its octant.  Where that whole box is empty the ray is a miss, whatever it would still
// walk through: the octant's field holds 0 there -- the code of "the march ends here", as at a solid voxel and in the border;
// the voxel id read at the same index (0) then says miss.  Hits are untouched (a ray that hits never stands on such a cell);
// what a miss leaves behind does not depend on where it left the volume (traceRay, frag:176-196: material, position and
// normal of a miss are 0) -- only the NUMBER of iterations does, which the count planes report: those are rendered through a
// copy of the fields without open cells (vrt_api.hip).
//   open(p) = AND over a, b, c >= 0 of empty(p + (a sx, b sy, c sz)): three one-sided AND scans.
// ---------------------------------------------------------------------------------------------

// scan along y (axis 1; blockIdx.y = z) or z (axis 2; blockIdx.y = y): one thread per line, x across the threads
__global__ __launch_bounds__(256) void k_open_scan(const uint8_t* __restrict__ vox, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                   int W, int H, int D, int axis, int dir)
{
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (x >= W) return;
    const int len = axis == 1 ? H : D;
    const size_t step = axis == 1 ? (size_t)W : (size_t)W * (size_t)H;
    const size_t base = (size_t)x + (axis == 1 ? (size_t)blockIdx.y * (size_t)W * (size_t)H : (size_t)blockIdx.y * (size_t)W);
    uint8_t flag = 1;
    for (int t = 0; t < len; t++) {                            // from the far end of the line towards the near one
        const size_t i = base + (size_t)(dir > 0 ? len - 1 - t : t) * step;
        flag &= src ? src[i] : (uint8_t)(vox[i] == 0);
        dst[i] = flag;
    }
}

// scan along x, one wave per line, and the result: 0 into the octant's zero-bordered field where the cell is open
__global__ __launch_bounds__(256) void k_open_x(const uint8_t* __restrict__ src, uint8_t* __restrict__ field, int W, int H, int D, int dir, int mark)
{
    const int lane = (int)(threadIdx.x & 63u);
    const size_t line = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (line >= (size_t)H * (size_t)D) return;                 // wave-uniform
    const int y = (int)(line % (size_t)H), z = (int)(line / (size_t)H);
    const uint8_t* row = src + line * (size_t)W;
    uint8_t* out = field + 1 + ((size_t)(y + 1) + (size_t)(z + 1) * ((size_t)H + 2u)) * ((size_t)W + 2u);
    bool carry = true;
    const int chunks = (W + 63) / 64;
    for (int c = 0; c < chunks; c++) {
        const int x = (dir > 0 ? chunks - 1 - c : c) * 64 + lane;
        const bool f = x < W ? row[x] != 0 : true;
        const uint64_t blocked = ~__ballot(f);
        const bool open = carry && f && (dir > 0 ? (blocked >> lane) == 0ull : (blocked << (63 - lane)) == 0ull);
        if (x < W && open) out[x] = mark ? (uint8_t)(out[x] | (uint8_t)mark) : (uint8_t)0;   // bricks: bit 7; voxels: the code 0
        carry = carry && blocked == 0ull;
    }
}

hipError_t launch_open_cells(const uint8_t* vox, int W, int H, int D, uint8_t* df, size_t stride, uint8_t* tmp0, uint8_t* tmp1, hipStream_t s, int mark)
{
    const unsigned bx = (unsigned)((W + 255) / 256);
    const size_t lines = (size_t)H * (size_t)D;
    for (int o = 0; o < 8; o++) {
        const int sx = (o & 1) ? 1 : -1, sy = (o & 2) ? 1 : -1, sz = (o & 4) ? 1 : -1;
        hipLaunchKernelGGL(k_open_scan, dim3(bx, (unsigned)D), dim3(256), 0, s, vox, (const uint8_t*)nullptr, tmp0, W, H, D, 1, sy);
        hipLaunchKernelGGL(k_open_scan, dim3(bx, (unsigned)H), dim3(256), 0, s, vox, (const uint8_t*)tmp0, tmp1, W, H, D, 2, sz);
        hipLaunchKernelGGL(k_open_x, dim3((unsigned)((lines + 3) / 4)), dim3(256), 0, s, (const uint8_t*)tmp1, df + (size_t)o * stride, W, H, D, sx, mark);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// brick scenes (vrt_scene_from_bricks): padded pointer grid, brick occupancy, per-voxel clearance of the occupied bricks
// ---------------------------------------------------------------------------------------------

// grid (nbx * nby * nbz) -> interior of the padded grid ((nbx+2)(nby+2)(nbz+2); its border was preset to 0xFFFFFFFF = outside
// the volume) and one byte per brick: occupied or not
__global__ __launch_bounds__(256) void k_brick_grid(const uint32_t* __restrict__ grid, int nbx, int nby, int nbz,
                                                    uint32_t* __restrict__ padded, uint8_t* __restrict__ occ)
{
    const size_t n = (size_t)nbx * nby * nbz, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % (size_t)nbx), y = (int)((i / (size_t)nbx) % (size_t)nby), z = (int)(i / ((size_t)nbx * nby));
    const uint32_t g = grid[i];
    padded[(size_t)(x + 1) + ((size_t)(y + 1) + (size_t)(z + 1) * ((size_t)nby + 2u)) * ((size_t)nbx + 2u)] = g;
    occ[i] = g != 0u ? 1 : 0;
}

// One workgroup per occupied brick: the clearance of each of its voxels in each octant, looking through the 26 neighbours
// (24^3 voxels in LDS; beyond them -- and outside the volume -- counts as solid, so values reach 9..16).  Per octant the
// three one-sided min-max passes of k_df_pass, restricted to the cells the centre brick's results depend on.
#define VRT_FINE_CAP 16
__global__ __launch_bounds__(256) void k_brick_fine(const uint32_t* __restrict__ padded, int pbx, int pby, const uint32_t* __restrict__ coord,
                                                    const uint8_t* __restrict__ pool, uint8_t* __restrict__ fine)
{
    __shared__ uint8_t A[24 * 24 * 24], B[24 * 24 * 24];
    const uint32_t b = blockIdx.x;                             // pool index
    const uint32_t pc = coord[b];                              // index of the brick in the padded grid
    const int cbx = (int)(pc % (uint32_t)pbx), cby = (int)((pc / (uint32_t)pbx) % (uint32_t)pby), cbz = (int)(pc / ((uint32_t)pbx * (uint32_t)pby));
    __shared__ uint32_t nb[27];                               // the 3 x 3 x 3 bricks around it: 0 empty, 0xFFFFFFFF outside the volume
    if (threadIdx.x < 27) {
        const int dx = (int)threadIdx.x % 3 - 1, dy = ((int)threadIdx.x / 3) % 3 - 1, dz = (int)threadIdx.x / 9 - 1;
        nb[threadIdx.x] = padded[(size_t)(cbx + dx) + ((size_t)(cby + dy) + (size_t)(cbz + dz) * (size_t)pby) * (size_t)pbx];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 24 * 24 * 24; t += 256) {
        const int x = t % 24, y = (t / 24) % 24, z = t / 576;
        const int k = (x >> 3) + (y >> 3) * 3 + (z >> 3) * 9;
        const uint32_t ptr = nb[k];
        uint8_t solid;
        if (ptr == 0xFFFFFFFFu) solid = 1;                     // outside the volume
        else if (ptr == 0u) solid = 0;
        else solid = pool[(size_t)(ptr - 1u) * 512u + (size_t)((x & 7) + (y & 7) * 8 + (z & 7) * 64)] != 0 ? 1 : 0;
        A[t] = solid ? 0 : VRT_FINE_CAP + 1;
    }
    __syncthreads();
    for (int o = 0; o < 8; o++) {
        const int sx = (o & 1) ? 1 : -1, sy = (o & 2) ? 1 : -1, sz = (o & 4) ? 1 : -1;
        // pass x: centre columns, every y and z     A -> B
        for (int t = threadIdx.x; t < 8 * 24 * 24; t += 256) {
            const int x = 8 + (t & 7), y = (t >> 3) % 24, z = (t >> 3) / 24;
            const int i = x + y * 24 + z * 576;
            int best = A[i];
            for (int k = 1; k < best; k++) {
                const int q = x + k * sx;
                const int val = (q < 0 || q >= 24) ? 0 : (int)A[i + k * sx];
                const int m = val > k ? val : k;
                best = best < m ? best : m;
            }
            B[i] = (uint8_t)best;
        }
        __syncthreads();
        // pass y: centre columns and rows, every z; the results go to the x-columns 0..7 of B, which this pass does not read
        for (int t = threadIdx.x; t < 8 * 8 * 24; t += 256) {
            const int x = 8 + (t & 7), y = 8 + ((t >> 3) & 7), z = t >> 6;
            const int i = x + y * 24 + z * 576;
            int best = B[i];
            for (int k = 1; k < best; k++) {
                const int q = y + k * sy;
                const int val = (q < 0 || q >= 24) ? 0 : (int)B[i + k * sy * 24];
                const int m = val > k ? val : k;
                best = best < m ? best : m;
            }
            B[(x - 8) + y * 24 + z * 576] = (uint8_t)best;
        }
        __syncthreads();
        // pass z: the centre brick
        for (int t = threadIdx.x; t < 512; t += 256) {
            const int lx = t & 7, ly = (t >> 3) & 7, lz = t >> 6;
            const int z = 8 + lz;
            const int i = lx + (8 + ly) * 24 + z * 576;
            int best = B[i];
            for (int k = 1; k < best; k++) {
                const int q = z + k * sz;
                const int val = (q < 0 || q >= 24) ? 0 : (int)B[i + k * sz * 576];
                const int m = val > k ? val : k;
                best = best < m ? best : m;
            }
            fine[((size_t)b * 8u + (size_t)o) * 512u + (size_t)t] = (uint8_t)(best > VRT_FINE_CAP ? VRT_FINE_CAP : best);
        }
        __syncthreads();
    }
}

// the padded pointer grid and the eight coarse fields folded into the one word per brick the march reads (brick_entry_pack)
__global__ __launch_bounds__(256) void k_brick_pack(const uint32_t* __restrict__ padded, const uint8_t* __restrict__ coarse, size_t cstride,
                                                    size_t npad, uint64_t* __restrict__ entry)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    uint8_t c8[8];
#pragma unroll
    for (int o = 0; o < 8; o++) c8[o] = coarse[(size_t)o * cstride + i];
    entry[i] = brick_entry_pack(padded[i], c8);
}

hipError_t launch_brick_pack(const uint32_t* padded, const uint8_t* coarse, size_t cstride, size_t npad, uint64_t* entry, hipStream_t s)
{
    hipLaunchKernelGGL(k_brick_pack, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, s, padded, coarse, cstride, npad, entry);
    return hipGetLastError();
}

hipError_t launch_brick_grid(const uint32_t* grid, int nbx, int nby, int nbz, uint32_t* padded, uint8_t* occ, hipStream_t s)
{
    const size_t n = (size_t)nbx * nby * nbz;
    hipLaunchKernelGGL(k_brick_grid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, grid, nbx, nby, nbz, padded, occ);
    return hipGetLastError();
}

hipError_t launch_brick_fine(const uint32_t* padded, int pbx, int pby, const uint32_t* coord, uint32_t n_bricks, const uint8_t* pool,
                             uint8_t* fine, hipStream_t s)
{
    if (n_bricks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_brick_fine, dim3(n_bricks), dim3(256), 0, s, padded, pbx, pby, coord, pool, fine);
    return hipGetLastError();
}

// field 8 of the clearance allocation: the voxel ids in the fields' zero-bordered layout (trace_df_fast reads the id of a hit
// at the index it already has); the border stays 0
__global__ __launch_bounds__(256) void k_pad_vox(const uint8_t* __restrict__ vox, uint8_t* __restrict__ dst, int W, int H, int D)
{
    size_t n = (size_t)W * H * D;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x = (int)(i % (size_t)W), y = (int)((i / (size_t)W) % (size_t)H), z = (int)(i / ((size_t)W * H));
    dst[(size_t)(x + 1) + ((size_t)(y + 1) + (size_t)(z + 1) * ((size_t)H + 2u)) * ((size_t)W + 2u)] = vox[i];
}

hipError_t launch_pad_vox(const uint8_t* vox, int W, int H, int D, uint8_t* dst, hipStream_t s)
{
    size_t n = (size_t)W * H * D;
    hipLaunchKernelGGL(k_pad_vox, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, vox, dst, W, H, D);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// traversal
// ---------------------------------------------------------------------------------------------

struct RayHit {            // RayHit, voxel_volume.frag:43-49
    uint32_t material;
    f3 pos, normal, dir;
    uint32_t ncode;        // which of the 26 face / edge / corner normals `normal` is: mask | (sx<0)<<3 | (sy<0)<<4 | (sz<0)<<5;
                           // 0xFFFFFFFF: none of them (the zero vector of rule A, or a masked axis the ray does not move along)
};

// Occupancy summaries as seen by a workgroup: LDS copies when they fit (typed address_space(3) pointers, so
// that the lookups compile to ds_read_b64 and not to flat loads), the L2-resident originals otherwise.
typedef const __attribute__((address_space(3))) uint64_t* lds_u64_ptr;
template <bool LDS> struct OccT;
template <> struct OccT<true>  { lds_u64_ptr o2, o3; };
template <> struct OccT<false> { const uint64_t* o2; const uint64_t* o3; };

__device__ __forceinline__ f3 hit_normal(uint32_t mask, int sx, int sy, int sz)
{
    // normalize(-mask * rayStep) (frag:190): the vector has k = popcount(mask) components of +-1, so its length is
    // RN(sqrt(k)) and every non-zero component is +-RN(1 / RN(sqrt(k))) -- three constants instead of a square root
    // and three IEEE divisions (k = 0: the zero vector, canonical rule A).  A masked axis with rayStep = 0 (possible only
    // through rule A's initial mask) changes k's meaning; that case keeps the general form.
    const uint32_t k = __builtin_popcount(mask & 7u);
    const float c = k == 1u ? 1.0f : (k == 2u ? __uint_as_float(0x3f3504f3u) : __uint_as_float(0x3f13cd3au));
    const bool general = ((mask & 1u) && sx == 0) || ((mask & 2u) && sy == 0) || ((mask & 4u) && sz == 0);
    f3 n = mk3((mask & 1u) ? (float)(-sx) : 0.0f, (mask & 2u) ? (float)(-sy) : 0.0f, (mask & 4u) ? (float)(-sz) : 0.0f);
    if (general) return normalize3(n);
    return mk3(n.x * c, n.y * c, n.z * c);
}

// traceRay, voxel_volume.frag:176-196
template <int TRAV, class Occ, bool AHEAD = false, bool PF = false>
__device__ __forceinline__ void trace_ray(const DevScene& s, const Occ occ, f3 start, f3 dir,
                                          uint32_t maxSteps, RayHit& h, RayInt& r)
{
    trace_int<TRAV, decltype(occ.o2), AHEAD, false, PF>(s.vol, occ.o2, occ.o3, start, dir, maxSteps, r);
    h.material = r.material;
    h.dir = dir;
    // values first, one assignment to h afterwards: stores to h from both sides of the branch were being merged into
    // address-selected scratch stores (28 B of scratch per lane, which also slows the wave launch)
    f3 pos = mk3(0.0f, 0.0f, 0.0f), nrm = mk3(0.0f, 0.0f, 0.0f);
    uint32_t ncode = 0xFFFFFFFFu;
    if (r.material != 0) {
        nrm = hit_normal(r.mask, r.sx, r.sy, r.sz);
        const bool general = (r.mask & 7u) == 0u || ((r.mask & 1u) && r.sx == 0) || ((r.mask & 2u) && r.sy == 0) || ((r.mask & 4u) && r.sz == 0);
        if (!general) ncode = (r.mask & 7u) | ((uint32_t)(r.sx < 0) << 3) | ((uint32_t)(r.sy < 0) << 4) | ((uint32_t)(r.sz < 0) << 5);
        f3 m = mk3((r.mask & 1u) ? (r.side.x - r.delta.x) : 0.0f,
                   (r.mask & 2u) ? (r.side.y - r.delta.y) : 0.0f,
                   (r.mask & 4u) ? (r.side.z - r.delta.z) : 0.0f);
        float d = len3(m);
        pos = mk3(r.pos.x + d * dir.x, r.pos.y + d * dir.y, r.pos.z + d * dir.z);
    }
    h.pos = pos;
    h.normal = nrm;
    h.ncode = ncode;
}

// ---------------------------------------------------------------------------------------------
// shading helpers
// ---------------------------------------------------------------------------------------------

// UNORM8 / SNORM8 code -> float: q0 = c * r, q = fma(fma(-D, q0, c), r, q0) with r = RN(1 / D) equals the IEEE quotient c / D
// for every one of the 256 codes (checked exhaustively in tests/test_denoise_decode.py): 3 VALU ops instead of ~11.
__device__ __forceinline__ float decode_unorm8(uint32_t c)
{
    const float r = 1.0f / 255.0f;
    float cf = (float)c, q0 = cf * r;
    return __builtin_fmaf(__builtin_fmaf(-255.0f, q0, cf), r, q0);
}
__device__ __forceinline__ float decode_snorm8(int32_t c)
{
    const float r = 1.0f / 127.0f;
    float cf = (float)c, q0 = cf * r;
    return fmaxf(__builtin_fmaf(__builtin_fmaf(-127.0f, q0, cf), r, q0), -1.0f);
}

// pc: the push block of the pixel's frame; noise: the pixel's blue-noise texel, decoded on first use (it is the same for
// every AO sample and every bounce of the pixel)
// (kernels of VRT_TRAVERSAL_DF_FAST never fill iteration-count planes -- vrt_api.hip sends every launch that has them to the counting twins,
// VRT_TRAVERSAL_DF_FAST_CNT -- so for them `fetches` is dead and the compiler drops it: VRT_COUNTS(TRAV))
#define VRT_COUNTS(TRAV) ((TRAV) != VRT_TRAVERSAL_DF_FAST && (TRAV) != VRT_TRAVERSAL_BRICK)
struct PixCtx { int px, py; uint32_t fetches, rays; const vrt_push* pc; f3 noise;
                uint32_t ldsw; };    // ldsw: byte address of the wave's VRT_AO_SLOT bytes of LDS (df_ao_pool_loop): kernels that trace AO rays through the hand-written loop

// skyColor, voxel_volume.frag:98-105
__device__ __forceinline__ f3 sky_color(const DevScene& s, f3 d)
{
    float u = atan2_spec(d.z, d.x) * 0.1591f + 0.5f;
    float v = asin_spec(-d.y) * 0.3183f + 0.5f;
    uint32_t x = wrap_texel(u, s.sky_w), y = wrap_texel(v, s.sky_h);
    const float4 t = reinterpret_cast<const float4*>(s.sky)[(size_t)y * s.sky_w + x];
    return mk3(t.x, t.y, t.z);
}

// skyColor of every normal a hit can have, by sky_color itself (so that the table holds bit for bit what the shading code
// would compute); entry = mask | (sx<0)<<3 | (sy<0)<<4 | (sz<0)<<5, 64 x float4
__global__ void k_sky_normals(const DevScene s, float4* table)
{
    const uint32_t code = threadIdx.x & 63u, mask = code & 7u;
    const int sx = (code & 8u) ? -1 : 1, sy = (code & 16u) ? -1 : 1, sz = (code & 32u) ? -1 : 1;
    f3 c = mk3(0.0f, 0.0f, 0.0f);
    if (mask != 0u) c = sky_color(s, hit_normal(mask, sx, sy, sz));
    table[code] = make_float4(c.x, c.y, c.z, 0.0f);
}

hipError_t launch_sky_normals(const DevScene& sc, float* table, hipStream_t s)
{
    hipLaunchKernelGGL(k_sky_normals, dim3(1), dim3(64), 0, s, sc, reinterpret_cast<float4*>(table));
    return hipGetLastError();
}

// the sky as the colour target stores it: unorm8 of r, g, b per texel (a = 0, canonical rule F)
__global__ __launch_bounds__(256) void k_sky_rgba8(const float4* __restrict__ sky, uint32_t* __restrict__ sky8, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 t = sky[i];
    sky8[i] = (uint32_t)unorm8(t.x) | ((uint32_t)unorm8(t.y) << 8) | ((uint32_t)unorm8(t.z) << 16);
}

hipError_t launch_sky_rgba8(const float* sky, uint32_t* sky8, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_sky_rgba8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(sky), sky8, n);
    return hipGetLastError();
}

// Diagnostic (vrt_debug_sky_texels): for n unnormalised directions the sky texel by the numeric spec (sky_color's own
// arithmetic) and by the fast path, as the hardware computes both: out[4i] = spec x | y << 16, [4i+1] = fast x | y << 16,
// [4i+2] = the fast path is sure, [4i+3] = float bits of the fast u * sky_w
__global__ __launch_bounds__(256) void k_debug_sky(const DevScene s, const float* __restrict__ v, size_t n, uint32_t* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float vx = v[3 * i], vy = v[3 * i + 1], vz = v[3 * i + 2];
    const f3 d = normalize3(mk3(vx, vy, vz));
    const float u = atan2_spec(d.z, d.x) * 0.1591f + 0.5f;
    const float w = asin_spec(-d.y) * 0.3183f + 0.5f;
    const uint32_t sx = wrap_texel(u, s.sky_w), sy = wrap_texel(w, s.sky_h);
    uint32_t tx = 0u, ty = 0u;
    float un = 0.0f, vn = 0.0f;
    const bool sure = s.skyk.w != 0u && sky_texel_fast(vx, vy, vz, s.skyk, tx, ty, un, vn);
    out[4 * i] = sx | (sy << 16); out[4 * i + 1] = tx | (ty << 16); out[4 * i + 2] = sure ? 1u : 0u; out[4 * i + 3] = __float_as_uint(un);
}

// development build (-DVRT_TRACE_COUNTERS): the brick march's look-up counters, read and reset
hipError_t debug_brick_counts(unsigned long long out[4])
{
#if defined(VRT_TRACE_COUNTERS)
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vrt_brick_counts), 32);
    if (e != hipSuccess) return e;
    const unsigned long long zero[4] = {0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_vrt_brick_counts), zero, 32);
#else
    out[0] = out[1] = out[2] = out[3] = 0ull;
    return hipSuccess;
#endif
}

hipError_t launch_debug_sky(const DevScene& sc, const float* v, size_t n, uint32_t* out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_debug_sky, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sc, v, n, out);
    return hipGetLastError();
}

// fragmentNoiseSeq + randomDir, voxel_volume.frag:80-95
__device__ __forceinline__ f3 random_dir(const DevScene& s, const vrt_push& pc, PixCtx& c, uint32_t num)
{
    uint32_t offset = num * 32u + pc.frame % 32u;
    const float g = 1.22074408460575947536f;
    const float a0 = 1.0f / g, a1 = 1.0f / (g * g), a2 = 1.0f / ((g * g) * g);
    {   // (the pixel's blue-noise texel is fetched anew for every sample: kept across the traces it cost four registers and, through the
        // branch around the fetch, a second copy of everything after it -- 1 350 instructions of the megakernel)
        float pxf = ((float)c.px + 0.5f) / 512.0f + 0.5f;
        float pyf = ((float)c.py + 0.5f) / 512.0f + 0.5f;
        uint32_t tx = wrap_texel(pxf, s.noise_w), ty = wrap_texel(pyf, s.noise_h);
        const uchar4 t = reinterpret_cast<const uchar4*>(s.noise)[(size_t)ty * s.noise_w + tx];
        c.noise = mk3(decode_unorm8(t.x), decode_unorm8(t.y), decode_unorm8(t.z));          // = t / 255.0f, exactly
    }
    float fo = (float)offset;
    float n0 = c.noise.x + fo * a0;
    float n1 = c.noise.y + fo * a1;
    float n2 = c.noise.z + fo * a2;
    n0 = n0 - floorf(n0); n1 = n1 - floorf(n1); n2 = n2 - floorf(n2);
    return normalize3(mk3(n0 * 2.0f - 1.0f, n1 * 2.0f - 1.0f, n2 * 2.0f - 1.0f));
}

// main() ray generation, voxel_volume.frag:312-322 (+ screen_quad.vert:18-31)
__device__ __forceinline__ f3 primary_dir(const FrameSlot& S, int px, int py)
{
    const RayGenConsts& g = S.rg;
    float sx = (((float)px + 0.5f) / g.W) * 2.0f - 1.0f;
    float sy = (((float)py + 0.5f) / g.H) * 2.0f - 1.0f;
    float vx = ((g.cd.x + sx * S.pc.cam_right[0]) + sy * g.planeV.x) + g.jx;
    float vy = ((g.cd.y + sx * S.pc.cam_right[1]) + sy * g.planeV.y) + g.jy;
    float vz = ((g.cd.z + sx * S.pc.cam_right[2]) + sy * g.planeV.z) + 0.0f;
    return normalize3(mk3(vx, vy, vz));
}

// primary_dir with fewer instructions (K1; an IEEE division expands to ~11 VALU instructions, and ray generation has five):
//  * the two screen divisions by the 3-instruction sequence of screen_div_fast when the host found it exact for every
//    pixel centre of this screen size (P.fast_screen_div);
//  * normalize: the three divisions by the length share the reciprocal.  The instructions are those of the compiler's
//    expansion of x / l without v_div_scale / v_div_fixup, which do nothing when numerator and denominator are normal
//    numbers whose exponents differ by less than 96 and whose quotient is normal: guaranteed here for a whole wave by
//    2^-40 <= min |component| and length <= 2^40 (a component never exceeds the length by more than rounding).  Any other
//    wave -- zero components, huge or tiny camera vectors, NaN -- takes normalize3.
__device__ __forceinline__ f3 primary_v(const RayGenConsts& g, float crx, float cry, float crz, float rcp_w, float rcp_h,
                                        int fast_screen_div, int px, int py)
{
    float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    float qx, qy;
    if (fast_screen_div) { qx = screen_div_fast(fx, g.W, rcp_w); qy = screen_div_fast(fy, g.H, rcp_h); }
    else                 { qx = fx / g.W; qy = fy / g.H; }
    float sx = qx * 2.0f - 1.0f;
    float sy = qy * 2.0f - 1.0f;
    float vx = ((g.cd.x + sx * crx) + sy * g.planeV.x) + g.jx;
    float vy = ((g.cd.y + sx * cry) + sy * g.planeV.y) + g.jy;
    float vz = ((g.cd.z + sx * crz) + sy * g.planeV.z) + 0.0f;
    return mk3(vx, vy, vz);
}
// ... and its normalize()
__device__ __forceinline__ f3 primary_normalize(const f3 v)
{
    const float vx = v.x, vy = v.y, vz = v.z;
    const float l = len3(v);
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(vx), __builtin_fabsf(vy)), __builtin_fabsf(vz));
    const bool tame = lo >= 0x1p-40f && l <= 0x1p40f;
    if (__ballot(!tame) != 0ull) return normalize3(v);
    const float r0 = __builtin_amdgcn_rcpf(l);
    const float r = __builtin_fmaf(__builtin_fmaf(-l, r0, 1.0f), r0, r0);
    f3 o;
    { float q = vx * r; q = __builtin_fmaf(__builtin_fmaf(-l, q, vx), r, q); o.x = __builtin_fmaf(__builtin_fmaf(-l, q, vx), r, q); }
    { float q = vy * r; q = __builtin_fmaf(__builtin_fmaf(-l, q, vy), r, q); o.y = __builtin_fmaf(__builtin_fmaf(-l, q, vy), r, q); }
    { float q = vz * r; q = __builtin_fmaf(__builtin_fmaf(-l, q, vz), r, q); o.z = __builtin_fmaf(__builtin_fmaf(-l, q, vz), r, q); }
    return o;
}

// calcAmbient + isShadowed + color + colorHit, voxel_volume.frag:205-264, in two halves: the secondary rays of a hit (what they
// find: how many AO rays hit something, whether the light is hidden) and the arithmetic on what they found.  color_hit is the two
// one after the other; the packed bounce chain (color_main_ray_packed) runs the first half on the way out and the second on the
// way back.
// SEC = false: the host has established ao_samples == 0 and shadows == 0 (K1 MODE 1), so neither loop is compiled in.
// active: the lane has a hit whose secondary rays are wanted.  Where the AO rays go through the wave's pool the function must be reached
// by the wave's other lanes as well (wave-uniform control flow at the call site): a lane without a hit of its own has no rays in
// the pool but takes rays from it like everybody else -- the pixels of a block's silhouette, and the few metallic pixels of a bounce,
// get the whole wave's help.  (Called from divergent code the pool simply serves the lanes that are there.)
template <int TRAV, class Occ, bool SEC = true>
__device__ __forceinline__ void secondary_rays(const GeomParams& P, const Occ occ, PixCtx& c, const f3 pos, const f3 normal, uint32_t depth,
                                               float& ambient, uint32_t& ao_hits, bool& shadowed, const bool active = true)
{
    const DevScene& s = P.sc;
    const vrt_settings& st = P.st;
    ambient = 0.0f; ao_hits = 0u;
    constexpr bool kBatch = TRAV == VRT_TRAVERSAL_DF_FAST || TRAV == VRT_TRAVERSAL_DF_FAST_CNT;
    if (!SEC || st.ao_samples == 0) {
        ambient = 1.0f;
    } else if (kBatch && __builtin_amdgcn_readfirstlane((int)s.vol.ao_batch) != 0) {
        // the hand-written loop: the AO rays of the wave's pixels from a pool in LDS that every lane draws on (df_ao_pool_loop) --
        // sample after sample each lane writes its pixel's ray into its column, and whichever lane is free traces it and reports
        // to the column's counter; which lane traces a ray changes nothing about what the ray finds
        constexpr bool kCnt = TRAV == VRT_TRAVERSAL_DF_FAST_CNT;
        const uint32_t ldsw = c.ldsw;
        const uint64_t act = __ballot(active);
        const uint32_t col = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
        __attribute__((address_space(3))) uint32_t* cnt = (__attribute__((address_space(3))) uint32_t*)(uintptr_t)(ldsw + 3072u + col * 4u);
        if (active) { cnt[0] = 0u; if (kCnt) cnt[64] = 0u; }
        // the fields as the loops address them (offsets count from one slice in front of field 0)
        const uint8_t* const fld = s.vol.df - (size_t)(s.vol.W + 2) * (size_t)(s.vol.H + 2);
        AoLane lane;
        ao_lane_rest(s.vol, lane);
        uint32_t next = 0u, looks = 0u, direct_hits = 0u, direct_fet = 0u;
        for (uint32_t i = 0; i < st.ao_samples; i++) {
            // The OWNER looks at its ray's first voxel itself -- every lane at once, where in the pool a ray's first look is a round of
            // the loop like any other: a ray in the open (the clearance covers its budget) and a ray that starts on a 0 byte are
            // decided here and never enter the pool; the others bring their first clearance with them and are marched from the round
            // they are taken up in.  The rays that will creep (clearance 1 or 2) wait in FRONT of the pool: the longest rays of a
            // sample start first, which is what the end of the AO phase waits for.
            bool store = false;
            uint32_t c0 = 0u;
            AoRay a;
            if (active) {
                f3 rd = random_dir(s, *c.pc, c, i + depth * st.ao_samples);
                f3 dir = mk3(normal.x + rd.x, normal.y + rd.y, normal.z + rd.z);
                f3 o = mk3(pos.x + dir.x * 0.01f, pos.y + dir.y * 0.01f, pos.z + dir.z * 0.01f);
                ao_ray_setup(s.vol, o, dir, a);
                c0 = fld[a.idx0];
                if (kCnt) looks += 1u;
                if (c0 == 0u) {                                // solid, border or open cell: the voxel id says which (frag:157 at iteration 0)
                    const uint32_t id = fld[a.idx0 + a.voxoff];
                    if (id != 0u) direct_hits++;
                    if (kCnt) { looks += 1u; direct_fet += id != 0u ? 1u : 0u; }
                } else if (c0 >= st.ao_steps) {                // nothing but empty voxels until the budget ends: a miss
                    if (kCnt) direct_fet += s.vol.count_marched != 0u ? 0u : st.ao_steps;
                } else store = true;
            }
            const bool creeps = store && c0 <= 2u;
            const uint64_t mc = __ballot(creeps), mo = __ballot(store && !creeps);
            const uint32_t nc = (uint32_t)__builtin_popcountll(mc);
            if (store) {
                const uint32_t slot = creeps ? __builtin_amdgcn_mbcnt_hi((uint32_t)(mc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mc, 0u))
                                             : nc + __builtin_amdgcn_mbcnt_hi((uint32_t)(mo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mo, 0u));
                ao_ray_store(ldsw, slot, a, col | (c0 << 8));
            }
            next = 0u;
            trace_ao_pool<kCnt>(s.vol, lane, ldsw, nc + (uint32_t)__builtin_popcountll(mo), i + 1u < st.ao_samples ? 1u : 0u, next, st.ao_steps, looks);
        }
        if (active) {
            ao_hits = cnt[0] + direct_hits;
            c.rays += st.ao_samples;
        }
        // (look-ups are counted by the lane that makes them, iterations for the pixel the ray belongs to)
        if (kCnt) c.fetches += s.vol.count_lookups != 0u ? looks : (active ? cnt[64] + direct_fet : 0u);
        // calcAmbient's sum (frag:219-222): one addition of 1 / aoSamples per ray that hit -- the value depends on their number only
        float sample_frac = 1.0f / (float)st.ao_samples;
        for (uint32_t q = 0; q < ao_hits; q++) ambient += sample_frac;
    } else if ((TRAV == VRT_TRAVERSAL_BRICK || TRAV == VRT_TRAVERSAL_BRICK_CNT) && __builtin_amdgcn_readfirstlane((int)s.vol.ao_batch) != 0) {
        // brick scenes: the same pool in the generic loop (brick_ao_pool)
        constexpr bool kCnt = TRAV == VRT_TRAVERSAL_BRICK_CNT;
        const uint32_t ldsw = c.ldsw;
        const uint64_t act = __ballot(active);
        const uint32_t col = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
        __attribute__((address_space(3))) uint32_t* cnt = (__attribute__((address_space(3))) uint32_t*)(uintptr_t)(ldsw + 3328u + col * 4u);
        if (active) { cnt[0] = 0u; if (kCnt) cnt[64] = 0u; }
        BrickAoLane lane;
        brick_ao_rest(lane);
        uint32_t next = 0u, looks = 0u, direct_hits = 0u, direct_fet = 0u;
        for (uint32_t i = 0; i < st.ao_samples; i++) {
            // (the owner looks at its ray's first voxel itself, as on dense scenes: rays in the open, rays on a 0 byte and rays that never
            // enter the volume are decided here; the rays that will creep wait in front of the pool)
            bool store = false;
            uint32_t c0 = 0u;
            DdaState rs; float gx = 0.0f, gy = 0.0f, gz = 0.0f;
            if (active) {
                f3 rd = random_dir(s, *c.pc, c, i + depth * st.ao_samples);
                f3 dir = mk3(normal.x + rd.x, normal.y + rd.y, normal.z + rd.z);
                f3 o = mk3(pos.x + dir.x * 0.01f, pos.y + dir.y * 0.01f, pos.z + dir.z * 0.01f);
                brick_ao_setup(s.vol, o, dir, rs, gx, gy, gz);
                if (!oob(s.vol, rs.mx, rs.my, rs.mz)) {          // (else: starts outside and misses the volume: leaves in iteration 0, no fetch)
                    const uint32_t oct = (uint32_t)(rs.sx > 0) | ((uint32_t)(rs.sy > 0) << 1) | ((uint32_t)(rs.sz > 0) << 2);
                    uint32_t m = 0u;
                    c0 = brick_clear(s.vol, rs.mx, rs.my, rs.mz, oct, rs.sx, rs.sy, rs.sz, m, kCnt ? &looks : nullptr);
                    if (c0 == 0u) { if (m != 0u) direct_hits++; if (kCnt) direct_fet += m != 0u ? 1u : 0u; }
                    else if (c0 >= st.ao_steps) { if (kCnt) direct_fet += s.vol.count_marched != 0u ? 0u : st.ao_steps; }
                    else store = true;
                }
            }
            const bool creeps = store && c0 <= 2u;
            const uint64_t mc = __ballot(creeps), mo = __ballot(store && !creeps);
            const uint32_t nc = (uint32_t)__builtin_popcountll(mc);
            if (store) {
                const uint32_t slot = creeps ? __builtin_amdgcn_mbcnt_hi((uint32_t)(mc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mc, 0u))
                                             : nc + __builtin_amdgcn_mbcnt_hi((uint32_t)(mo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mo, 0u));
                brick_ao_store(ldsw, slot, rs, gx, gy, gz, col | (c0 << 8));
            }
            next = 0u;
            brick_ao_pool<kCnt>(s.vol, lane, ldsw, nc + (uint32_t)__builtin_popcountll(mo), i + 1u < st.ao_samples, next, st.ao_steps, looks);
        }
        if (active) {
            ao_hits = cnt[0] + direct_hits;
            c.rays += st.ao_samples;
        }
        if (kCnt) c.fetches += s.vol.count_lookups != 0u ? looks : (active ? cnt[64] + direct_fet : 0u);
        float sample_frac = 1.0f / (float)st.ao_samples;
        for (uint32_t q = 0; q < ao_hits; q++) ambient += sample_frac;
    } else if (active) {
        float sample_frac = 1.0f / (float)st.ao_samples;
        for (uint32_t i = 0; i < st.ao_samples; i++) {
            f3 rd = random_dir(s, *c.pc, c, i + depth * st.ao_samples);
            f3 dir = mk3(normal.x + rd.x, normal.y + rd.y, normal.z + rd.z);
            f3 o = mk3(pos.x + dir.x * 0.01f, pos.y + dir.y * 0.01f, pos.z + dir.z * 0.01f);
            RayInt r;
            // AO rays have a 64-iteration budget: too short for jumps to pay, and budget ties would force re-traces
            // (the hand-written loop's kernels: every lane its own clearance is the batched path above; here the wave's smallest)
            trace_int<((TRAV == VRT_TRAVERSAL_JUMP || TRAV == VRT_TRAVERSAL_DFJ) ? VRT_TRAVERSAL_DF : TRAV), decltype(occ.o2), false, true, false, !kBatch>(s.vol, occ.o2, occ.o3, o, dir, st.ao_steps, r);   // (no prefetch: AO rays point every way, three gathers instead of one measured +18 %)
            if (VRT_COUNTS(TRAV)) c.fetches += r.fetches;
            c.rays++;
            if (r.material != 0) { ambient += sample_frac; ao_hits++; }
        }
    }
    shadowed = false;
    if (SEC && st.shadows && active) {
        f3 L = mk3(st.light_dir[0], st.light_dir[1], st.light_dir[2]);
        f3 o = mk3(pos.x + normal.x * 0.01f, pos.y + normal.y * 0.01f, pos.z + normal.z * 0.01f);
        RayInt r;
        trace_int<TRAV, decltype(occ.o2), false, true, true>(s.vol, occ.o2, occ.o3, o, L, st.max_steps, r);      // traceRayHit: only "did it hit" is used
        if (VRT_COUNTS(TRAV)) c.fetches += r.fetches;
        c.rays++;
        shadowed = r.material != 0;
    }
}

// color() of a hit (frag:236-248) and colorHit's division by depth + 1 (frag:258).  `sky` = skyColor(normal).
__device__ __forceinline__ f3 shade_eval(const GeomParams& P, uint32_t material, const f3 normal, const f3 sky, float ambient, bool shadowed,
                                         f3 reflection, uint32_t depth)
{
    const vrt_settings& st = P.st;
    float k = ambient * st.ambient_intensity;
    f3 amb = mk3(k * sky.x, k * sky.y, k * sky.z);
    f3 L = mk3(st.light_dir[0], st.light_dir[1], st.light_dir[2]);
    f3 diffuse = mk3(0.0f, 0.0f, 0.0f);
    if (!shadowed) {
        float diff = fmaxf(dot3(normal, L), 0.0f);
        diffuse = mk3((diff * st.light_color[0]) * st.light_intensity,
                      (diff * st.light_color[1]) * st.light_intensity,
                      (diff * st.light_color[2]) * st.light_intensity);
    }
    const vrt_material mat = P.sc.palette[material];
    float inv = (float)(depth + 1);
    f3 out;
    out.x = ((((diffuse.x + reflection.x * mat.metallic) + amb.x) * mat.diffuse[0]) * 1.0f) / inv;
    out.y = ((((diffuse.y + reflection.y * mat.metallic) + amb.y) * mat.diffuse[1]) * 1.0f) / inv;
    out.z = ((((diffuse.z + reflection.z * mat.metallic) + amb.z) * mat.diffuse[2]) * 1.0f) / inv;
    return out;
}

// active = false: the lane has nothing to shade and is here for the others' AO rays (secondary_rays); its result is not used
template <int TRAV, class Occ, bool SEC = true>
__device__ f3 color_hit(const GeomParams& P, const Occ occ, PixCtx& c, const RayHit& hit,
                        f3 reflection, uint32_t depth, const bool active = true)
{
    const DevScene& s = P.sc;
    float ambient; uint32_t ao_hits; bool shadowed;
    secondary_rays<TRAV, Occ, SEC>(P, occ, c, hit.pos, hit.normal, depth, ambient, ao_hits, shadowed, active && hit.material != 0);
    if (!active) return mk3(0.0f, 0.0f, 0.0f);
    if (hit.material == 0) return sky_color(s, hit.dir);
    // skyColor(hit.normal): the normal is one of 26 vectors, whose sky texels the scene holds in a table (computed by this very
    // function, k_sky_normals); any other normal is looked up here
    f3 sky;
#if defined(VRT_NO_SKY_TABLE)
    if (false) {
#else
    if (__ballot(hit.ncode == 0xFFFFFFFFu) == 0ull) {
#endif
        const float4 t = reinterpret_cast<const float4*>(s.sky_normals)[hit.ncode];
        sky = mk3(t.x, t.y, t.z);
    } else sky = sky_color(s, hit.normal);
    return shade_eval(P, hit.material, hit.normal, sky, ambient, shadowed, reflection, depth);
}

// colorMainRay, voxel_volume.frag:267-307
// BOUNCE = false: the host has established that no ray of the frame can bounce (max_bounces == 0, or no voxel of the scene has
// a metallic material): the loop and its stack of hits -- 352 bytes of scratch per lane, which every wave of the kernel is
// given whether it bounces or not -- are compiled out
template <int TRAV, class Occ, bool BOUNCE = true>
__device__ f3 color_main_ray(const GeomParams& P, const Occ occ, PixCtx& c, const RayHit& hit, const bool active = true)
{
    const DevScene& s = P.sc;
    const vrt_settings& st = P.st;
    f3 reflection = mk3(0.0f, 0.0f, 0.0f);
    if (BOUNCE && active && s.palette[hit.material].metallic > 0.0f && st.max_bounces > 0) {
        RayHit bounces[VRT_MAX_BOUNCES];
        RayHit last = hit;
        int last_idx = -1;
        int nb = st.max_bounces > VRT_MAX_BOUNCES ? VRT_MAX_BOUNCES : (int)st.max_bounces;
        for (int i = 0; i < nb; i++) {
            float k = 2.0f * dot3(last.normal, last.dir);
            f3 rdir = mk3(last.dir.x - k * last.normal.x, last.dir.y - k * last.normal.y, last.dir.z - k * last.normal.z);
            f3 o = mk3(last.pos.x + last.normal.x * 0.01f, last.pos.y + last.normal.y * 0.01f, last.pos.z + last.normal.z * 0.01f);
            RayHit rh; RayInt ri;
            trace_ray<TRAV, Occ, false, true>(s, occ, o, rdir, st.max_steps, rh, ri);
            if (VRT_COUNTS(TRAV)) c.fetches += ri.fetches;
            c.rays++;
            bounces[i] = rh;
            last = rh;
            if (last.material == 0 || s.palette[last.material].metallic <= 0.0f) { last_idx = i; break; }
        }
        for (int i = last_idx; i >= 0; i--) {
            f3 col = color_hit<TRAV>(P, occ, c, bounces[i], reflection, (uint32_t)i);
            reflection = mk3(reflection.x + col.x, reflection.y + col.y, reflection.z + col.z);
        }
    }
    return color_hit<TRAV>(P, occ, c, hit, reflection, 0, active);        // (wave-uniform again: the lanes without a hit help with the AO rays)
}

// colorMainRay with the bounce chain as ONE WORD per hit instead of a stack of RayHits (44 B each: 352 B of scratch per lane for
// every wave of the launch, and 0.6 GB of scratch writes per 4K frame on the Mandelbulb).  What the way back needs of a hit on the
// chain is what color() consumes: its material, which of the 27 normals it has (26 face / edge / corner vectors or the zero
// vector of rule A: every normal traceRay can produce, hit_normal), how many of its AO rays hit and whether its shadow ray did --
// 8 + 6 + 16 + 1 bits.  So the secondary rays of every hit are traced on the way OUT, where the hit is at hand, at one call
// site for the primary hit and every bounce; the way back is arithmetic on the words, in the order frag:300-303 prescribes.
// The secondary rays of a METALLIC bounce are traced before it is known whether the chain will end (frag:281-298: a chain of
// max_bounces metallic hits shades none of them, lastIdx = -1): in that one case they were traced for nothing, and their rays
// and steps are taken out of the count planes again, which then hold the reference's numbers as before.
// Entry k of the chain: k = 0 the primary hit, k = i + 1 bounce i (shaded with depth i; the primary with depth 0).
__device__ __forceinline__ uint32_t chain_pack(uint32_t material, const f3 n, uint32_t ao_hits, bool shadowed)
{
    // the normal's code from the vector itself: bit a = component a is not 0, bit 3 + a = it is positive (hit_normal's ncode: the
    // component is -rayStep); a masked axis the ray does not move along has a zero component and drops out of the mask, which is
    // the same vector hit_normal's general form returns
    const uint32_t nc = (uint32_t)(n.x != 0.0f) | ((uint32_t)(n.y != 0.0f) << 1) | ((uint32_t)(n.z != 0.0f) << 2) |
                        ((uint32_t)(n.x > 0.0f) << 3) | ((uint32_t)(n.y > 0.0f) << 4) | ((uint32_t)(n.z > 0.0f) << 5);
    return material | (nc << 8) | ((uint32_t)shadowed << 14) | (ao_hits << 16);
}
__device__ __forceinline__ f3 chain_shade(const GeomParams& P, uint32_t code, f3 reflection, uint32_t depth)
{
    const DevScene& s = P.sc;
    const uint32_t nc = (code >> 8) & 63u, hits = code >> 16;
    const f3 normal = hit_normal(nc & 7u, (nc & 8u) ? -1 : 1, (nc & 16u) ? -1 : 1, (nc & 32u) ? -1 : 1);
    f3 sky;
    if ((nc & 7u) != 0u) { const float4 t = reinterpret_cast<const float4*>(s.sky_normals)[nc]; sky = mk3(t.x, t.y, t.z); }
    else sky = sky_color(s, normal);                         // the zero normal of rule A
    // calcAmbient's sum: `hits` additions of 1 / ao_samples (frag:219-222), not a product
    float ambient = 1.0f;
    if (P.st.ao_samples != 0u) {
        const float sample_frac = 1.0f / (float)P.st.ao_samples;
        ambient = 0.0f;
        for (uint32_t q = 0; q < hits; q++) ambient += sample_frac;
    }
    return shade_eval(P, code & 0xFFu, normal, sky, ambient, ((code >> 14) & 1u) != 0u, reflection, depth);
}

// NBT: the most bounces the launch can ask for (the chain's words are registers: 2, 5 or VRT_MAX_BOUNCES + 1 of them)
template <int TRAV, class Occ, int NBT>
__device__ f3 color_main_ray_packed(const GeomParams& P, const Occ occ, PixCtx& c, const RayHit& hit, const bool is_hit = true)
{
    const DevScene& s = P.sc;
    const vrt_settings& st = P.st;
    const int nb = st.max_bounces > NBT ? NBT : (int)st.max_bounces;
    uint32_t codes[NBT + 1];
#pragma unroll
    for (int q = 0; q <= NBT; q++) codes[q] = 0u;
    f3 reflection = mk3(0.0f, 0.0f, 0.0f);
    RayHit cur = hit;
    int last = 0;                                              // the chain's last entry that is shaded: 0 = the primary hit alone
    uint32_t spec_fetches = 0u, spec_rays = 0u;
    // The loop over the chain's entries is WAVE-UNIFORM: a lane whose chain has ended (or that never had a hit) stays in it for as long
    // as some lane's chain goes on, and takes AO rays from the pool like the others (secondary_rays) -- the few metallic pixels of a
    // bounce get the whole wave's help.  Everything else a lane does here is under `on`: its own chain is still being followed.
    bool on = is_hit;
    for (int k = 0; __ballot(on) != 0ull; k++) {
        // the secondary rays of entry k (a hit: the primary, or a bounce that found something)
        float ambient; uint32_t ao_hits; bool shadowed;
        const uint32_t f0 = c.fetches, r0 = c.rays;
        secondary_rays<TRAV, Occ, true>(P, occ, c, cur.pos, cur.normal, k > 0 ? (uint32_t)(k - 1) : 0u, ambient, ao_hits, shadowed, on);
        if (on) {
            codes[k] = chain_pack(cur.material, cur.normal, ao_hits, shadowed);
            last = k;
            const bool metal = s.palette[cur.material].metallic > 0.0f;
            if (!metal) on = false;                            // (k = 0: no chain at all; k > 0: the chain ends on a hit that does not reflect)
            else {
                if (k > 0) { if (VRT_COUNTS(TRAV)) spec_fetches += c.fetches - f0; spec_rays += c.rays - r0; }    // a metallic bounce: shaded only if the chain ends
                if (k >= nb) { last = -1; on = false; }        // max_bounces metallic bounces (or max_bounces == 0): nothing on the chain is shaded
            }
        }
        if (on) {
            float d2 = 2.0f * dot3(cur.normal, cur.dir);
            f3 rdir = mk3(cur.dir.x - d2 * cur.normal.x, cur.dir.y - d2 * cur.normal.y, cur.dir.z - d2 * cur.normal.z);
            f3 o = mk3(cur.pos.x + cur.normal.x * 0.01f, cur.pos.y + cur.normal.y * 0.01f, cur.pos.z + cur.normal.z * 0.01f);
            RayHit rh; RayInt ri;
            trace_ray<TRAV, Occ, false, true>(s, occ, o, rdir, st.max_steps, rh, ri);
            if (VRT_COUNTS(TRAV)) c.fetches += ri.fetches;
            c.rays++;
            if (rh.material == 0u) {                           // the chain ends in the sky: colorHit of a miss is skyColor(dir)
                const f3 col = sky_color(s, rh.dir);
                reflection = mk3(reflection.x + col.x, reflection.y + col.y, reflection.z + col.z);
                on = false;
            } else cur = rh;
        }
    }
    if (!is_hit) return mk3(0.0f, 0.0f, 0.0f);
    if (last < 0) {
        // frag:281-303 with lastIdx = -1: the bounces' secondary rays were traced for nothing -- the reference never traces them
        if (VRT_COUNTS(TRAV)) c.fetches -= spec_fetches;
        c.rays -= spec_rays;
        last = 0;
    }
    // the way back: entry j (bounce j - 1) with the reflection gathered behind it, frag:300-303
#pragma unroll
    for (int j = NBT; j >= 1; j--) {
        if (j <= last) {
            const f3 col = chain_shade(P, codes[j], reflection, (uint32_t)(j - 1));
            reflection = mk3(reflection.x + col.x, reflection.y + col.y, reflection.z + col.z);
        }
    }
    return chain_shade(P, codes[0], reflection, 0u);
}

// ---------------------------------------------------------------------------------------------
// tile mapping
// ---------------------------------------------------------------------------------------------

// Workgroup -> screen tile.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an
// XCD and its private 4 MiB L2), so XCD slot (b % 8) gets one contiguous run of `chunk` tiles in
// row-major tile order: neighbouring tiles traverse neighbouring volume cells and share L2 lines.
// n / d for wave-uniform operands with rcp = floor(2^32 / d): mulhi is the quotient or one below it, one correction
// step makes it exact for every n < 2^32.  Stays on the scalar unit (a generic 32-bit division is ~20 VALU ops).
__device__ __forceinline__ uint32_t udiv_uniform(uint32_t n, uint32_t d, uint32_t rcp, uint32_t& rem)
{
    uint32_t q = (uint32_t)(((uint64_t)n * (uint64_t)rcp) >> 32);
    uint32_t r = n - q * d;
    if (r >= d) { q++; r -= d; }
    rem = r;
    return q;
}

// The slot (camera, planes, strip assignment) of frame `frame` of the launch.  TABLE = false: a reference into the kernel
// arguments.  TABLE = true (launches of more than VRT_MAX_BATCH frames): a copy read from the table in device memory
// through the constant address space -- the table is not written while the kernel runs, and only loads the compiler
// knows to be invariant become scalar loads (a plain global pointer gives vector loads and the slot in VGPRs).
typedef const __attribute__((address_space(4))) uint32_t* const_u32_ptr;
// n dwords starting at byte offset `off` of slot `frame` of the table, read through the constant address space
template <int N> __device__ __forceinline__ void table_read(const GeomParams& P, uint32_t frame, size_t off, void* dst)
{
    const_u32_ptr w = (const_u32_ptr)((const char*)(P.table + frame) + off);
    uint32_t tmp[N];
#pragma unroll
    for (int i = 0; i < N; i++) tmp[i] = w[i];
    __builtin_memcpy(dst, tmp, sizeof tmp);
}
// The kernel's own arguments (GeomParams is the one argument, at offset 0 of the segment) as words to be read NOW: the
// compiler hoists ordinary argument loads to the top of the kernel, where each costs scalar registers across ray generation;
// what only a rare or late branch needs is read through this pointer, which it cannot see through.
__device__ __forceinline__ const_u32_ptr kernarg_words(size_t byte_offset)
{
    const_u32_ptr p = (const_u32_ptr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + byte_offset);
    asm volatile("" : "+s"(p));
    return p;
}
// base + 32-bit byte offset as a pointer into GLOBAL memory (address space 1): global_load / global_store with the base in a
// scalar pair and the offset in one vector register
template <class T> __device__ __forceinline__ __attribute__((address_space(1))) T* gptr(const void* base, uint32_t byte_offset)
{
    return (__attribute__((address_space(1))) T*)((__attribute__((address_space(1))) char*)base + byte_offset);
}
typedef float vrt_f4 __attribute__((ext_vector_type(4)));
typedef float vrt_f2 __attribute__((ext_vector_type(2)));
// the planes a miss pixel is stored to (the fast sky wave reads these eight pointers, not all fourteen)
struct MissPlanes { uint8_t* color8; float* depth; float* motion; uint8_t* mask8; float* position; int8_t* normal8; uint8_t* hit_id; uint8_t* color8_strips; };
template <bool TABLE> struct SlotOf;
template <> struct SlotOf<false> {
    static __device__ __forceinline__ void head(const GeomParams& P, uint32_t frame, RayGenConsts& g, float* cam_right, int& shard_rank, uint32_t& box)
    {
        const FrameSlot& S = P.slot[frame];
        box = (uint32_t)S.box[0] | ((uint32_t)S.box[1] << 8) | ((uint32_t)S.box[2] << 16) | ((uint32_t)S.box[3] << 24);
        g = S.rg;
        cam_right[0] = S.pc.cam_right[0]; cam_right[1] = S.pc.cam_right[1]; cam_right[2] = S.pc.cam_right[2];
        shard_rank = S.shard_rank;
    }
    static __device__ __forceinline__ vrt_frame planes(const GeomParams& P, uint32_t frame) { return P.slot[frame].fr; }
    // the camera position: only waves that trace need it, and they read it when they know they do (three scalar registers
    // less across ray generation for everybody)
    static __device__ __forceinline__ f3 cam_pos(const GeomParams& P, uint32_t frame)
    {
        const_u32_ptr w = kernarg_words(offsetof(GeomParams, slot) + (size_t)frame * sizeof(FrameSlot) + offsetof(FrameSlot, pc) + offsetof(vrt_push, cam_pos));
        return mk3(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]));
    }
    // N plane pointers of the frame starting with field `first` of vrt_frame, read NOW (kernarg_words)
    template <int N> static __device__ __forceinline__ void ptrs(const GeomParams& P, uint32_t frame, int first, void** out)
    {
        const_u32_ptr fp = kernarg_words(offsetof(GeomParams, slot) + (size_t)frame * sizeof(FrameSlot) + offsetof(FrameSlot, fr) + 8u * (size_t)first);
        uint32_t tmp[2 * N];
#pragma unroll
        for (int q = 0; q < 2 * N; q++) tmp[q] = fp[q];
        __builtin_memcpy(out, tmp, sizeof tmp);
    }
    static __device__ __forceinline__ MissPlanes miss_planes(const GeomParams& P, uint32_t frame)
    {
        // (read late, like the fast path's other constants: through a pointer into the arguments the compiler cannot hoist from)
        const_u32_ptr fp = kernarg_words(offsetof(GeomParams, slot) + (size_t)frame * sizeof(FrameSlot) + offsetof(FrameSlot, fr));
        MissPlanes m;
        uint32_t tmp[16];
#pragma unroll
        for (int q = 0; q < 12; q++) tmp[q] = fp[q];
        tmp[12] = fp[14]; tmp[13] = fp[15]; tmp[14] = fp[26]; tmp[15] = fp[27];
        __builtin_memcpy(&m, tmp, sizeof m);
        return m;
    }
    static __device__ __forceinline__ const vrt_push* push(const GeomParams& P, uint32_t frame) { return &P.slot[frame].pc; }
};
// The table form reads the pieces when they are needed, like the kernel-argument form does: a copy of the whole slot at the
// top keeps the fourteen plane pointers in scalar registers through the traversal (82 + 6 SGPRs: one wave per SIMD less).
template <> struct SlotOf<true> {
    static __device__ __forceinline__ void head(const GeomParams& P, uint32_t frame, RayGenConsts& g, float* cam_right, int& shard_rank, uint32_t& box)
    {
        table_read<1>(P, frame, offsetof(FrameSlot, box), &box);
        table_read<sizeof(RayGenConsts) / 4>(P, frame, offsetof(FrameSlot, rg), &g);
        table_read<3>(P, frame, offsetof(FrameSlot, pc) + offsetof(vrt_push, cam_right), cam_right);
        table_read<1>(P, frame, offsetof(FrameSlot, shard_rank), &shard_rank);
    }
    static __device__ __forceinline__ vrt_frame planes(const GeomParams& P, uint32_t frame)
    {
        vrt_frame f;
        table_read<sizeof(vrt_frame) / 4>(P, frame, offsetof(FrameSlot, fr), &f);
        return f;
    }
    static __device__ __forceinline__ f3 cam_pos(const GeomParams& P, uint32_t frame)
    {
        const_u32_ptr w = (const_u32_ptr)((const char*)(P.table + frame) + offsetof(FrameSlot, pc) + offsetof(vrt_push, cam_pos));
        asm volatile("" : "+s"(w));
        return mk3(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]));
    }
    template <int N> static __device__ __forceinline__ void ptrs(const GeomParams& P, uint32_t frame, int first, void** out)
    {
        const_u32_ptr w = (const_u32_ptr)((const char*)(P.table + frame) + offsetof(FrameSlot, fr) + 8u * (size_t)first);
        asm volatile("" : "+s"(w));                            // (read NOW: not hoisted to where the slot's head is read)
        uint32_t tmp[2 * N];
#pragma unroll
        for (int q = 0; q < 2 * N; q++) tmp[q] = w[q];
        __builtin_memcpy(out, tmp, sizeof tmp);
    }
    static __device__ __forceinline__ MissPlanes miss_planes(const GeomParams& P, uint32_t frame)
    {
        // color8 .. normal8 are the first six pointers of vrt_frame, hit_id the eighth, color8_strips the fourteenth
        static_assert(offsetof(vrt_frame, normal8) == 40 && offsetof(vrt_frame, hit_id) == 56 && offsetof(vrt_frame, color8_strips) == 104, "vrt_frame layout");
        MissPlanes m;
        table_read<12>(P, frame, offsetof(FrameSlot, fr), &m);
        table_read<2>(P, frame, offsetof(FrameSlot, fr) + offsetof(vrt_frame, hit_id), &m.hit_id);
        table_read<2>(P, frame, offsetof(FrameSlot, fr) + offsetof(vrt_frame, color8_strips), &m.color8_strips);
        return m;
    }
    static __device__ __forceinline__ const vrt_push* push(const GeomParams& P, uint32_t frame) { return &P.table[frame].pc; }
};

// workgroup -> frame of the launch and tile row within the frame's local rows (ty) and tile column (tx).  Frames of a
// batch follow one another in the grid: the next frame's first tiles start while this one drains.
// XCD x (= workgroup id & 7) owns every 8th tile row: every XCD gets an even sample of sky and geometry (a contiguous
// band per XCD leaves the XCDs that drew the sky idle), while the tiles of one row -- which walk neighbouring volume cells
// -- still share that XCD's L2.
//   xcd_turn == 0: per frame, row ty belongs to XCD ty % 8; ceil(rows / 8) * 8 row slots per frame (the surplus
//                  workgroups exit at once).
//   xcd_turn == 1: the rows of ALL frames of the launch are dealt round-robin in one sequence (row L = frame * rows + ty
//                  to XCD L % 8).  For row counts far from a multiple of 8 -- a rank's 18 rows of a sharded 1080p frame
//                  would be 3 rows for two XCDs and 2 for the others, and a third of the grid would be surplus -- the XCDs
//                  stay even and only the last seven row slots of the launch can be empty.
// MAP: the launch's xcd_turn as a compile-time constant (the product traversals), or -1: looked at here
template <int MAP>
__device__ __forceinline__ bool block_to_tile(const TileMap& M, uint32_t& frame, int& ty, int& tx)
{
    // xcd_turn 0 and 2 are launched as THREE-dimensional grids (8 x columns, rows, frames): the workgroup's three indices are in
    // scalar registers when the wave starts, and the XCD (workgroups are dealt to the eight XCDs in dispatch order, x fastest)
    // is the low three bits of the x index because the grid's x extent is a multiple of 8 -- no division, where the linear
    // form spends two or three (ten scalar instructions each, in a kernel whose scalar unit -- ONE per CU, shared by its four
    // SIMDs -- is as busy as its vector units: a 1080p frame of sky-only waves that return once they know their block takes
    // 11 us, 158 scalar instructions per wave).
    if (MAP == 2 || (MAP < 0 && M.xcd_turn == 2)) {
        // per frame every XCD owns ONE of 8 screen regions (2 columns x 4 rows of tiles), and the assignment rotates from frame
        // to frame (XCD x traces region (x + frame) % 8): an XCD's rays of one frame then walk one eighth of the volume in one
        // or two direction octants -- a working set of clearance bytes that fits its 4 MiB L2 instead of the whole 17 MB field
        // -- while over 8 frames every XCD traces every region once, so sky and geometry regions balance.
        const uint32_t rw = ((uint32_t)M.tiles_x + 1u) >> 1, rh = ((uint32_t)M.tiles_y_local + 3u) >> 2;
        frame = blockIdx.z;
        const uint32_t region = ((blockIdx.x & 7u) + frame) & 7u;
        tx = (int)((blockIdx.x >> 3) + (region & 1u) * rw);
        ty = (int)(blockIdx.y + (region >> 1) * rh);
        return tx < M.tiles_x && ty < M.tiles_y_local;
    }
    if (MAP == 0 || (MAP < 0 && M.xcd_turn == 0)) {
        // per frame, row ty belongs to XCD ty % 8: every XCD gets an even sample of sky and geometry (a contiguous band per XCD
        // leaves the XCDs that drew the sky idle), while the tiles of one row -- which walk neighbouring volume cells -- still
        // share that XCD's L2; ceil(rows / 8) * 8 row slots per frame (the surplus workgroups exit at once)
        frame = blockIdx.z;
        tx = (int)(blockIdx.x >> 3);
        ty = (int)(blockIdx.y * 8u + (blockIdx.x & 7u));
        return ty < M.tiles_y_local;
    }
    // xcd_turn == 1 (a one-dimensional grid): the rows of ALL frames of the launch are dealt round-robin in one sequence (row
    // L = frame * rows + ty to XCD L % 8).  For row counts far from a multiple of 8 -- a rank's 18 rows of a sharded 1080p
    // frame would be 3 rows for two XCDs and 2 for the others, and a third of the grid would be surplus -- the XCDs stay even
    // and only the last seven row slots of the launch can be empty.
    uint32_t b = blockIdx.x, utx, uty;
    uint32_t L = udiv_uniform(b >> 3, (uint32_t)M.tiles_x, M.tiles_x_rcp, utx) * 8u + (b & 7u);
    frame = udiv_uniform(L, (uint32_t)M.tiles_y_local, M.tiles_y_rcp, uty);
    ty = (int)uty; tx = (int)utx;
    return frame < (uint32_t)M.n_frames;
}

// yp0: the row of y0 in the rank's packed strips (vrt_pack_rows order)
__device__ __forceinline__ bool tile_origin(const TileMap& M, int ty, int tx, int shard_rank, int& x0, int& y0, int& yp0)
{
    // bottom rows first: the rows dispatched last only have the drain of the machine to hide in, and the top of a
    // frame is where the cheap sky-only tiles usually are
    ty = M.tiles_y_local - 1 - ty;
    uint32_t within = (uint32_t)ty;
    int strip_local = 0;                                       // (unsharded: the frame is one strip)
    if (M.nranks > 1) strip_local = (int)udiv_uniform((uint32_t)ty, M.tps, M.tps_rcp, within);
    x0 = tx * M.tile;
    yp0 = strip_local * M.strip_rows + (int)within * M.tile;
    y0 = (strip_local * M.nranks + shard_rank) * M.strip_rows + (int)within * M.tile;
    return y0 < M.H;
}

// Stage the 16^3 and 64^3 occupancy summaries into LDS (16 B per lane per iteration, coalesced).
template <bool LDS> __device__ __forceinline__ OccT<LDS> stage_occ(const GeomParams& P, uint64_t* lds);
template <> __device__ __forceinline__ OccT<false> stage_occ<false>(const GeomParams& P, uint64_t*)
{
    OccT<false> o; o.o2 = P.sc.vol.occ2; o.o3 = P.sc.vol.occ3; return o;
}
template <> __device__ __forceinline__ OccT<true> stage_occ<true>(const GeomParams& P, uint64_t* lds)
{
    const uint4* src2 = reinterpret_cast<const uint4*>(P.sc.vol.occ2);
    const uint4* src3 = reinterpret_cast<const uint4*>(P.sc.vol.occ3);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    uint32_t n2 = P.occ2_bytes / 16, n3 = P.occ3_bytes / 16;
    for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) dst[i] = src2[i];
    for (uint32_t i = threadIdx.x; i < n3; i += blockDim.x) dst[n2 + i] = src3[i];
    __syncthreads();
    OccT<true> o;
    o.o2 = (lds_u64_ptr)lds; o.o3 = (lds_u64_ptr)(lds + P.occ2_bytes / 8);
    return o;
}

// ---------------------------------------------------------------------------------------------
// K1: primary rays
// ---------------------------------------------------------------------------------------------

// colorHit() of every (material, normal) a primary ray can hit, for launches without secondary rays (GeomParams::hit_colors)
__global__ __launch_bounds__(256) void k_hit_colors(const GeomParams P, uint32_t* table)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t code = t & 63u, material = t >> 6, mask = code & 7u;
    const int sx = (code & 8u) ? -1 : 1, sy = (code & 16u) ? -1 : 1, sz = (code & 32u) ? -1 : 1;
    uint32_t c8 = 0u;
    if (mask != 0u && material != 0u && material < 256u) {
        RayHit h;
        h.material = material; h.pos = mk3(0.0f, 0.0f, 0.0f); h.dir = mk3(0.0f, 0.0f, 0.0f);
        h.normal = hit_normal(mask, sx, sy, sz); h.ncode = code;
        PixCtx c; c.px = 0; c.py = 0; c.fetches = 0; c.rays = 0; c.pc = nullptr; c.ldsw = 0u;
        OccT<false> occ; occ.o2 = nullptr; occ.o3 = nullptr;
        const f3 col = color_hit<VRT_TRAVERSAL_DF_FAST, OccT<false>, false>(P, occ, c, h, mk3(0.0f, 0.0f, 0.0f), 0);
        c8 = (uint32_t)unorm8(col.x) | ((uint32_t)unorm8(col.y) << 8) | ((uint32_t)unorm8(col.z) << 16);
    }
    if (t < 256u * 64u) table[t] = c8;
}

hipError_t launch_hit_colors(const GeomParams& p, uint32_t* table, hipStream_t s)
{
    hipLaunchKernelGGL(k_hit_colors, dim3(64), dim3(256), 0, s, p, table);
    return hipGetLastError();
}

// the pixel's colour into color_f (debug), color8 and the packed strips
__device__ __forceinline__ void store_color(const vrt_frame& f, f3 col, size_t i, uint32_t i32, uint32_t strip_off)
{
    if (f.color_f) { f.color_f[i * 3 + 0] = col.x; f.color_f[i * 3 + 1] = col.y; f.color_f[i * 3 + 2] = col.z; }
    if (f.color8 || f.color8_strips) {
        const uint32_t c8 = (uint32_t)unorm8(col.x) | ((uint32_t)unorm8(col.y) << 8) | ((uint32_t)unorm8(col.z) << 16);
        if (f.color8) *gptr<uint32_t>(f.color8, i32 << 2) = c8;
        if (f.color8_strips) *gptr<uint32_t>(f.color8_strips, strip_off) = c8;
    }
}

// MODE 0: write hit records for K2 (split);  1: no secondary rays enabled, shade inline;  4: the megakernel without its bounce loop;
// MODE 5, 6, 7: the megakernel with the bounce chain as one word per hit (color_main_ray_packed: no stack of hits; what the product
//         traversals launch) for at most 2 / 5 / VRT_MAX_BOUNCES bounces;
// MODE 2: megakernel -- the lanes that hit go on to trace their AO / shadow / bounce rays in this same kernel, so that
//         the secondary rays' latency hides under the primary work of the other waves (a separate K2 launch has a
//         single round of waves and is bound by the longest ray's dependency chain).
#ifndef VRT_CHAIN_WAVES
#define VRT_CHAIN_WAVES 7     // (development: the packed-chain megakernels at 6 or 5 waves per SIMD, i.e. 80 / 96 VGPRs)
#endif
#ifndef VRT_MODE4_WAVES
#define VRT_MODE4_WAVES 7     // (development: 8 forces the megakernel without its bounce loop into 64 VGPRs, at the price of 12 B of scratch per lane)
#endif
template <int TRAV, bool OCC_LDS, int MODE, bool TABLE, int MAP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(((MODE == 4 && TRAV == VRT_TRAVERSAL_DF_FAST) ? VRT_MODE4_WAVES : (MODE >= 5 ? VRT_CHAIN_WAVES : 7)), 8))) void k_primary(const GeomParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_occ[];
    // the tile map arrives with one 64-byte scalar load (and one wait) before anything depends on it
    TileMap M = P.map;
#ifdef VRT_EXP_STAMPS
    const uint64_t t_begin = wall_clock64();                                 // (development build, tools/exp_timeline2.py: every wave's start / end stamp goes to the motion plane)
#else
    const uint64_t t_begin = (M.flags & 2u) ? wall_clock64() : 0ull;         // diagnostic timeline (100 MHz)
#endif
    int x0, y0, yp0, ty, tx;
    uint32_t frame;
    if (!block_to_tile<MAP>(M, frame, ty, tx)) return;                   // uniform per workgroup
    // the block's tile tag (k_tile_tags) and the frame's "the tags say nothing" word: two scalar loads that leave together with
    // the frame slot's (their address needs nothing from the slot: tags are laid out in the launch's LOCAL rows of 8x8 blocks,
    // in dispatch order), looked at after ray generation.  Written by the kernel before this one: constant address space.
    // (A launch without tags points at one word per frame that holds its tile_gen, with tags_x = 0 and tags_per_frame = 1.)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t tile_gen = P.tile_gen;
    uint32_t tag, tag_all;
    {
        const_u32_ptr tg = (const_u32_ptr)(P.tile_tags + (size_t)frame * P.tags_per_frame);
        const uint32_t tags_x = P.tags_x, sh = (uint32_t)M.tile >> 4;
        const uint32_t row8 = ((uint32_t)ty << sh) + (uint32_t)(wave >> 1), col8 = ((uint32_t)tx << sh) + (uint32_t)(wave & 1);
        tag = tg[row8 * tags_x + (tags_x ? col8 : 0u)];
        tag_all = tg[P.tags_per_frame - 1u];
    }
    // ... and so does the frame's part of ray generation (camera, hoisted constants, strip assignment), in one batch
    RayGenConsts g;
    float cam_right[3];
    int shard_rank;
    uint32_t box;
    SlotOf<TABLE>::head(P, frame, g, cam_right, shard_rank, box);
    asm volatile("" : "+s"(box));
    // (one scalar from here on, not three: past 80 scalar registers a SIMD holds 7 of these waves, not 8)
    const uint32_t untagged = (uint32_t)__builtin_amdgcn_readfirstlane((tag != tile_gen && tag_all != tile_gen) ? 1 : 0);
    const uint4 boxr = make_uint4(box & 0xFFu, (box >> 8) & 0xFFu, (box >> 16) & 0xFFu, box >> 24);
    float crx = cam_right[0], cry = cam_right[1], crz = cam_right[2];
    float rcp_w = P.rcp_w, rcp_h = P.rcp_h;
    int fast_div = P.fast_screen_div;
    asm volatile("" : "+s"(g.cd.x), "+s"(g.cd.y), "+s"(g.cd.z), "+s"(g.planeV.x), "+s"(g.planeV.y), "+s"(g.planeV.z), "+s"(g.jx), "+s"(g.jy),
                      "+s"(g.W), "+s"(g.H), "+s"(crx), "+s"(cry), "+s"(crz), "+s"(shard_rank),
                      "+s"(rcp_w), "+s"(rcp_h), "+s"(fast_div));
    if (!tile_origin(M, ty, tx, shard_rank, x0, y0, yp0)) return;
    constexpr bool kLds = OCC_LDS && (TRAV == VRT_TRAVERSAL_BITMASK || TRAV == VRT_TRAVERSAL_JUMP);
    const OccT<kLds> occ = stage_occ<kLds>(P, lds_occ);

    // wave w -> 8x8 block (w&1, w>>1) of the tile (one-wave workgroups: w = 0); lane -> (l&7, l>>3).
    // (A 16x4 block would make every store of the 4-byte planes a full 64-byte line, but measured 3 % slower:
    // the wider footprint lowers the wave-wide clearance minimum by more than the stores gain.)
    // (the wave index through readfirstlane: the compiler then knows the block's origin, the rectangle test and the tag test
    // below are scalar -- a branch, not an EXEC mask around the traversal)
    int lane = threadIdx.x & 63;
    const int px0 = x0 + (wave & 1) * 8, py0 = y0 + (wave >> 1) * 8;       // the wave's 8x8 block (wave-uniform)
    int px = px0 + (lane & 7);
    int py = py0 + (lane >> 3);
    int W = M.W, H = M.H;
    if (px >= W || py >= H) return;
    size_t i = (size_t)py * (size_t)W + (size_t)px;

    const DevScene& s = P.sc;
    f3 start = mk3(0.0f, 0.0f, 0.0f);
    const f3 v = primary_v(g, crx, cry, crz, rcp_w, rcp_h, fast_div, px, py);
    RayHit h; RayInt r;
    // a wave whose 8x8 pixels lie outside the frame's box rectangle (FrameSlot::box, vrt_internal.h box_rect) cannot meet the
    // volume: it writes what a miss writes without testing the box (wave-uniform: the rectangle is in units of 32 pixels) --
    // and inside the rectangle neither can a wave whose block no occupied 4^3 cell of the volume projects onto (k_tile_tags)
    const bool skip = box != 0xFF00FF00u && ((uint32_t)(px0 >> 5) < boxr.x || (uint32_t)(px0 >> 5) >= boxr.y || (uint32_t)(py0 >> 5) < boxr.z ||
                                             (uint32_t)(py0 >> 5) >= boxr.w || untagged != 0u);
    // ... and of everything normalize(), atan() and asin() compute for such a pixel only the sky TEXEL is ever seen: vrt_sky.h
    // decides it from the unnormalised direction with a bound on its distance to the spec's own coordinate; a wave in which
    // some lane lies within that bound of a texel edge goes the long way round, every other wave stores the miss pixel here
    // (32-bit byte offsets from the plane pointers: one shift per plane instead of a 64-bit address each)
    f3 dir;
    if (skip) {
    if (M.flags & VRT_MAPFLAG_SKY_FAST) {
        // (everything the short path reads -- the texel constants, the RGBA8 sky, the eight planes a miss is stored to -- is
        // requested HERE in one batch, through pointers the compiler cannot see through: hoisted to the top of the kernel with
        // the other arguments they would be 28 more scalar registers live across ray generation, one wave per SIMD less for
        // every wave of the kernel; read one after the other where each is used they are three dependent round trips in a
        // wave that does little else)
        SkyFastConsts k;
        const uint32_t* sky8;
        {
            const_u32_ptr kp = kernarg_words(offsetof(GeomParams, sc) + offsetof(DevScene, sky8));
            static_assert(offsetof(DevScene, skyk) == offsetof(DevScene, sky8) + 8 && sizeof(SkyFastConsts) == 40, "sky8 and skyk are read as one block");
            uint32_t tmp[12];
#pragma unroll
            for (int q = 0; q < 12; q++) tmp[q] = kp[q];
            __builtin_memcpy(&sky8, tmp, 8);
            __builtin_memcpy(&k, tmp + 2, sizeof k);
        }
        const MissPlanes f = SlotOf<TABLE>::miss_planes(P, frame);
        uint32_t tx, ty;
        const bool sure = sky_texel_fast(v.x, v.y, v.z, k, tx, ty);
        if (__ballot(!sure) == 0ull) {
            // (the pointers were assembled from words, so the compiler no longer knows they are global memory: say so, or every
            // access below is a flat instruction with a 64-bit address in two vector registers)
            const uint32_t c8 = *gptr<const uint32_t>(sky8, (ty * k.w + tx) << 2);
            const uint32_t i32 = (uint32_t)py * (uint32_t)W + (uint32_t)px;
#ifndef VRT_EXP_STAMPS
            if (M.flags & VRT_MAPFLAG_SIX) {                       // the reference's six targets and nothing else: six stores, no pointer tested
                *gptr<vrt_f4>(f.position, i32 << 4) = (vrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
                *gptr<vrt_f2>(f.motion, i32 << 3) = (vrt_f2){0.0f, 0.0f};
                *gptr<float>(f.depth, i32 << 2) = 0.0f;
                *gptr<uint32_t>(f.normal8, i32 << 2) = 0u;
                *gptr<uint8_t>(f.mask8, i32) = (uint8_t)0;
                *gptr<uint32_t>(f.color8, i32 << 2) = c8;
                return;
            }
#endif
            if (f.position) *gptr<vrt_f4>(f.position, i32 << 4) = (vrt_f4){0.0f, 0.0f, 0.0f, 0.0f};
#ifdef VRT_EXP_STAMPS
            if (f.motion) *gptr<vrt_f2>(f.motion, i32 << 3) = (vrt_f2){__uint_as_float((uint32_t)t_begin), __uint_as_float((uint32_t)wall_clock64())};
#else
            if (f.motion) *gptr<vrt_f2>(f.motion, i32 << 3) = (vrt_f2){0.0f, 0.0f};
#endif
            if (f.depth) *gptr<float>(f.depth, i32 << 2) = 0.0f;
            if (f.normal8) *gptr<uint32_t>(f.normal8, i32 << 2) = 0u;
            if (f.mask8) *gptr<uint8_t>(f.mask8, i32) = (uint8_t)0;
            if (f.hit_id) *gptr<uint8_t>(f.hit_id, i32) = (uint8_t)0;
            if (f.color8) *gptr<uint32_t>(f.color8, i32 << 2) = c8;
            if (f.color8_strips) *gptr<uint32_t>(f.color8_strips, ((uint32_t)(yp0 + (py - y0)) * (uint32_t)W + (uint32_t)px) << 2) = c8;
            return;
        }
    }
        dir = primary_normalize(v);
        h.material = 0u; h.dir = dir; h.pos = mk3(0.0f, 0.0f, 0.0f); h.normal = mk3(0.0f, 0.0f, 0.0f); h.ncode = 0xFFFFFFFFu;
        r.material = 0u; r.mask = 0u; r.fetches = 0u; r.mx = r.my = r.mz = 0; r.dbg0 = 1u; r.dbg1 = 0u;
    } else {
        // (the scene scalars the DDA set-up will ask for one by one: requested here, used from these registers later)
        asm volatile("" :: "s"(P.sc.vol.W), "s"(P.sc.vol.H), "s"(P.sc.vol.D), "s"(P.st.max_steps), "s"(P.sc.vol.df), "s"(P.sc.vol.df_stride),
                           "s"(P.sc.vol.vox));
        start = SlotOf<TABLE>::cam_pos(P, frame);
        dir = primary_normalize(v);
        trace_ray<TRAV, OccT<kLds>, MODE == 1>(s, occ, start, dir, P.st.max_steps, h, r);   // look-ahead request: primary-only kernel
    }
    bool hit = h.material != 0;

    const vrt_frame f = SlotOf<TABLE>::planes(P, frame);   // by value: the fourteen plane pointers arrive with two scalar loads, not one by one before each store
    float depth = 0.0f;
    if (hit) depth = len3(mk3(h.pos.x - start.x, h.pos.y - start.y, h.pos.z - start.z));
    // (the reference's targets through 32-bit byte offsets from pointers SAID to be global memory -- the table form assembles
    // them from words and would otherwise get flat instructions with a 64-bit address each; the host admits frames below
    // 2^28 pixels)
    const uint32_t i32 = (uint32_t)i;
#ifndef VRT_EXP_STAMPS
    // (the launch's flags once more from the kernel's own arguments: a scalar register held across the march would be the 77th)
    const bool six = (kernarg_words(offsetof(GeomParams, map) + offsetof(TileMap, flags))[0] & VRT_MAPFLAG_SIX) != 0u;
#else
    const bool six = false;
#endif
    if (six) {                                                  // the reference's six targets and nothing else (the colour below)
        uint32_t n = 0u;
        if (hit) n = (uint32_t)(uint8_t)snorm8(h.normal.x) | ((uint32_t)(uint8_t)snorm8(h.normal.y) << 8) | ((uint32_t)(uint8_t)snorm8(h.normal.z) << 16);
        *gptr<float>(f.depth, i32 << 2) = depth;
        *gptr<vrt_f2>(f.motion, i32 << 3) = (vrt_f2){0.0f, 0.0f};
        *gptr<uint8_t>(f.mask8, i32) = hit ? (uint8_t)230 : (uint8_t)0;
        *gptr<vrt_f4>(f.position, i32 << 4) = (vrt_f4){h.pos.x, h.pos.y, h.pos.z, 0.0f};
        *gptr<uint32_t>(f.normal8, i32 << 2) = n;
    } else {
    if (f.depth) *gptr<float>(f.depth, i32 << 2) = depth;
#ifndef VRT_EXP_STAMPS
    if (f.motion) *gptr<vrt_f2>(f.motion, i32 << 3) = (vrt_f2){0.0f, 0.0f};
#endif
    if (f.mask8) *gptr<uint8_t>(f.mask8, i32) = hit ? (uint8_t)230 : (uint8_t)0;          // unorm8(0.9f) = 230 (tests/test_oracle_kat.py), unorm8(0) = 0
    if (f.position) *gptr<vrt_f4>(f.position, i32 << 4) = (vrt_f4){h.pos.x, h.pos.y, h.pos.z, 0.0f};
    if (f.normal8) {
        uint32_t n = 0u;                                                    // miss: normal = 0; a wave without a hit skips the conversions
        if (hit) n = (uint32_t)(uint8_t)snorm8(h.normal.x) | ((uint32_t)(uint8_t)snorm8(h.normal.y) << 8) | ((uint32_t)(uint8_t)snorm8(h.normal.z) << 16);
        *gptr<uint32_t>(f.normal8, i32 << 2) = n;
    }
    if (f.hit_id) *gptr<uint8_t>(f.hit_id, i32) = (uint8_t)h.material;
    if (f.hit_voxel) {
        f.hit_voxel[i * 3 + 0] = hit ? (int16_t)r.mx : (int16_t)0;
        f.hit_voxel[i * 3 + 1] = hit ? (int16_t)r.my : (int16_t)0;
        f.hit_voxel[i * 3 + 2] = hit ? (int16_t)r.mz : (int16_t)0;
    }
    if (f.hit_mask) f.hit_mask[i] = hit ? (uint8_t)r.mask : (uint8_t)0;
    if (VRT_COUNTS(TRAV) && f.steps_primary) f.steps_primary[i] = r.fetches;
    if (VRT_COUNTS(TRAV) && f.steps_total) f.steps_total[i] = (P.st.flags & VRT_FLAG_DEBUG_PLANES) ? r.dbg0 : r.fetches;
    if (f.rays_total) f.rays_total[i] = (P.st.flags & VRT_FLAG_DEBUG_PLANES) ? r.dbg1 : 1u;
    if ((P.st.flags & 2u) && f.steps_total && f.rays_total) {             // wave start / end stamps, 10 ns units
        f.steps_total[i] = (uint32_t)t_begin;
        f.rays_total[i] = (uint32_t)wall_clock64();
    }
    }
    uint32_t* const steps_total = six ? nullptr : f.steps_total; uint32_t* const rays_total = six ? nullptr : f.rays_total;

    // primary rays only: a hit's colour is an entry of the launch's table (GeomParams::hit_colors: colorHit() of every material
    // and normal, made by colorHit() itself); a wave one of whose hits has none of the 26 normals computes as before
    if (MODE == 1) {
        const uint32_t* const hc = P.hit_colors;
        if (hc && !f.color_f && __ballot(hit && h.ncode == 0xFFFFFFFFu) == 0ull) {
            uint32_t c8;
            if (hit) c8 = *gptr<const uint32_t>(hc, ((h.material << 6) | h.ncode) << 2);
            else {
                const f3 col = sky_color(s, dir);
                c8 = (uint32_t)unorm8(col.x) | ((uint32_t)unorm8(col.y) << 8) | ((uint32_t)unorm8(col.z) << 16);
            }
            if (six) { *gptr<uint32_t>(f.color8, i32 << 2) = c8; return; }
            if (f.color8) *gptr<uint32_t>(f.color8, i32 << 2) = c8;
            if (f.color8_strips) *gptr<uint32_t>(f.color8_strips, ((uint32_t)(yp0 + (py - y0)) * (uint32_t)W + (uint32_t)px) << 2) = c8;
            return;
        }
    }
    if (MODE != 0) {                                           // 1: primary only; 2: megakernel; 4: megakernel, nothing can bounce
        f3 col;
        if (MODE >= 5) {
            // The packed chain is wave-uniform: in a wave with a hit EVERY lane goes through it, the lanes that missed (and those whose chain
            // has ended) for the AO rays' sake -- they draw on the wave's pool like the others (secondary_rays); what they return is not
            // used.  (The other forms keep the divergent call: with helpers at the primary hit alone the megakernel without its bounce
            // loop needs 67 VGPRs instead of 64 and the reference defaults measured 132.8 against 128 us, config 5 3.32 against 3.07 ms.)
            PixCtx c; c.px = px; c.py = py; c.fetches = 0; c.rays = 0; c.pc = SlotOf<TABLE>::push(P, frame);
            c.ldsw = (uint32_t)(uintptr_t)(lds_u64_ptr)lds_occ + (uint32_t)wave * (uint32_t)VRT_AO_SLOT;   // (the hand-written loop's kernels are launched with a pool per wave)
            f3 colh = mk3(0.0f, 0.0f, 0.0f);
            if (__ballot(hit) != 0ull) colh = color_main_ray_packed<TRAV, OccT<kLds>, (MODE == 5 ? 2 : (MODE == 6 ? 5 : VRT_MAX_BOUNCES))>(P, occ, c, h, hit);
            if (hit) {
                col = colh;
                if (VRT_COUNTS(TRAV) && steps_total && !(P.st.flags & 3u)) steps_total[i] = r.fetches + c.fetches;
                if (rays_total && !(P.st.flags & 3u)) rays_total[i] = 1u + c.rays;
            } else {
                col = sky_color(s, dir);
                // (VRT_FLAG_LOOKUP_COUNTS: the bytes a lane asked for while it helped with the others' AO rays belong to the frame's sum)
                if (VRT_COUNTS(TRAV) && steps_total && P.sc.vol.count_lookups != 0u && !(P.st.flags & 3u)) steps_total[i] = r.fetches + c.fetches;
            }
        } else if (hit) {
            PixCtx c; c.px = px; c.py = py; c.fetches = 0; c.rays = 0; c.pc = SlotOf<TABLE>::push(P, frame);
            c.ldsw = (uint32_t)(uintptr_t)(lds_u64_ptr)lds_occ + (uint32_t)wave * (uint32_t)VRT_AO_SLOT;
            if (MODE == 1) col = color_hit<TRAV, OccT<kLds>, false>(P, occ, c, h, mk3(0.0f, 0.0f, 0.0f), 0);   // ambient = 1, unshadowed, no reflection
            else {
                col = color_main_ray<TRAV, OccT<kLds>, MODE != 4>(P, occ, c, h);
                if (VRT_COUNTS(TRAV) && steps_total && !(P.st.flags & 3u)) steps_total[i] = r.fetches + c.fetches;
                if (rays_total && !(P.st.flags & 3u)) rays_total[i] = 1u + c.rays;
            }
        } else {
            col = sky_color(s, dir);
        }
#ifdef VRT_EXP_LATE_INDEX
        {   // the pixel's indices once more, from px and py alone: nothing but those two stays live across the secondary rays
            int pxl = px, pyl = py;
            asm volatile("" : "+v"(pxl), "+v"(pyl));
            const size_t il = (size_t)pyl * (size_t)W + (size_t)pxl;
            store_color(f, col, il, (uint32_t)il, ((uint32_t)(yp0 + (pyl - y0)) * (uint32_t)W + (uint32_t)pxl) << 2);
        }
#else
        store_color(f, col, i, i32, ((uint32_t)(yp0 + (py - y0)) * (uint32_t)W + (uint32_t)px) << 2);
#endif
#ifdef VRT_EXP_STAMPS
        if (f.motion) *gptr<vrt_f2>(f.motion, i32 << 3) = (vrt_f2){__uint_as_float((uint32_t)t_begin), __uint_as_float((uint32_t)wall_clock64())};
#endif
    } else if (hit) {
        // hit record for K2 (position bits + material | mask << 8 | (step+1) codes) and a slot in the compacted list of
        // hit pixels: K2 then runs one lane per HIT pixel instead of one per pixel (hipcc folds the per-lane
        // atomicAdd into one atomic per wave)
        uint32_t packed = h.material | (r.mask << 8) | ((uint32_t)(r.sx + 1) << 11) | ((uint32_t)(r.sy + 1) << 13) |
                          ((uint32_t)(r.sz + 1) << 15);
        P.records[i] = make_uint4(__float_as_uint(h.pos.x), __float_as_uint(h.pos.y), __float_as_uint(h.pos.z), packed);
        uint32_t slot = atomicAdd(P.hit_count, 1u);
        P.hit_list[slot] = (uint32_t)i;
    } else {
        // misses are final here: colorMainRay is never reached (voxel_volume.frag:337-345)
        store_color(f, sky_color(s, dir), i, i32, ((uint32_t)(yp0 + (py - y0)) * (uint32_t)W + (uint32_t)px) << 2);
    }
}

// ---------------------------------------------------------------------------------------------
// K2: secondary rays + shading
// ---------------------------------------------------------------------------------------------

template <int TRAV, bool OCC_LDS>
__global__ __launch_bounds__(256) void k_shade(const GeomParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_occ[];
    constexpr bool kLds = OCC_LDS && (TRAV == VRT_TRAVERSAL_BITMASK || TRAV == VRT_TRAVERSAL_JUMP);
    uint32_t count = *P.hit_count;                        // written by K1 (previous kernel on the stream)
    if (blockIdx.x * 256u >= count) return;               // uniform per workgroup
    const OccT<kLds> occ = stage_occ<kLds>(P, lds_occ);
    uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid >= count) return;
    size_t i = P.hit_list[gid];
    int W = P.W;
    int px = (int)(i % (size_t)W), py = (int)(i / (size_t)W);

    uint4 rec = P.records[i];
    RayHit h;
    h.material = rec.w & 0xFFu;
    h.dir = primary_dir(P.slot[0], px, py);
    h.pos = mk3(__uint_as_float(rec.x), __uint_as_float(rec.y), __uint_as_float(rec.z));
    uint32_t mask = (rec.w >> 8) & 7u;
    int sx = (int)((rec.w >> 11) & 3u) - 1, sy = (int)((rec.w >> 13) & 3u) - 1, sz = (int)((rec.w >> 15) & 3u) - 1;
    PixCtx c; c.px = px; c.py = py; c.fetches = 0; c.rays = 0; c.pc = &P.slot[0].pc;
    c.ldsw = (uint32_t)(uintptr_t)(lds_u64_ptr)lds_occ + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (uint32_t)VRT_AO_SLOT;
    h.normal = hit_normal(mask, sx, sy, sz);
    {
        const bool general = (mask & 7u) == 0u || ((mask & 1u) && sx == 0) || ((mask & 2u) && sy == 0) || ((mask & 4u) && sz == 0);
        h.ncode = general ? 0xFFFFFFFFu : ((mask & 7u) | ((uint32_t)(sx < 0) << 3) | ((uint32_t)(sy < 0) << 4) | ((uint32_t)(sz < 0) << 5));
    }
    f3 col = color_main_ray<TRAV>(P, occ, c, h);
    const vrt_frame& f = P.slot[0].fr;
    if (f.color_f) { f.color_f[i * 3 + 0] = col.x; f.color_f[i * 3 + 1] = col.y; f.color_f[i * 3 + 2] = col.z; }
    if (f.color8 || f.color8_strips) {
        uchar4 c8; c8.x = unorm8(col.x); c8.y = unorm8(col.y); c8.z = unorm8(col.z); c8.w = 0;
        if (f.color8) reinterpret_cast<uchar4*>(f.color8)[i] = c8;
        if (f.color8_strips) {                 // the split form has no tile: row -> packed row by division
            int strip = py / P.sh.strip_rows;
            int yp = (strip / P.sh.nranks) * P.sh.strip_rows + (py - strip * P.sh.strip_rows);
            reinterpret_cast<uchar4*>(f.color8_strips)[(size_t)yp * (size_t)P.W + (size_t)px] = c8;
        }
    }
    if (VRT_COUNTS(TRAV) && f.steps_total) f.steps_total[i] += c.fetches;
    if (f.rays_total) f.rays_total[i] += c.rays;
}

// ---------------------------------------------------------------------------------------------
// launch plumbing for K1 / K2
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// Tile tags: which 8x8-pixel blocks of a frame can a primary ray meet anything in?  One lane per occupied 4^3-voxel cell of
// the volume (the scene's list of them); the cell, grown by one voxel on every side, is projected through the frame's
// camera and the blocks its screen rectangle (+ 2 pixels) touches are tagged with the launch's generation number -- plain
// stores of one value, no clearing between launches.  A block without the tag holds no pixel whose ray passes within a
// voxel of anything solid: its wave writes what a miss writes (k_primary).  Frames without a box rectangle (camera in or
// near the volume, a degenerate basis) have no tags, nor have frames with a cell closer than the projection can bound.
// ---------------------------------------------------------------------------------------------
template <bool TABLE>
__global__ __launch_bounds__(256) void k_tile_tags(const GeomParams P)
{
    const uint32_t frame = blockIdx.y;
    RayGenConsts g;
    float cam[3], U[3];
    int shard_rank;
    uint32_t box;
    SlotOf<TABLE>::head(P, frame, g, U, shard_rank, box);
    if (box == 0xFF00FF00u) return;
    { const f3 cp = SlotOf<TABLE>::cam_pos(P, frame); cam[0] = cp.x; cam[1] = cp.y; cam[2] = cp.z; }
    const uint32_t ci = blockIdx.x * 256u + threadIdx.x;
    if (ci >= P.n_cells) return;
    const uint32_t cell = P.cells[ci];
    const uint32_t cs = P.cell_size;                            // 4 (dense scenes: the cells of the 16^3 summaries' bits) or 8 (bricks)
    const float ext = (float)cs + 2.0f;
    const float lo[3] = {(float)((cell & 1023u) * cs) - 1.0f, (float)(((cell >> 10) & 1023u) * cs) - 1.0f, (float)((cell >> 20) * cs) - 1.0f};
    uint32_t* tags = P.tile_tags + (size_t)frame * P.tags_per_frame;
    // the cell's screen rectangle and the bound on its error (vrt_tags.h: the projection, its rounding analysis and the rule
    // that a rectangle is only used while its bound is below what the two pixels of margin absorb)
    TagCam tc;
    tc.U[0] = U[0]; tc.U[1] = U[1]; tc.U[2] = U[2];
    tc.V[0] = g.planeV.x; tc.V[1] = g.planeV.y; tc.V[2] = g.planeV.z;
    tc.C[0] = g.cd.x + g.jx; tc.C[1] = g.cd.y + g.jy; tc.C[2] = g.cd.z;
    tc.cam[0] = cam[0]; tc.cam[1] = cam[1]; tc.cam[2] = cam[2];
    tc.W = g.W; tc.H = g.H;
    // a rank that owns ONE band of rows (the banded assignment of a multi-GPU batch): seven cells in eight of an 8-GPU step lie
    // outside it, and two linear forms say so without the eight corners' divisions (vrt_tags.h tag_band_cull, with its own bound)
    if (P.sh.nranks > 1) {
        const int rows = P.sh.strip_rows, total = ((int)g.H + rows - 1) / rows;
        if (shard_rank + P.sh.nranks >= total && tag_band_cull(tc, lo, ext, (float)(shard_rank * rows), (float)(shard_rank * rows + rows))) return;
    }
    float x0, x1, y0, y1, ex, ey;
    const int st = tag_project(tc, lo, ext, x0, x1, y0, y1, ex, ey);
    bool all = (st & 1) != 0;
    const float m = VRT_TAG_MARGIN_PX;
    if (!all) {
        // off the screen by more than the margin AND more than its own error bound: nothing to tag, whatever the bound is
        const float gx = fmaxf(m, ex + 0.5f), gy = fmaxf(m, ey + 0.5f);
        if (x1 + gx < 0.0f || y1 + gy < 0.0f || x0 - gx > g.W || y0 - gy > g.H) return;
        if (st & 2) all = true;
    }
    int tx0 = (int)floorf(fmaxf(x0 - m, 0.0f) * 0.125f), tx1 = (int)floorf(fminf(x1 + m, g.W - 1.0f) * 0.125f);
    int ty0 = (int)floorf(fmaxf(y0 - m, 0.0f) * 0.125f), ty1 = (int)floorf(fminf(y1 + m, g.H - 1.0f) * 0.125f);
    if (!all && (tx1 - tx0 + 1) * (ty1 - ty0 + 1) > 1024) all = true;                           // (a cell that covers the screen: tagging it costs more than it saves)
    if (all) { tags[P.tags_per_frame - 1u] = P.tile_gen; return; }
    // screen row of blocks -> the row K1 finds it in: the launch's local rows (the strips this frame's rank owns), in dispatch
    // order (tile_origin: bottom rows first)
    const int nranks = P.sh.nranks, rows8 = P.sh.strip_rows >> 3, k = P.tile_h >> 3, T = P.tiles_y_local;
    if (nranks <= 1) {
        for (int ty = ty0; ty <= ty1; ty++) {
            const int tyl = ty / k;                                // the workgroup tile's local row, and the block's row within the tile
            if (tyl >= T) continue;
            const uint32_t row = (uint32_t)((T - 1 - tyl) * k + (ty - tyl * k));
            // (plain stores; looking first whether the block has its tag already -- most are covered by many cells -- measured slower)
            for (int tx = tx0; tx <= tx1; tx++) tags[row * P.tags_x + (uint32_t)tx] = P.tile_gen;
        }
        return;
    }
    // sharded: only the strips this frame's rank traces -- strip by strip, not row by row with a test (a rank of eight owns one
    // band of the screen: seven eighths of a cell's rows are somebody else's)
    const int s0 = ty0 / rows8, s1 = ty1 / rows8;
    int s = s0 + ((shard_rank - s0 % nranks) + nranks) % nranks;   // the first strip >= s0 that is dealt to shard_rank
    for (; s <= s1; s += nranks) {
        const int a = ty0 > s * rows8 ? ty0 : s * rows8, b = ty1 < (s + 1) * rows8 - 1 ? ty1 : (s + 1) * rows8 - 1;
        for (int ty = a; ty <= b; ty++) {
            const int local = (s / nranks) * rows8 + (ty - s * rows8);
            const int tyl = local / k;
            if (tyl >= T) continue;
            const uint32_t row = (uint32_t)((T - 1 - tyl) * k + (local - tyl * k));
            for (int tx = tx0; tx <= tx1; tx++) tags[row * P.tags_x + (uint32_t)tx] = P.tile_gen;
        }
    }
}

hipError_t launch_tile_tags(const GeomParams& p, hipStream_t s)
{
    if (p.n_cells == 0u) return hipSuccess;
    dim3 grid((p.n_cells + 255u) / 256u, (unsigned)p.n_frames), block(256);
    if (p.table) hipLaunchKernelGGL(k_tile_tags<true>, grid, block, 0, s, p);
    else         hipLaunchKernelGGL(k_tile_tags<false>, grid, block, 0, s, p);
    return hipGetLastError();
}

template <int TRAV, bool OCC_LDS>
static hipError_t launch_primary_t(const GeomParams& p, hipStream_t s)
{
    // (block_to_tile: three-dimensional grids whose x extent is a multiple of 8 for xcd_turn 0 and 2, a line for xcd_turn 1)
    dim3 grid((unsigned)p.tiles_x * 8u, (unsigned)((p.tiles_y_local + 7) / 8), (unsigned)p.n_frames);
    if (p.xcd_turn == 2) grid = dim3(8u * (((unsigned)p.tiles_x + 1u) / 2u), ((unsigned)p.tiles_y_local + 3u) / 4u, (unsigned)p.n_frames);
    else if (p.xcd_turn) grid = dim3((unsigned)p.tiles_x * 8u * (unsigned)((p.tiles_y_local * p.n_frames + 7) / 8));
    dim3 block(p.tile_h == 8 ? 64 : 256);
    size_t lds = (OCC_LDS && (TRAV == VRT_TRAVERSAL_BITMASK || TRAV == VRT_TRAVERSAL_JUMP)) ? p.occ2_bytes + p.occ3_bytes : 0;
    // (the hand-written loop's AO batches: one slot of waiting rays per wave, df_ao_batch_loop)
    if ((TRAV == VRT_TRAVERSAL_DF_FAST || TRAV == VRT_TRAVERSAL_DF_FAST_CNT || TRAV == VRT_TRAVERSAL_BRICK || TRAV == VRT_TRAVERSAL_BRICK_CNT) && p.fused_shade != 1)
        lds = (size_t)(block.x / 64u) * (size_t)VRT_AO_SLOT;
    // (the product traversals with the tile map's form as a compile-time constant: block_to_tile)
    constexpr bool kProduct = TRAV == VRT_TRAVERSAL_DF_FAST || TRAV == VRT_TRAVERSAL_DF_FAST_CNT || TRAV == VRT_TRAVERSAL_BRICK || TRAV == VRT_TRAVERSAL_BRICK_CNT;
    constexpr bool kCount = TRAV == VRT_TRAVERSAL_DF_FAST_CNT || TRAV == VRT_TRAVERSAL_BRICK_CNT;      // (counting launches: the general tile map only -- fewer kernels to build)
    const int map = (kProduct && !kCount && p.xcd_turn != 1) ? p.xcd_turn : -1;
#define VRT_LAUNCH_K1(MODE_, TABLE_)                                                                                           \
    do {                                                                                                                       \
        if (kProduct && !kCount && map == 0)      hipLaunchKernelGGL((k_primary<TRAV, OCC_LDS, MODE_, TABLE_, (kProduct && !kCount) ? 0 : -1>), grid, block, lds, s, p); \
        else if (kProduct && !kCount && map == 2) hipLaunchKernelGGL((k_primary<TRAV, OCC_LDS, MODE_, TABLE_, (kProduct && !kCount) ? 2 : -1>), grid, block, lds, s, p); \
        else                           hipLaunchKernelGGL((k_primary<TRAV, OCC_LDS, MODE_, TABLE_, -1>), grid, block, lds, s, p);    \
    } while (0)
    if (p.table) {               // the split form renders one frame per launch and never gets here
        if (p.fused_shade == 1)      VRT_LAUNCH_K1(1, true);
        else if (p.fused_shade == 2) { if (p.no_bounce) VRT_LAUNCH_K1(4, true); else if (kProduct && p.packed_chain) { if (p.st.max_bounces <= 2) VRT_LAUNCH_K1((kProduct ? 5 : 2), true); else if (p.st.max_bounces <= 5) VRT_LAUNCH_K1((kProduct ? 6 : 2), true); else VRT_LAUNCH_K1((kProduct ? 7 : 2), true); } else VRT_LAUNCH_K1(2, true); }
        else return hipErrorInvalidValue;
    }
    else if (p.fused_shade == 1) VRT_LAUNCH_K1(1, false);
    else if (p.fused_shade == 2) { if (p.no_bounce) VRT_LAUNCH_K1(4, false); else if (kProduct && p.packed_chain) { if (p.st.max_bounces <= 2) VRT_LAUNCH_K1((kProduct ? 5 : 2), false); else if (p.st.max_bounces <= 5) VRT_LAUNCH_K1((kProduct ? 6 : 2), false); else VRT_LAUNCH_K1((kProduct ? 7 : 2), false); } else VRT_LAUNCH_K1(2, false); }
    else                         VRT_LAUNCH_K1(0, false);
#undef VRT_LAUNCH_K1
    return hipGetLastError();
}

template <int TRAV, bool OCC_LDS>
static hipError_t launch_shade_t(const GeomParams& p, hipStream_t s)
{
    // one lane per hit pixel of the compacted list; sized for the worst case (every local pixel hit), surplus
    // workgroups leave at once
    dim3 grid((unsigned)(((size_t)p.total_tiles * p.tile_w * p.tile_h + 255) / 256)), block(256);
    size_t lds = (OCC_LDS && (TRAV == VRT_TRAVERSAL_BITMASK || TRAV == VRT_TRAVERSAL_JUMP)) ? p.occ2_bytes + p.occ3_bytes : 0;
    if (TRAV == VRT_TRAVERSAL_DF_FAST || TRAV == VRT_TRAVERSAL_DF_FAST_CNT || TRAV == VRT_TRAVERSAL_BRICK || TRAV == VRT_TRAVERSAL_BRICK_CNT) lds = 4u * (size_t)VRT_AO_SLOT;
    hipLaunchKernelGGL((k_shade<TRAV, OCC_LDS>), grid, block, lds, s, p);
    return hipGetLastError();
}

static int effective_traversal(int t, int fast_loop)
{
    if (t == VRT_TRAVERSAL_BRICK) return fast_loop == 2 ? VRT_TRAVERSAL_BRICK_CNT : t;
    if (fast_loop == 2 && (t == VRT_TRAVERSAL_AUTO || t == VRT_TRAVERSAL_DF)) return VRT_TRAVERSAL_DF_FAST_CNT;     // the loops' counting twins
    if (t == VRT_TRAVERSAL_DENSE || t == VRT_TRAVERSAL_BITMASK || t == VRT_TRAVERSAL_JUMP || t == VRT_TRAVERSAL_DFJ) return t;
    return fast_loop ? VRT_TRAVERSAL_DF_FAST : VRT_TRAVERSAL_DF;        // AUTO / DF
}

hipError_t launch_primary(const GeomParams& p, hipStream_t s)
{
    int t = effective_traversal((int)p.st.traversal, p.fast_loop);
    if (t == VRT_TRAVERSAL_DF_FAST) return launch_primary_t<VRT_TRAVERSAL_DF_FAST, false>(p, s);
    if (t == VRT_TRAVERSAL_DF_FAST_CNT) return launch_primary_t<VRT_TRAVERSAL_DF_FAST_CNT, false>(p, s);
    if (t == VRT_TRAVERSAL_BRICK) return launch_primary_t<VRT_TRAVERSAL_BRICK, false>(p, s);
    if (t == VRT_TRAVERSAL_BRICK_CNT) return launch_primary_t<VRT_TRAVERSAL_BRICK_CNT, false>(p, s);
    if (t == VRT_TRAVERSAL_DENSE) return launch_primary_t<VRT_TRAVERSAL_DENSE, false>(p, s);
    if (t == VRT_TRAVERSAL_DF) return launch_primary_t<VRT_TRAVERSAL_DF, false>(p, s);
    if (t == VRT_TRAVERSAL_DFJ) return launch_primary_t<VRT_TRAVERSAL_DFJ, false>(p, s);
    if (t == VRT_TRAVERSAL_BITMASK) return p.occ_in_lds ? launch_primary_t<VRT_TRAVERSAL_BITMASK, true>(p, s) : launch_primary_t<VRT_TRAVERSAL_BITMASK, false>(p, s);
    return p.occ_in_lds ? launch_primary_t<VRT_TRAVERSAL_JUMP, true>(p, s) : launch_primary_t<VRT_TRAVERSAL_JUMP, false>(p, s);
}

hipError_t launch_shade(const GeomParams& p, hipStream_t s)
{
    int t = effective_traversal((int)p.st.traversal, p.fast_loop);
    if (t == VRT_TRAVERSAL_DF_FAST) return launch_shade_t<VRT_TRAVERSAL_DF_FAST, false>(p, s);
    if (t == VRT_TRAVERSAL_DF_FAST_CNT) return launch_shade_t<VRT_TRAVERSAL_DF_FAST_CNT, false>(p, s);
    if (t == VRT_TRAVERSAL_BRICK) return launch_shade_t<VRT_TRAVERSAL_BRICK, false>(p, s);
    if (t == VRT_TRAVERSAL_BRICK_CNT) return launch_shade_t<VRT_TRAVERSAL_BRICK_CNT, false>(p, s);
    if (t == VRT_TRAVERSAL_DENSE) return launch_shade_t<VRT_TRAVERSAL_DENSE, false>(p, s);
    if (t == VRT_TRAVERSAL_DF) return launch_shade_t<VRT_TRAVERSAL_DF, false>(p, s);
    if (t == VRT_TRAVERSAL_DFJ) return launch_shade_t<VRT_TRAVERSAL_DFJ, false>(p, s);
    if (t == VRT_TRAVERSAL_BITMASK) return p.occ_in_lds ? launch_shade_t<VRT_TRAVERSAL_BITMASK, true>(p, s) : launch_shade_t<VRT_TRAVERSAL_BITMASK, false>(p, s);
    return p.occ_in_lds ? launch_shade_t<VRT_TRAVERSAL_JUMP, true>(p, s) : launch_shade_t<VRT_TRAVERSAL_JUMP, false>(p, s);
}

const char* primary_kernel_name(int traversal, int fused, int occ_lds)
{
    int t = effective_traversal(traversal, 0);
    (void)fused; (void)occ_lds;
    return t == VRT_TRAVERSAL_DENSE ? "k_primary<dense>" : (t == VRT_TRAVERSAL_BITMASK ? "k_primary<bitmask>" : (t == VRT_TRAVERSAL_DF ? "k_primary<df>" : "k_primary<jump>"));
}

// ---------------------------------------------------------------------------------------------
// K3: a-trous denoiser pass
// ---------------------------------------------------------------------------------------------
//
// Same arithmetic as denoiser.frag:38-73 / the oracle, bit for bit, but with the work the values make
// unnecessary left out:
//   * UNORM8 / SNORM8 decode c/255, c/127: q0 = c*r, q = fma(fma(-D, q0, c), r, q0) with r = RN(1/D) equals the
//     IEEE quotient for every one of the 256 codes (checked exhaustively in tests/test_denoise_decode.py);
//     3 instructions instead of a ~12-instruction division sequence, 8 decodes per tap;
//   * pass 0 has phi = 1/0 * phi0 = +inf, so every edge-stopping weight is min(exp(-0), 1) = 1 exactly
//     (NaN / inf distances included: fminf ignores the NaN): the pass is specialised to a plain weighted blur;
//   * a distance of exactly 0 gives exp(-0) = 1 and a quotient below -87 gives exp = 0 (vrt_spec.h exp_spec):
//     neither needs the division + exponential; a zero weight makes the whole tap contribute +0.
// One wave = 64 consecutive pixels of a row (coalesced 256 B / 1 KiB accesses).

struct Guides { float c[4], n[4], p[4]; };


__device__ __forceinline__ void texel_guides(const DenoiseParams& P, int x, int y, Guides& g)
{
    x = x < 0 ? 0 : (x > P.W - 1 ? P.W - 1 : x);
    y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);
    size_t i = (size_t)y * (size_t)P.W + (size_t)x;
    uchar4 c = reinterpret_cast<const uchar4*>(P.color_in)[i];
    char4 n = reinterpret_cast<const char4*>(P.normal)[i];
    float4 p = reinterpret_cast<const float4*>(P.position)[i];
    g.c[0] = decode_unorm8(c.x); g.c[1] = decode_unorm8(c.y); g.c[2] = decode_unorm8(c.z); g.c[3] = decode_unorm8(c.w);
    g.n[0] = decode_snorm8(n.x); g.n[1] = decode_snorm8(n.y); g.n[2] = decode_snorm8(n.z); g.n[3] = decode_snorm8(n.w);
    g.p[0] = p.x; g.p[1] = p.y; g.p[2] = p.z; g.p[3] = p.w;
}

__device__ __forceinline__ void sample_guides(const DenoiseParams& P, int px, int py, float ox, float oy, Guides& g)
{
    if (ox == floorf(ox) && oy == floorf(oy)) { texel_guides(P, px + (int)ox, py + (int)oy, g); return; }
    float fx = ((float)px + 0.5f + ox) - 0.5f, fy = ((float)py + 0.5f + oy) - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    Guides g00, g10, g01, g11;
    texel_guides(P, x0, y0, g00); texel_guides(P, x0 + 1, y0, g10);
    texel_guides(P, x0, y0 + 1, g01); texel_guides(P, x0 + 1, y0 + 1, g11);
    for (int k = 0; k < 4; k++) {
        float a, b;
        a = g00.c[k] + tx * (g10.c[k] - g00.c[k]); b = g01.c[k] + tx * (g11.c[k] - g01.c[k]); g.c[k] = a + ty * (b - a);
        a = g00.n[k] + tx * (g10.n[k] - g00.n[k]); b = g01.n[k] + tx * (g11.n[k] - g01.n[k]); g.n[k] = a + ty * (b - a);
        a = g00.p[k] + tx * (g10.p[k] - g00.p[k]); b = g01.p[k] + tx * (g11.p[k] - g01.p[k]); g.p[k] = a + ty * (b - a);
    }
}

__device__ __forceinline__ float dist2_4(const float* a, const float* b)
{
    float t0 = a[0] - b[0], t1 = a[1] - b[1], t2 = a[2] - b[2], t3 = a[3] - b[3];
    return ((t0 * t0 + t1 * t1) + t2 * t2) + t3 * t3;
}

// min(exp(-(d2)/phi), 1) (denoiser.frag:55,60,65) for a finite phi > 0, skipping the division and the exponential
// when the value of d2 already decides the result.
__device__ __forceinline__ float edge_weight(float d2, float phi)
{
    if (d2 == 0.0f) return 1.0f;                      // (-0)/phi = -0, exp(-0) = 1
    float x = (-d2) / phi;
    if (x < -87.0f) return 0.0f;                      // exp_spec's own cut-off
    return fminf(exp_spec(x), 1.0f);
}

// Row mapping shared by the denoiser and the strip copy kernels: local row index -> frame row.
__device__ __forceinline__ int strip_row(const ShardMap& sh, int extend, int r, int H)
{
    int per = sh.strip_rows + 2 * extend;
    int k = r / per, j = r % per;
    int g = k * sh.nranks + sh.rank;
    int y = g * sh.strip_rows - extend + j;
    int end = (g + 1) * sh.strip_rows; if (end > H) end = H;
    if (y < 0 || y >= end + extend || y >= H) return -1;
    return y;
}

// One pixel of a pass, the shader's own way (denoiser.frag:38-73 tap by tap): the body of k_denoise.
template <bool PHI_INF>
__device__ __forceinline__ uchar4 denoise_pixel(const DenoiseParams& P, int px, int py)
{
    const bool shipped = (P.mode & 1) == VRT_DENOISE_AS_SHIPPED;
    const int ntaps = shipped ? 3 : 9;
    float sw = P.step_width;
    float sw2 = sw * sw;
    Guides s, o;
    texel_guides(P, px, py, s);
    float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float total = 0.0f;
    for (int i = 0; i < ntaps; i++) {
        int tx, ty; float kern;
        if (shipped) {        // std140 aliasing: taps (-1,-1)*G2, (1,-1)*G0, (0,0)*G2
            tx = i == 0 ? -1 : (i == 1 ? 1 : 0); ty = i == 2 ? 0 : -1;
            kern = i == 1 ? kGauss0 : kGauss2;
        } else {
            tx = i % 3 - 1; ty = i / 3 - 1;
            int r2 = tx * tx + ty * ty;
            kern = r2 == 0 ? kGauss0 : (r2 == 1 ? kGauss1 : kGauss2);
        }
        sample_guides(P, px, py, (float)tx * sw, (float)ty * sw, o);
        float w = 1.0f;
        if (!PHI_INF) {
            float pw = edge_weight(dist2_4(s.p, o.p), P.phi_pos);
            float cw = 1.0f, nw = 1.0f;
            if (pw != 0.0f) {                         // a zero factor makes w = +0 whatever the other two are (all finite)
                cw = edge_weight(dist2_4(s.c, o.c), P.phi_color);
                float dn = dist2_4(s.n, o.n);
                nw = dn == 0.0f ? 1.0f : edge_weight(fmaxf(dn / sw2, 0.0f), P.phi_normal);
            }
            w = (cw * nw) * pw;
        }
        for (int k = 0; k < 4; k++) sum[k] += (o.c[k] * w) * kern;
        total += w * kern;
    }
    uchar4 out;
    out.x = unorm8(sum[0] / total); out.y = unorm8(sum[1] / total);
    out.z = unorm8(sum[2] / total); out.w = unorm8(sum[3] / total);
    return out;
}

// One TAP of a weighted pass with an integral tap offset R, the shader's own way: the tap's weight (cw * nw) * pw and its
// colour texel (k_denoise_ver's redone pixels: nine lanes share a pixel, one tap each, so that a pixel costs one tap's
// chain of dependent instructions instead of nine; the sums are then taken by one lane in the shader's order).  The
// arithmetic is denoise_pixel<false>'s, operation for operation.
__device__ __forceinline__ float guides_tap_weight(float phi_color, float phi_normal, float phi_pos, float sw, const Guides& s, const Guides& o)
{
    const float sw2 = sw * sw;
    float pw = edge_weight(dist2_4(s.p, o.p), phi_pos);
    float cw = 1.0f, nw = 1.0f;
    if (pw != 0.0f) {
        cw = edge_weight(dist2_4(s.c, o.c), phi_color);
        float dn = dist2_4(s.n, o.n);
        nw = dn == 0.0f ? 1.0f : edge_weight(fmaxf(dn / sw2, 0.0f), phi_normal);
    }
    return (cw * nw) * pw;
}
__device__ __forceinline__ float denoise_tap_weight(const DenoiseParams& P, int px, int py, int tx, int ty, int R, uint32_t& color)
{
    Guides s, o;
    texel_guides(P, px + tx * R, py + ty * R, o);         // (both texels requested before either is used)
    texel_guides(P, px, py, s);
    {
        int x = px + tx * R, y = py + ty * R;
        x = x < 0 ? 0 : (x > P.W - 1 ? P.W - 1 : x);
        y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);
        color = reinterpret_cast<const uint32_t*>(P.color_in)[(size_t)y * (size_t)P.W + (size_t)x];
    }
    return guides_tap_weight(P.phi_color, P.phi_normal, P.phi_pos, P.step_width, s, o);
}

template <bool PHI_INF>
__global__ __launch_bounds__(256) void k_denoise(const DenoiseParams P)
{
    int px = blockIdx.x * 64 + (threadIdx.x & 63);
    int r = blockIdx.y * 4 + (threadIdx.x >> 6);
    int py = strip_row(P.sh, P.extend, r, P.H);
    if (py < 0 || px >= P.W) return;
    reinterpret_cast<uchar4*>(P.color_out)[(size_t)py * (size_t)P.W + (size_t)px] = denoise_pixel<PHI_INF>(P, px, py);
}

typedef float v2f __attribute__((ext_vector_type(2)));

// ---- the exact weights of TWO taps at a time, without branches -------------------------------------------------------------
// The weighted pass is bound by instruction issue (two Cephes exponentials and two correctly rounded divisions per channel
// and tap, behind data-dependent shortcuts whose short EXEC-masked blocks cost as much as they save).  Every shortcut of
// edge_weight() is the value the long way round gives anyway -- exp_spec(-0) = 1 exactly, a factor 0 makes the product +0 --
// so the long way round, for two taps at once in the two halves of packed fp32 instructions (v_pk_mul / v_pk_add / v_pk_fma:
// the same IEEE operations element-wise, never contracted), is the same arithmetic: 12 packed exponentials per pixel
// instead of 27 scalar ones.

// RN(a / b) element-wise for a pass-uniform b with r = RN(1 / b): q0 = a r and two residual corrections -- the core of the
// compiler's own IEEE division sequence (which refines an approximate reciprocal to within an ulp, multiplies, and corrects
// twice), without its range scaling: exact while no intermediate leaves the normal range, i.e. for b in [2^-20, 2^20] (the
// host checks) and a = 0 or a in [2^-90, 2^90] (the caller checks; decoded 8-bit guides cannot leave it).
__device__ __forceinline__ v2f div_uniform2(v2f a, float b, float r)
{
    const v2f nb = {-b, -b}, rr = {r, r};
    v2f q = a * rr;
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(nb, q, a), rr, q);
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(nb, q, a), rr, q);
    return q;
}

// min(exp_spec(-q), 1) element-wise for q >= 0 (vrt_spec.h exp_spec, operation for operation)
__device__ __forceinline__ v2f edge_weight2(v2f q)
{
    const v2f x0 = -q;
    const v2f fx = __builtin_elementwise_floor(x0 * 1.44269504088896341f + 0.5f);
    v2f x = x0 - fx * 0.693359375f;
    x = x - fx * -2.12194440e-4f;
    const v2f z = x * x;
    const v2f p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x
                     + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    const int n0 = (int)fx.x, n1 = (int)fx.y;
    const v2f sc = {__uint_as_float(((uint32_t)n0 + 127u) << 23), __uint_as_float(((uint32_t)n1 + 127u) << 23)};
    const v2f e = p * sc;
    v2f w;
    w.x = x0.x < -87.0f ? 0.0f : fminf(e.x, 1.0f);
    w.y = x0.y < -87.0f ? 0.0f : fminf(e.y, 1.0f);
    return w;
}

// |a - b|^2 in the order of dist2_4, the two halves of each float4 in one packed instruction
__device__ __forceinline__ float dist2_4pk(const float4& a, const float4& b)
{
    const v2f t01 = (v2f){a.x, a.y} - (v2f){b.x, b.y}, t23 = (v2f){a.z, a.w} - (v2f){b.z, b.w};
    const v2f q01 = t01 * t01, q23 = t23 * t23;
    return ((q01.x + q01.y) + q23.x) + q23.y;
}

// LDS-tiled form for integral stepWidth: a workgroup owns 64x4 pixels; the guides of that tile plus a halo of
// R = stepWidth pixels are fetched, decoded ONCE and parked in LDS as three float4 planes (48 B per pixel), so each
// pixel's guides are read from HBM/L2 once per pass instead of once per tap that lands on it (9x), and the 8-bit
// decodes are not repeated per tap.  Same arithmetic on the same decoded values as k_denoise.
template <bool PHI_INF, bool SHIPPED, bool FAST = false, bool PACKED = false>
__global__ __launch_bounds__(256) void k_denoise_lds(const DenoiseParams P, int R)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds_g[];
    const int RW = 64 + 2 * R, RH = 4 + 2 * R, NP = RW * RH;
    float4* lc = lds_g; float4* ln = lds_g + NP; float4* lp = lds_g + 2 * NP;
    const int x0 = blockIdx.x * 64, r0 = blockIdx.y * 4;
    // rows of one block are consecutive frame rows (checked by the launcher); a single rank owns every row in order
    const bool whole = P.sh.nranks == 1 && P.extend == 0;
    // Sharded: the four rows of a block lie in one extended strip (per % 4 == 0), but the strip's first `extend` rows
    // do not exist above the top of the frame (and its last ones may not below the bottom), so the block's frame row
    // is taken from its first row that exists -- not from row r0, which for extend % 4 == 2 is missing while r0 + 2
    // and r0 + 3 are frame rows 0 and 1.  y0 may be negative; the staging clamps and the row test below masks.
    int y0 = -1;
    bool any_row = false;
    if (whole) { y0 = r0; any_row = r0 < P.H; }
    else {
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const int yk = strip_row(P.sh, P.extend, r0 + k, P.H);
            if (yk >= 0) { y0 = yk - k; any_row = true; }
        }
    }
    if (!any_row) return;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    // staging: texel t = threadIdx.x + 256 k of the haloed tile, k = 0, 1, ... -- every thread gets the same number of
    // texels (+-1); (cx, cy) = (t % RW, t / RW) is kept without divisions: RW is 66..74, so threadIdx.x / RW is 0..3
    {
        const int t0 = (int)threadIdx.x;
        int cy = (t0 >= RW ? 1 : 0) + (t0 >= 2 * RW ? 1 : 0) + (t0 >= 3 * RW ? 1 : 0);
        int cx = t0 - cy * RW;
        const int dy = 256 >= 3 * RW + RW ? 4 : 3;             // 256 / RW (RW <= 64: 4 never happens; RW in 66..74: 3)
        const int dx = 256 - dy * RW;
        for (int t = t0; t < NP; t += 256) {
            int x = x0 - R + cx, y = y0 - R + cy;
            x = x < 0 ? 0 : (x > P.W - 1 ? P.W - 1 : x);
            y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);
            const size_t i = (size_t)y * (size_t)P.W + (size_t)x;
            const uchar4 c = reinterpret_cast<const uchar4*>(P.color_in)[i];
            lc[t] = make_float4(decode_unorm8(c.x), decode_unorm8(c.y), decode_unorm8(c.z), decode_unorm8(c.w));
            if (!PHI_INF) {                                   // pass 0 weighs every tap 1: only the colour is ever read
                const char4 n = reinterpret_cast<const char4*>(P.normal)[i];
                ln[t] = make_float4(decode_snorm8(n.x), decode_snorm8(n.y), decode_snorm8(n.z), decode_snorm8(n.w));
                lp[t] = reinterpret_cast<const float4*>(P.position)[i];
            }
            cx += dx; cy += dy;
            if (cx >= RW) { cx -= RW; cy++; }
        }
    }
    __syncthreads();
    const int px = x0 + lx, py = y0 + ly;
    if (px >= P.W || py < 0 || py >= P.H) return;
    if (!whole && strip_row(P.sh, P.extend, r0 + ly, P.H) != py) return;   // above / past the end of the strip or frame

    constexpr int ntaps = SHIPPED ? 3 : 9;
    const float sw = P.step_width, sw2 = sw * sw;
    const int c0 = (ly + R) * RW + (lx + R);
    const float4 sc = lc[c0], sn = PHI_INF ? sc : ln[c0], sp = PHI_INF ? sc : lp[c0];
    const float s_c[4] = {sc.x, sc.y, sc.z, sc.w}, s_n[4] = {sn.x, sn.y, sn.z, sn.w}, s_p[4] = {sp.x, sp.y, sp.z, sp.w};
    float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float total = 0.0f;
    const int rowoff = R * RW;
    if constexpr (PACKED && !PHI_INF && !FAST) {
        // taps in the shader's order, two at a time; the centre tap (all three distances are 0 or NaN: every weight is 1)
        // between them where the order has it
        v2f s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
        constexpr int npairs = SHIPPED ? 1 : 4;
#pragma unroll 1
        for (int j = 0; j < npairs; j++) {
            int ta, tb; float ka, kb;                          // tap offsets (in units of R, relative to c0) and kernel weights
            if (SHIPPED) { ta = -rowoff - R; tb = -rowoff + R; ka = kGauss2; kb = kGauss0; }
            else if (j == 0) { ta = -rowoff - R; tb = -rowoff; ka = kGauss2; kb = kGauss1; }
            else if (j == 1) { ta = -rowoff + R; tb = -R; ka = kGauss2; kb = kGauss1; }
            else if (j == 2) { ta = R; tb = rowoff - R; ka = kGauss1; kb = kGauss2; }
            else { ta = rowoff; tb = rowoff + R; ka = kGauss1; kb = kGauss2; }
            if (!SHIPPED && j == 2) {                          // tap 4, the centre: w = 1, kern = 1
                s01 += (v2f){sc.x, sc.y}; s23 += (v2f){sc.z, sc.w};
                total += 1.0f;
            }
            const float4 oca = lc[c0 + ta], ocb = lc[c0 + tb], opa = lp[c0 + ta], opb = lp[c0 + tb], ona = ln[c0 + ta], onb = ln[c0 + tb];
            const v2f dp = {dist2_4pk(sp, opa), dist2_4pk(sp, opb)};
            const v2f dc = {dist2_4pk(sc, oca), dist2_4pk(sc, ocb)};
            const v2f dn = {dist2_4pk(sn, ona), dist2_4pk(sn, onb)};
            // positions are the caller's floats: a distance outside the range the short division is exact for (tiny, huge,
            // inf, NaN) sends the wave through edge_weight() for this pair
            const uint32_t ua = __float_as_uint(dp.x), ub = __float_as_uint(dp.y);
            const bool odd = (ua != 0u && ua - 0x12800000u > 0x6C800000u - 0x12800000u) || (ub != 0u && ub - 0x12800000u > 0x6C800000u - 0x12800000u);
            v2f w;
            if (__builtin_expect(__ballot(odd) != 0ull, 0)) {
                const float pa = edge_weight(dp.x, P.phi_pos), pb = edge_weight(dp.y, P.phi_pos);
                float ca = 1.0f, na = 1.0f, cb = 1.0f, nb = 1.0f;
                if (pa != 0.0f) { ca = edge_weight(dc.x, P.phi_color); na = dn.x == 0.0f ? 1.0f : edge_weight(fmaxf(dn.x / sw2, 0.0f), P.phi_normal); }
                if (pb != 0.0f) { cb = edge_weight(dc.y, P.phi_color); nb = dn.y == 0.0f ? 1.0f : edge_weight(fmaxf(dn.y / sw2, 0.0f), P.phi_normal); }
                w = (v2f){(ca * na) * pa, (cb * nb) * pb};
            } else {
                // a channel in which all 64 pixels agree with both their taps (sky: every position is 0; a flat wall: one
                // normal) has weight exp(-0) = 1 throughout: one compare and a branch the whole wave takes or not
                const v2f one = {1.0f, 1.0f};
                v2f pw = one, cw = one, nw = one;
                if (__ballot((ua | ub) != 0u) != 0ull) pw = edge_weight2(div_uniform2(dp, P.phi_pos, P.rp));
                if (__ballot((__float_as_uint(dc.x) | __float_as_uint(dc.y)) != 0u) != 0ull) cw = edge_weight2(div_uniform2(dc, P.phi_color, P.rc));
                if (__ballot((__float_as_uint(dn.x) | __float_as_uint(dn.y)) != 0u) != 0ull)
                    nw = edge_weight2(div_uniform2(div_uniform2(dn, sw2, P.rs), P.phi_normal, P.rn));
                w = (cw * nw) * pw;
            }
            s01 += ((v2f){oca.x, oca.y} * w.x) * ka; s23 += ((v2f){oca.z, oca.w} * w.x) * ka; total += w.x * ka;
            s01 += ((v2f){ocb.x, ocb.y} * w.y) * kb; s23 += ((v2f){ocb.z, ocb.w} * w.y) * kb; total += w.y * kb;
        }
        if (SHIPPED) {                                         // tap 2, the centre: w = 1, kern = G2
            s01 += (v2f){sc.x, sc.y} * kGauss2; s23 += (v2f){sc.z, sc.w} * kGauss2;
            total += kGauss2;
        }
        sum[0] = s01.x; sum[1] = s01.y; sum[2] = s23.x; sum[3] = s23.y;
    } else {
    // pass 0 unrolls into nine LDS reads and 72 multiply-adds; the weighted taps stay a loop (unrolled they need 72
    // VGPRs and 14 KB of code, and measured 7 % slower)
    constexpr int kUnroll = PHI_INF ? 9 : 1;
#pragma unroll kUnroll
    for (int i = 0; i < ntaps; i++) {
        int tx, ty; float kern;
        if (SHIPPED) {
            tx = i == 0 ? -1 : (i == 1 ? 1 : 0); ty = i == 2 ? 0 : -1;
            kern = i == 1 ? kGauss0 : kGauss2;
        } else {
            tx = i % 3 - 1; ty = i / 3 - 1;
            int r2 = tx * tx + ty * ty;
            kern = r2 == 0 ? kGauss0 : (r2 == 1 ? kGauss1 : kGauss2);
        }
        const int ci = c0 + ty * rowoff + tx * R;
        const float4 oc = lc[ci];
        const float o_c[4] = {oc.x, oc.y, oc.z, oc.w};
        float w = 1.0f;
        if (!PHI_INF && FAST) {
            // the product of the three weights as one exponential (each argument is <= 0, so no factor exceeds 1 and the
            // shader's min(., 1) has nothing to do): three multiply-adds and one v_exp_f32 instead of three divisions and
            // three polynomial exponentials
            const float4 op = lp[ci], on = ln[ci];
            const float o_p[4] = {op.x, op.y, op.z, op.w}, o_n[4] = {on.x, on.y, on.z, on.w};
            const float e = __builtin_fmaf(dist2_4(s_p, o_p), P.kp, __builtin_fmaf(dist2_4(s_c, o_c), P.kc, fmaxf(dist2_4(s_n, o_n), 0.0f) * P.kn));
            w = __builtin_amdgcn_exp2f(-e);
        } else if (!PHI_INF) {
            const float4 op = lp[ci];
            const float o_p[4] = {op.x, op.y, op.z, op.w};
            float pw = edge_weight(dist2_4(s_p, o_p), P.phi_pos);
            float cw = 1.0f, nw = 1.0f;
            if (pw != 0.0f) {
                const float4 on = ln[ci];
                const float o_n[4] = {on.x, on.y, on.z, on.w};
                cw = edge_weight(dist2_4(s_c, o_c), P.phi_color);
                float dn = dist2_4(s_n, o_n);
                nw = dn == 0.0f ? 1.0f : edge_weight(fmaxf(dn / sw2, 0.0f), P.phi_normal);
            }
            w = (cw * nw) * pw;
        }
        for (int k = 0; k < 4; k++) sum[k] += (o_c[k] * w) * kern;
        total += w * kern;
    }
    }
    uchar4 out;
    if (PHI_INF) {
        // sums are 0 or >= 1/255 * 0.77 and total is the fixed sum of the tap weights (3.3 .. 7.7): no operand or
        // quotient of these four divisions is anywhere near the range where the IEEE sequence rescales, so its core --
        // reciprocal refined once, then two residual corrections per quotient -- can share the reciprocal (23 VALU
        // ops instead of 40) and still round every quotient correctly
        const float r0 = __builtin_amdgcn_rcpf(total);
        const float r = __builtin_fmaf(__builtin_fmaf(-total, r0, 1.0f), r0, r0);
        float q[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float q0 = sum[k] * r;
            float q1 = __builtin_fmaf(__builtin_fmaf(-total, q0, sum[k]), r, q0);
            q[k] = __builtin_fmaf(__builtin_fmaf(-total, q1, sum[k]), r, q1);
        }
        out.x = unorm8(q[0]); out.y = unorm8(q[1]); out.z = unorm8(q[2]); out.w = unorm8(q[3]);
    } else if (FAST) {
        const float r = __builtin_amdgcn_rcpf(total);
        out.x = unorm8(sum[0] * r); out.y = unorm8(sum[1] * r); out.z = unorm8(sum[2] * r); out.w = unorm8(sum[3] * r);
    } else {
        out.x = unorm8(sum[0] / total); out.y = unorm8(sum[1] / total);
        out.z = unorm8(sum[2] / total); out.w = unorm8(sum[3] / total);
    }
    reinterpret_cast<uchar4*>(P.color_out)[(size_t)py * (size_t)P.W + (size_t)px] = out;
}

// VRT_DENOISE_FAST on a whole frame (one rank, integral stepWidth, a weighted pass): a workgroup owns 64 x TH pixels, four rows
// at a time per wave, so that the guides of the tile + halo are fetched and decoded once for TH rows instead of four (a
// halo of R = 3 rows above and below makes a 4-row tile read 2.5x its own rows, a 16-row tile 1.4x); the weights are one
// hardware exponential per tap (see VRT_DENOISE_FAST in vrt.h).
// |a - b|^2 of two float4 in packed fp32 operations (v_pk_add / v_pk_mul / v_pk_fma: two lanes' worth per instruction);
// fused and re-associated -- the fast mode states a tolerance, not a rounding
__device__ __forceinline__ float dist2_pk(const float4& a, const float4& b)
{
    const v2f d0 = (v2f){a.x, a.y} - (v2f){b.x, b.y}, d1 = (v2f){a.z, a.w} - (v2f){b.z, b.w};
    const v2f q = __builtin_elementwise_fma(d1, d1, d0 * d0);
    return q.x + q.y;
}

__device__ __forceinline__ constexpr int r2_of(int tx, int ty) { return tx * tx + ty * ty; }
// as-shipped taps: (-1,-1) * G2, (1,-1) * G0, (0,0) * G2
__device__ __forceinline__ constexpr float shipped_lk(int i) { return i == 1 ? 0.0f : 0.36067376022224085f; }

template <bool SHIPPED, int TH>
__global__ __launch_bounds__(256) void k_denoise_fast(const DenoiseParams P, int R)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds_g[];
    const int RW = 64 + 2 * R, RH = TH + 2 * R, NP = RW * RH;
    float4* lc = lds_g; float4* ln = lds_g + NP; float4* lp = lds_g + 2 * NP;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * TH;
    {
        int cy = (int)threadIdx.x / RW, cx = (int)threadIdx.x - cy * RW;       // (RW >= 66: cy is 0..3)
        const int dy = 256 / RW, dx = 256 - dy * RW;
        for (int t = (int)threadIdx.x; t < NP; t += 256) {
            int x = x0 - R + cx, y = y0 - R + cy;
            x = x < 0 ? 0 : (x > P.W - 1 ? P.W - 1 : x);
            y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);
            const size_t i = (size_t)y * (size_t)P.W + (size_t)x;
            const uchar4 c = reinterpret_cast<const uchar4*>(P.color_in)[i];
            const char4 n = reinterpret_cast<const char4*>(P.normal)[i];
            // the CODES as floats (SNORM -128 = -127): 1/255 and 1/127 ride in the distances' scale factors, and the output is a
            // mean of codes already
            lc[t] = make_float4((float)c.x, (float)c.y, (float)c.z, (float)c.w);
            ln[t] = make_float4(fmaxf((float)n.x, -127.0f), fmaxf((float)n.y, -127.0f), fmaxf((float)n.z, -127.0f), fmaxf((float)n.w, -127.0f));
            lp[t] = reinterpret_cast<const float4*>(P.position)[i];
            cx += dx; cy += dy;
            if (cx >= RW) { cx -= RW; cy++; }
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, px = x0 + lx;
    if (px >= P.W) return;
    constexpr int ntaps = SHIPPED ? 3 : 9;
    const int rowoff = R * RW;
    const float kc = P.kc * (1.0f / (255.0f * 255.0f)), kn = P.kn * (1.0f / (127.0f * 127.0f));
    for (int ly = (int)(threadIdx.x >> 6); ly < TH; ly += 4) {
        const int py = y0 + ly;
        if (py >= P.H) break;
        const int c0 = (ly + R) * RW + (lx + R);
        const float4 sc = lc[c0], sn = ln[c0], sp = lp[c0];
        v2f s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
        float total = 0.0f;
#pragma unroll
        for (int i = 0; i < ntaps; i++) {
            int tx, ty; float kern;
            if (SHIPPED) {
                tx = i == 0 ? -1 : (i == 1 ? 1 : 0); ty = i == 2 ? 0 : -1;
                kern = i == 1 ? kGauss0 : kGauss2;
            } else {
                tx = i % 3 - 1; ty = i / 3 - 1;
                const int r2 = tx * tx + ty * ty;
                kern = r2 == 0 ? kGauss0 : (r2 == 1 ? kGauss1 : kGauss2);
            }
            if (tx == 0 && ty == 0) {                            // the centre tap: every distance is 0, its weight is the kernel's
                s01 += (v2f){sc.x, sc.y} * kern; s23 += (v2f){sc.z, sc.w} * kern; total += kern;
                continue;
            }
            const int ci = c0 + ty * rowoff + tx * R;
            const float4 oc = lc[ci], op = lp[ci], on = ln[ci];
            // the kernel weight rides in the exponent: w * kern = exp2(-(e - log2 kern))
            const float lk = r2_of(tx, ty) == 0 ? 0.0f : (r2_of(tx, ty) == 1 ? 0.18033688011112042f : 0.36067376022224085f);   // -log2(G1), -log2(G2)
            const float e = __builtin_fmaf(dist2_pk(sp, op), P.kp, __builtin_fmaf(dist2_pk(sc, oc), kc, __builtin_fmaf(dist2_pk(sn, on), kn, SHIPPED ? shipped_lk(i) : lk)));
            const float wk = __builtin_amdgcn_exp2f(-e);
            const v2f w2 = {wk, wk};
            s01 = __builtin_elementwise_fma((v2f){oc.x, oc.y}, w2, s01);
            s23 = __builtin_elementwise_fma((v2f){oc.z, oc.w}, w2, s23);
            total += wk;
        }
        const float r = __builtin_amdgcn_rcpf(total);
        // a weighted mean of codes: round half up, clamp (the weights are positive, the mean cannot leave 0..255 by more than rounding)
        uchar4 out;
        out.x = (uint8_t)fminf(floorf(fmaf(s01.x, r, 0.5f)), 255.0f); out.y = (uint8_t)fminf(floorf(fmaf(s01.y, r, 0.5f)), 255.0f);
        out.z = (uint8_t)fminf(floorf(fmaf(s23.x, r, 0.5f)), 255.0f); out.w = (uint8_t)fminf(floorf(fmaf(s23.y, r, 0.5f)), 255.0f);
        reinterpret_cast<uchar4*>(P.color_out)[(size_t)py * (size_t)P.W + (size_t)px] = out;
    }
}

#define VRT_DEN_FIXCAP 1024
// The listed pixels of a workgroup, the shader's own way (k_denoise_ver, k_denoise_pair): nine lanes per listed pixel, a tap
// each (denoise_tap_weight); then four of them a channel each, the sums in the shader's order.  More flagged than the list holds -- hostile
// input -- or `all`: every pixel of the segment instead (columns xo .. xo + ow - 1, rows ys .. ye - 1; pixels that were sure get
// the value they already have).  Called by every thread of the workgroup, behind the barrier that made n and the list final.
template <bool SHIPPED, bool PASS0>
__device__ __forceinline__ void denoise_redo(const DenoiseParams& P, int R, uint32_t n, bool all, int xo, int ow, int ys, int ye,
                                             const uint32_t* fl_px, float (*fx_w)[9], uint32_t (*fx_c)[9], uint32_t first = 0u)
{
    constexpr int ntaps = SHIPPED ? 3 : 9;
    const bool overflow = all || n > VRT_DEN_FIXCAP;
    const uint32_t entries = overflow ? 64u * (uint32_t)(ye - ys) : n;
    const bool shipped = SHIPPED;
    // nine lanes per listed pixel: a tap each (denoise_tap_weight), then four of them a channel each -- the sums in the shader's order
    const uint32_t per = blockDim.x / 9u, grp = threadIdx.x / 9u;          // pixels per round (fx_w, fx_c hold that many)
    const int tap = (int)(threadIdx.x - grp * 9u);
    int tx, ty;
    if (shipped) { tx = tap == 0 ? -1 : (tap == 1 ? 1 : 0); ty = tap == 2 ? 0 : -1; }
    else { tx = tap % 3 - 1; ty = tap / 3 - 1; }
    for (uint32_t base = overflow ? 0u : first; base < entries; base += per) {        // (`first`: entries below it have been done)
        const uint32_t e = base + grp;
        bool live = grp < per && e < entries;
        uint32_t idx = 0u;
        if (live) {
            if (overflow) { const uint32_t qx = (uint32_t)xo + (e & 63u); live = (int)(e & 63u) < ow && qx < (uint32_t)P.W; idx = (uint32_t)(ys + (int)(e >> 6)) * (uint32_t)P.W + qx; }
            else idx = fl_px[e];
        }
        const int py = (int)(idx / (uint32_t)P.W), qx = (int)(idx - (uint32_t)py * (uint32_t)P.W);
        if (live && tap < ntaps) {
            uint32_t col;
            float w = 1.0f;
            if (PASS0) {
                int x = qx + tx * R, y = py + ty * R;
                x = x < 0 ? 0 : (x > P.W - 1 ? P.W - 1 : x);
                y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);
                col = reinterpret_cast<const uint32_t*>(P.color_in)[(size_t)y * (size_t)P.W + (size_t)x];
            } else w = denoise_tap_weight(P, qx, py, tx, ty, R, col);
            fx_w[grp][tap] = w; fx_c[grp][tap] = col;
        }
        __syncthreads();
        if (live && tap < 4) {                                   // channel `tap` of the pixel
            float sum = 0.0f, total = 0.0f;
#pragma unroll
            for (int i = 0; i < ntaps; i++) {
                float kern;
                if (shipped) kern = i == 1 ? kGauss0 : kGauss2;
                else { const int ux = i % 3 - 1, uy = i / 3 - 1, r2 = ux * ux + uy * uy; kern = r2 == 0 ? kGauss0 : (r2 == 1 ? kGauss1 : kGauss2); }
                const float w = fx_w[grp][i];
                const float oc = decode_unorm8((fx_c[grp][i] >> (8 * tap)) & 0xFFu);
                sum += (oc * w) * kern;
                total += w * kern;
            }
            P.color_out[(size_t)idx * 4u + (size_t)tap] = unorm8(sum / total);
        }
        __syncthreads();
    }
}

// ---- the verified pass -----------------------------------------------------------------------------------------------------
// vrt_denoise_bound.h: of a weighted pass only floor(mean * 255 + 0.5) is ever seen.  This kernel computes the mean cheaply --
// code distances as exact integers (three v_dot4_u32_u8: |a - b|^2 = a.a + b.b - 2 a.b over the four bytes of a texel; normals
// biased by 128, which differences do not see), the three edge-stopping weights and the kernel weight as ONE hardware
// exponential, fused accumulation, a reciprocal -- and every pixel one of whose channels lies within P.guard codes of a
// rounding boundary (NaN included: the comparison fails) is evaluated once more at the end, the shader's own way
// (denoise_pixel), by the workgroup that found it.  What the kernel leaves in color_out is the exact kernels' output bit for
// bit (tests/test_gpu_denoise.py).  FLAG = false is VRT_DENOISE_FAST: the same arithmetic, nothing redone (<= 1 code away).
//
// A workgroup owns a column strip of 64 pixels and `seg_rows` rows of it and walks DOWN the strip four rows at a time (one row
// per wave) through a ring of rows in LDS -- position 16 B, colour and (biased) normal codes 8 B per texel --: while a group
// of rows is being filtered the four rows the next group adds are already on their way from memory into registers, and go
// into the ring slots of the four rows the group no longer needs.  A texel is fetched once per strip and segment (1.1 - 1.3x
// the planes, against 1.9x for 64 x 8 tiles with their halo), the fetch latency hides under the arithmetic, and the launch is
// ONE round of workgroups that all end together.  RT: the tap offset at compile time (LDS offsets become immediates), 0: any.
// u - 2 v (a shift and a subtraction).  Not as v_mad_i32_i24 through inline assembly: the result of a v_dot4 may not be read by
// another vector instruction for three wait states on gfx950, and only instructions the compiler knows get their s_nops -- an
// asm block here read stale registers; the compiler's own 24-bit multiply-add sign-extends first and is three instructions.
__device__ __forceinline__ int mad24_minus2(uint32_t v, uint32_t u) { return (int)(u - 2u * v); }
template <bool SHIPPED, bool FLAG, int RT, bool PASS0 = false>
__global__ __launch_bounds__(256) void k_denoise_ver(const DenoiseParams P, int Rrt, int seg_rows, int segs_per_strip)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds_g[];
    __shared__ uint32_t fl_n, fl_w4;
    __shared__ uint32_t fl_px[FLAG ? VRT_DEN_FIXCAP : 1];
    __shared__ float fx_w[FLAG ? 28 : 1][9];
    __shared__ uint32_t fx_c[FLAG ? 28 : 1][9];
#ifdef VRT_K3_STAMPS
    // (development build, tools/exp_k3_timeline.py: a workgroup's start / ring filled / rows done / end on the 100 MHz clock, written
    // over the first words of color_out when it ends -- the image is garbage)
    uint32_t k3_t[4] = {(uint32_t)wall_clock64(), 0u, 0u, 0u};
    auto k3_stamp = [&]() {
        if (threadIdx.x == 0) {
            uint32_t* o = reinterpret_cast<uint32_t*>(P.color_out) + 4u * (blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = k3_t[0]; o[1] = k3_t[1]; o[2] = k3_t[2]; o[3] = (uint32_t)wall_clock64();
        }
    };
#endif
    const int R = RT ? RT : Rrt;
    const int RW = 64 + 2 * R;
    const int U = (4 + 2 * R + 3) / 4;                    // units of four rows a group of four output rows reads
    const int NR = 4 * (U + 1);                           // ring: those + the unit on its way in
    float4* lp = lds_g; uint2* lq = reinterpret_cast<uint2*>(lds_g + NR * RW);
    uint32_t* lc = reinterpret_cast<uint32_t*>(lds_g);      // PASS0 (phi = +inf, every weight exactly 1): the ring holds the colour codes only
    // the rows of this workgroup: segment j of the rank's local strip k, the strip taken with the `extend` rows either side that
    // this pass must also produce (strip_row's rows; one rank: the one strip is the frame)
    const int x0 = blockIdx.x * 64;
    int ys, ye;
    {
        const int k = (int)blockIdx.y / segs_per_strip, j = (int)blockIdx.y - k * segs_per_strip;
        const int g = k * P.sh.nranks + P.sh.rank;
        int r0 = g * P.sh.strip_rows - P.extend, r1 = (g + 1) * P.sh.strip_rows;
        r1 = (r1 < P.H ? r1 : P.H) + P.extend;
        r0 = r0 < 0 ? 0 : r0; r1 = r1 < P.H ? r1 : P.H;
        ys = r0 + j * seg_rows;
        ye = ys + seg_rows < r1 ? ys + seg_rows : r1;
    }
    if (ys >= ye) return;                                 // uniform per workgroup (a strip's last segment may be empty)
    const int groups = (ye - ys + 3) >> 2;
    if (threadIdx.x == 0) { fl_n = 0u; fl_w4 = 0u; }
    __syncthreads();
    // a thread's two texels of a unit (4 * RW <= 512 of them): row in the unit, clamped frame column
    const int tA = (int)threadIdx.x, tB = tA + 256;
    const int rA = tA / RW, cA = tA - rA * RW, rB = tB / RW, cB = tB - rB * RW;
    const bool hasB = tB < 4 * RW;
    int xA = x0 - R + cA, xB = x0 - R + cB;
    xA = xA < 0 ? 0 : (xA > P.W - 1 ? P.W - 1 : xA);
    xB = xB < 0 ? 0 : (xB > P.W - 1 ? P.W - 1 : xB);
    const uint32_t* const gc = reinterpret_cast<const uint32_t*>(P.color_in);
    const uint32_t* const gn = reinterpret_cast<const uint32_t*>(P.normal);
    const float4* const gp = reinterpret_cast<const float4*>(P.position);
    float4 pA = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pB = pA;
    uint32_t colA, nrmA = 0u, colB = 0u, nrmB = 0u;
    auto fetch = [&](int unit) {                           // unit u = relative rows 4u .. 4u + 3 = frame rows ys - R + 4u ...
        int yA = ys - R + 4 * unit + rA, yB = ys - R + 4 * unit + rB;
        yA = yA < 0 ? 0 : (yA > P.H - 1 ? P.H - 1 : yA);
        yB = yB < 0 ? 0 : (yB > P.H - 1 ? P.H - 1 : yB);
        const uint32_t iA = (uint32_t)yA * (uint32_t)P.W + (uint32_t)xA, iB = (uint32_t)yB * (uint32_t)P.W + (uint32_t)xB;
        if (PASS0) { colA = gc[iA]; if (hasB) colB = gc[iB]; return; }
        pA = gp[iA]; colA = gc[iA]; nrmA = gn[iA];
        if (hasB) { pB = gp[iB]; colB = gc[iB]; nrmB = gn[iB]; }
    };
    auto bias = [](uint32_t n) {
        // SNORM code -128 decodes like -127 (max(c / 127, -1)): bytes 0x80 become 0x81; then every byte biased by 128
        uint32_t z = n ^ 0x80808080u;                                                   // bytes that were 0x80 are 0 now
        z = ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu);                     // 0x80 exactly in those bytes
        return (n | (z >> 7)) ^ 0x80808080u;
    };
    auto stash = [&](int unit) {
        const int slot = (unit % (U + 1)) * 4;
        // (a colour alpha or a position w other than +-0: the rows take their general form from the next barrier on)
        if (((colA | colB) >> 24) != 0u || ((__float_as_uint(pA.w) | __float_as_uint(pB.w)) << 1) != 0u) fl_w4 = 1u;
        if (PASS0) { lc[(slot + rA) * RW + cA] = colA; if (hasB) lc[(slot + rB) * RW + cB] = colB; return; }
        lp[(slot + rA) * RW + cA] = pA; lq[(slot + rA) * RW + cA] = make_uint2(colA, bias(nrmA));
        if (hasB) { lp[(slot + rB) * RW + cB] = pB; lq[(slot + rB) * RW + cB] = make_uint2(colB, bias(nrmB)); }
    };
    {
        // the first U units (U <= 4), all requested before the first is stored: one round trip to memory, not U (every workgroup
        // of the launch stands here at the same time: nothing else hides them)
        float4 qpA[4], qpB[4]; uint32_t qcA[4], qnA[4], qcB[4], qnB[4];
#pragma unroll
        for (int u = 0; u < 4; u++) if (u < U) { fetch(u); qpA[u] = pA; qpB[u] = pB; qcA[u] = colA; qnA[u] = nrmA; qcB[u] = colB; qnB[u] = nrmB; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (u < U) { pA = qpA[u]; pB = qpB[u]; colA = qcA[u]; nrmA = qnA[u]; colB = qcB[u]; nrmB = qnB[u]; stash(u); }
    }
    __syncthreads();

#ifdef VRT_K3_STAMPS
    k3_t[1] = (uint32_t)wall_clock64();
#endif
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lx = threadIdx.x & 63, px = x0 + lx;
    constexpr int ntaps = SHIPPED ? 3 : 9;
    const float kc = P.vkc, kn = P.vkn, kp = P.vkp;
    const float half_guard = 0.5f - P.guard;
    // One output row of a wave.  W4 = false: no texel in the ring has a colour alpha or a position w other than 0 (what K1 writes,
    // SURVEY 9.4-F): the fourth channel's sums and the fourth difference are +0 whatever the weights are -- the same values
    // without the instructions (a texel that has one raises fl_w4 when it is stored into the ring, before its first use).
    auto row = [&](auto w4_tag, int yr, int py, uint32_t& out_codes, uint32_t& out_idx, bool& out_sure) {
        constexpr bool W4 = decltype(w4_tag)::value;
        const int b0 = (yr % NR) * RW + lx, b1 = ((yr + R) % NR) * RW + lx, b2 = ((yr + 2 * R) % NR) * RW + lx;   // column of tap tx = -1
        uint2 sq; float4 sp = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (PASS0) sq = make_uint2(lc[b1 + R], 0u); else { sq = lq[b1 + R]; sp = lp[b1 + R]; }
        const uint32_t scc = PASS0 ? 0u : __builtin_amdgcn_udot4(sq.x, sq.x, 0u, false), snn = PASS0 ? 0u : __builtin_amdgcn_udot4(sq.y, sq.y, 0u, false);
        constexpr float kcen = SHIPPED ? kGauss2 : kGauss0;
        float a0 = (float)(sq.x & 0xFFu) * kcen, a1 = (float)((sq.x >> 8) & 0xFFu) * kcen, a2 = (float)((sq.x >> 16) & 0xFFu) * kcen, a3 = W4 ? (float)(sq.x >> 24) * kcen : 0.0f;
        float total = kcen;
#pragma unroll
        for (int i = 0; i < ntaps; i++) {
            int tx, ty;
            if (SHIPPED) { tx = i == 0 ? -1 : (i == 1 ? 1 : 0); ty = i == 2 ? 0 : -1; }
            else { tx = i % 3 - 1; ty = i / 3 - 1; }
            if (tx == 0 && ty == 0) continue;                // the centre tap: every distance is 0, its weight is the kernel's (above)
            const int ci = (ty < 0 ? b0 : (ty == 0 ? b1 : b2)) + (tx + 1) * R;
            if (PASS0) {                                     // every edge-stopping weight is exactly 1: the tap's weight is the kernel's
                const uint32_t oc = lc[ci];
                const float kk = SHIPPED ? (i == 1 ? kGauss0 : kGauss2) : (r2_of(tx, ty) == 1 ? kGauss1 : kGauss2);
                a0 = __builtin_fmaf((float)(oc & 0xFFu), kk, a0);
                a1 = __builtin_fmaf((float)((oc >> 8) & 0xFFu), kk, a1);
                a2 = __builtin_fmaf((float)((oc >> 16) & 0xFFu), kk, a2);
                if (W4) a3 = __builtin_fmaf((float)(oc >> 24), kk, a3);
                continue;
            }
            const uint2 oq = lq[ci];
            const float4 op = lp[ci];
            const int dc = mad24_minus2(__builtin_amdgcn_udot4(sq.x, oq.x, 0u, false), __builtin_amdgcn_udot4(oq.x, oq.x, scc, false));
            const int dn = mad24_minus2(__builtin_amdgcn_udot4(sq.y, oq.y, 0u, false), __builtin_amdgcn_udot4(oq.y, oq.y, snn, false));
            const float dx = sp.x - op.x, dy = sp.y - op.y, dz = sp.z - op.z;
            float dp = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            if (W4) { const float dw = sp.w - op.w; dp = __builtin_fmaf(dw, dw, dp); }
            // the kernel weight rides in the exponent: w * kern = exp2(-(e + -log2 kern))
            const float lk = SHIPPED ? shipped_lk(i) : (r2_of(tx, ty) == 1 ? 0.18033688011112042f : 0.36067376022224085f);
            const float e = __builtin_fmaf(dp, kp, __builtin_fmaf((float)dc, kc, __builtin_fmaf((float)dn, kn, lk)));
            const float wk = __builtin_amdgcn_exp2f(-e);
            a0 = __builtin_fmaf((float)(oq.x & 0xFFu), wk, a0);
            a1 = __builtin_fmaf((float)((oq.x >> 8) & 0xFFu), wk, a1);
            a2 = __builtin_fmaf((float)((oq.x >> 16) & 0xFFu), wk, a2);
            if (W4) a3 = __builtin_fmaf((float)(oq.x >> 24), wk, a3);
            total += wk;
        }
        // (PASS0: the weights' sum is a constant; its reciprocal rounded once from double)
        const float r = PASS0 ? (SHIPPED ? (float)(1.0 / (2.0 * 0.7788007830714049 + 1.0)) : (float)(1.0 / (1.0 + 4.0 * 0.8824969025845955 + 4.0 * 0.7788007830714049))) : __builtin_amdgcn_rcpf(total);
        // a weighted mean of codes, + 0.5: its floor is the output, its fraction says how far the nearest rounding boundary is
        const float y0f = __builtin_fmaf(a0, r, 0.5f), y1f = __builtin_fmaf(a1, r, 0.5f), y2f = __builtin_fmaf(a2, r, 0.5f);
        const float f0 = __builtin_amdgcn_fractf(y0f), f1 = __builtin_amdgcn_fractf(y1f), f2 = __builtin_amdgcn_fractf(y2f);
        // (the truncation is the floor -- the means are positive -- and cannot pass 255: a mean of codes with positive weights is at most
        // 255 (1 + 20 eps), + 0.5; a NaN converts to 0 and is redone anyway)
        const uint32_t o0 = (uint32_t)y0f, o1 = (uint32_t)y1f, o2 = (uint32_t)y2f;
        // sure <=> every channel's fraction lies further than the guard from 0 and from 1 (a NaN mean compares false)
        bool sure = !FLAG || (__builtin_fabsf(f0 - 0.5f) < half_guard && __builtin_fabsf(f1 - 0.5f) < half_guard && __builtin_fabsf(f2 - 0.5f) < half_guard);
        uint32_t o3 = 0u;                                        // (W4 = false: the mean of zeros is 0, half a code from either boundary)
        if (W4) {
            const float y3f = __builtin_fmaf(a3, r, 0.5f), f3 = __builtin_amdgcn_fractf(y3f);
            o3 = (uint32_t)y3f;
            if (FLAG) sure = sure && __builtin_fabsf(f3 - 0.5f) < half_guard;
        }
        out_codes = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24); out_idx = (uint32_t)py * (uint32_t)P.W + (uint32_t)px; out_sure = sure;
    };
    for (int g = 0; g < groups; g++) {
        const bool more = g + 1 < groups;
#ifdef VRT_VER_PRIO
        // (a wave's priority falls as it gets on: see k_denoise_pair)
        if (4 * g < groups) __builtin_amdgcn_s_setprio(3); else if (2 * g < groups) __builtin_amdgcn_s_setprio(2);
        else if (4 * g < 3 * groups) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
        if (more) fetch(g + U);                                  // the unit group g + 1 adds: in flight while group g is filtered
        const int yr = 4 * g + wave, py = ys + yr;               // relative row of the output; its taps' rows are yr, yr + R, yr + 2R in ring terms
        const bool have = py < ye && px < P.W;
        uint32_t out_codes = 0u, out_idx = 0u;
        bool out_sure = true;
        const bool w4 = fl_w4 != 0u;                             // (uniform: read after the barrier that follows every store into the ring)
        if (have) {
            if (w4) row(std::true_type{}, yr, py, out_codes, out_idx, out_sure);
            else    row(std::false_type{}, yr, py, out_codes, out_idx, out_sure);
        }
        // (the ring first, the output after it: the wait for the fetched unit would otherwise also wait for this group's store
        // -- on gfx950 one counter covers both -- once per group, with nothing left to hide it)
        if (more) stash(g + U);                                  // into the slots of the unit group g no longer reads
        if (have) {
            if (out_sure) reinterpret_cast<uint32_t*>(P.color_out)[out_idx] = out_codes;
            else if (FLAG) { const uint32_t slot = atomicAdd(&fl_n, 1u); if (slot < VRT_DEN_FIXCAP) fl_px[slot] = out_idx; }
        }
        __syncthreads();
    }
#ifdef VRT_K3_STAMPS
    k3_t[2] = (uint32_t)wall_clock64();
#endif
    if (FLAG) {
        // (the loop's last barrier is behind us: fl_n and fl_px are final)
        const uint32_t n = fl_n;
#ifdef VRT_K3_STAMPS
        if (n == 0u) { __syncthreads(); k3_stamp(); return; }
#endif
        if (n == 0u) return;                                     // uniform per workgroup
        if (P.fix_counts && threadIdx.x == 0) atomicAdd(&P.fix_counts[(blockIdx.y * gridDim.x + blockIdx.x) & (VRT_DENOISE_SEGS - 1u)], n);
        denoise_redo<SHIPPED, PASS0>(P, R, n, false, x0, 64, ys, ye, fl_px, fx_w, fx_c);
    }
#ifdef VRT_K3_STAMPS
    __syncthreads(); k3_stamp();
#endif
}

// ---- the verified pass, every weight computed once (round 4) ---------------------------------------------------------------
// The weight of a tap is symmetric: w(p, q) = w(q, p) -- integer code distances, squares of differences, the same kernel weight
// for d and -d -- bit for bit in the arithmetic above.  Of the eight taps of a pixel p the four "forward" ones (1, 0), (-1, 1),
// (0, 1), (1, 1) are computed by p's lane; the four "backward" ones are forward weights of the pixels R to the left / R rows
// above.  A workgroup is R waves and a group of rows is R rows, one per wave, so that the row R above a wave's row is the row
// the SAME wave did one group earlier: its three downward weights are still in registers (the one straight above stays in its
// lane, the diagonal ones come through the crossbar, ds_bpermute), and the weight of the tap to the left is this row's own
// (1, 0) of the lane R to the left.  Four exponentials per pixel instead of eight, no weight ever in memory; in exchange R rows
// above every segment only compute downward weights and a wave's 64 lanes are 64 columns of which the inner 64 - 2 R produce
// output.  The ring holds a texel as position x, y, z + the biased normal codes (16 B) and the four colour codes as halves
// (8 B): exact in fp16, so the colour distance is four v_dot2_f32_f16 (integers below 2^24: exact in fp32 in any order) and a
// tap's colour goes into the sums by v_fma_mix_f32 without a conversion.  The sums are taken in the order of k_denoise_ver
// (centre, then taps 0 .. 8): the output is that kernel's bit for bit, redone pixels and all.  The position's w has a plane of its
// own in the ring, read only once a texel with a w other than +-0 or a colour alpha has been met (K1 writes neither).
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
#ifndef VRT_PAIR_FIXCAP
#define VRT_PAIR_FIXCAP 256     // listed pixels of a workgroup (of <= 58 x ~30): a few are listed, 2 % of a hostile frame; more: all of them redone
#endif
template <bool FLAG, int R>
__global__ __launch_bounds__(64 * R) void k_denoise_pair(const DenoiseParams P, int seg_rows, int segs_per_strip)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds_g[];
    __shared__ uint32_t fl_n, fl_w4;
    __shared__ uint32_t fl_px[VRT_PAIR_FIXCAP];
    __shared__ float fx_w[(64 * R) / 9][9];
    __shared__ uint32_t fx_c[(64 * R) / 9][9];
#ifdef VRT_K3_STAMPS
    uint32_t k3_t[4] = {(uint32_t)wall_clock64(), 0u, 0u, 0u};
    uint32_t k3_redo = 0u;                                // (time inside the in-loop redo rounds; the stamp "ring filled" is moved back by it)
    auto k3_stamp = [&]() {
        if (threadIdx.x == 0) {
            uint32_t* o = reinterpret_cast<uint32_t*>(P.color_out) + 4u * (blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = k3_t[0]; o[1] = k3_t[1] + k3_redo; o[2] = k3_t[2]; o[3] = (uint32_t)wall_clock64();
        }
    };
#endif
    constexpr int OW = 64 - 2 * R;                        // output columns of a strip
    constexpr int UT = R * 64;                            // texels of a unit of R rows
    constexpr int NT = 4 * UT;                            // ring: four units (three that a group reads + the one on its way in)
    float4* lp = lds_g;                                   // x, y, z, biased normal codes
    uint2* lh = reinterpret_cast<uint2*>(lds_g + NT);     // colour codes as four halves
    float* lw = reinterpret_cast<float*>(lh + NT);        // position w (read by the general form of a row only)
    const int x0 = blockIdx.x * OW;                       // first output column; lane l is column x0 - R + l
    int ys, ye;
    {
        const int k = (int)blockIdx.y / segs_per_strip, j = (int)blockIdx.y - k * segs_per_strip;
        const int g = k * P.sh.nranks + P.sh.rank;
        int r0 = g * P.sh.strip_rows - P.extend, r1 = (g + 1) * P.sh.strip_rows;
        r1 = (r1 < P.H ? r1 : P.H) + P.extend;
        r0 = r0 < 0 ? 0 : r0; r1 = r1 < P.H ? r1 : P.H;
        ys = r0 + j * seg_rows;
        ye = ys + seg_rows < r1 ? ys + seg_rows : r1;
    }
    if (ys >= ye) return;                                 // uniform per workgroup
    const int nrows = ye - ys;
    const int groups = (nrows + 2 * R - 1) / R;           // centre rows f = 0 .. nrows + R - 1 in ring terms (ring row 0 = frame row ys - R)
    if (threadIdx.x == 0) { fl_n = 0u; fl_w4 = 0u; }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l = threadIdx.x & 63, px = x0 - R + l;
    // a lane's texel, the texel R to its left, the texel R to its right within this wave's row of a unit (the ends are never used)
    const int tl = wave * 64 + l, tL = wave * 64 + (l - R < 0 ? 0 : l - R), tR = wave * 64 + (l + R > 63 ? 63 : l + R);
    const int bL = (l - R < 0 ? 0 : l - R) << 2, bR = (l + R > 63 ? 63 : l + R) << 2;      // the same lanes for ds_bpermute
    // a thread's texel of a unit: row `wave` of the unit, its own column (clamped to the frame: the ring holds copies of the border)
    const uint32_t xA = (uint32_t)(px < 0 ? 0 : (px > P.W - 1 ? P.W - 1 : px));
    const uint32_t* const gc = reinterpret_cast<const uint32_t*>(P.color_in);
    const uint32_t* const gn = reinterpret_cast<const uint32_t*>(P.normal);
    const float4* const gp = reinterpret_cast<const float4*>(P.position);
    float4 pA = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t colA = 0u, nrmA = 0u;
    auto fetch = [&](int unit) {                           // (the row is the wave's: a scalar base and the lane's column)
        int yA = ys - R + R * unit + wave;
        yA = yA < 0 ? 0 : (yA > P.H - 1 ? P.H - 1 : yA);
        const size_t ro = (size_t)yA * (size_t)P.W;
        pA = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(gp + ro) + xA * 16u);
        colA = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(gc + ro) + xA * 4u);
        nrmA = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(gn + ro) + xA * 4u);
    };
    auto bias = [](uint32_t n) {                          // (k_denoise_ver's: SNORM -128 reads as -127, then every byte + 128)
        uint32_t z = n ^ 0x80808080u;
        z = ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu);
        return (n | (z >> 7)) ^ 0x80808080u;
    };
    auto stash = [&](int slot) {
        const int i = slot * UT + tl;
        // (a colour alpha or a position w other than +-0: the rows take their general form from the next barrier on)
        if ((colA >> 24) != 0u || (__float_as_uint(pA.w) << 1) != 0u) fl_w4 = 1u;
        // codes as halves: 1024 + c is 0x6400 | c in fp16, exactly; minus 1024
        const v2h k1024 = {(_Float16)1024.0f, (_Float16)1024.0f};
        const v2h c01 = __builtin_bit_cast(v2h, __builtin_amdgcn_perm(colA, 0x64646464u, 0x00050004u)) - k1024;
        const v2h c23 = __builtin_bit_cast(v2h, __builtin_amdgcn_perm(colA, 0x64646464u, 0x00070006u)) - k1024;
        lp[i] = make_float4(pA.x, pA.y, pA.z, __uint_as_float(bias(nrmA)));
        lh[i] = make_uint2(__builtin_bit_cast(uint32_t, c01), __builtin_bit_cast(uint32_t, c23));
#ifndef VRT_PAIR_NOLW
        lw[i] = pA.w;
#endif
    };
    {
        fetch(0);                                          // (both requested before the first is stored: one round trip)
        const float4 qp = pA; const uint32_t qc = colA, qn = nrmA;
        fetch(1);
        const float4 rp = pA; const uint32_t rcol = colA, rn = nrmA;
        pA = qp; colA = qc; nrmA = qn; stash(0);
        pA = rp; colA = rcol; nrmA = rn; stash(1);
    }
    __syncthreads();
#ifdef VRT_K3_STAMPS
    k3_t[1] = (uint32_t)wall_clock64();
#endif
    const float kc = P.vkc, kn = P.vkn, kp = P.vkp;
    const float half_guard = 0.5f - P.guard;
    float pf1 = 0.0f, pf2 = 0.0f, pf3 = 0.0f;             // the downward weights (-1, 1), (0, 1), (1, 1) of this wave's row of the group before
    // One centre row of a wave -- in ring slot S, the row below it in slot S + 1, the row above in slot S - 1 --: the forward
    // weights of its 64 texels, and (`outrow`: it is a row of the segment) the pixel.
    auto row = [&](auto w4_tag, auto s_tag, bool outrow, int py, uint32_t& out_codes, uint32_t& out_idx, bool& out_sure) {
        constexpr bool W4 = decltype(w4_tag)::value;
        constexpr int rc = decltype(s_tag)::value * UT, rf = ((decltype(s_tag)::value + 1) & 3) * UT, rb = ((decltype(s_tag)::value + 3) & 3) * UT;
        const float4 s4 = lp[rc + tl];
        const uint2 sh = lh[rc + tl];
        const v2h s01 = __builtin_bit_cast(v2h, sh.x), s23 = __builtin_bit_cast(v2h, sh.y);
        const v2h m01 = s01 * (_Float16)(-2.0f), m23 = s23 * (_Float16)(-2.0f);            // -2 s: |s - o|^2 = |s|^2 + |o|^2 + (-2 s) . o
        const uint32_t sn = __float_as_uint(s4.w);
        const float sw = W4 ? lw[rc + tl] : 0.0f;
        const float scc = __builtin_amdgcn_fdot2(s01, s01, __builtin_amdgcn_fdot2(s23, s23, 0.0f, false), false);
        const uint32_t snn = __builtin_amdgcn_udot4(sn, sn, 0u, false);
        float wf[4]; uint2 hf[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int ci = k == 0 ? rc + tR : (k == 1 ? rf + tL : (k == 2 ? rf + tl : rf + tR));
            const float4 o4 = lp[ci];
            const uint2 oh = lh[ci];
            hf[k] = oh;
            const v2h o01 = __builtin_bit_cast(v2h, oh.x), o23 = __builtin_bit_cast(v2h, oh.y);
            const uint32_t on = __float_as_uint(o4.w);
            // integers below 2^24 at every step: exact
            const float dc = __builtin_amdgcn_fdot2(m23, o23, __builtin_amdgcn_fdot2(m01, o01, __builtin_amdgcn_fdot2(o23, o23, __builtin_amdgcn_fdot2(o01, o01, scc, false), false), false), false);
            const int dn = mad24_minus2(__builtin_amdgcn_udot4(sn, on, 0u, false), __builtin_amdgcn_udot4(on, on, snn, false));
            const float dx = s4.x - o4.x, dy = s4.y - o4.y, dz = s4.z - o4.z;
            float dp = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            if (W4) { const float dw = sw - lw[ci]; dp = __builtin_fmaf(dw, dw, dp); }
            const float lk = (k == 0 || k == 2) ? 0.18033688011112042f : 0.36067376022224085f;      // -log2(G1), -log2(G2)
            const float e = __builtin_fmaf(dp, kp, __builtin_fmaf(dc, kc, __builtin_fmaf((float)dn, kn, lk)));
            wf[k] = __builtin_amdgcn_exp2f(-e);
        }
        // the backward taps: (-1, -1) is the (1, 1) of the texel R left and R up, (0, -1) the (0, 1) of the texel R up, (1, -1) the
        // (-1, 1) of the texel R right and R up, (-1, 0) the (1, 0) of the texel R to the left
        const float w0 = __int_as_float(__builtin_amdgcn_ds_bpermute(bL, __float_as_int(pf3)));
        const float w1 = pf2;
        const float w2 = __int_as_float(__builtin_amdgcn_ds_bpermute(bR, __float_as_int(pf1)));
        const float w3 = __int_as_float(__builtin_amdgcn_ds_bpermute(bL, __float_as_int(wf[0])));
        pf1 = wf[1]; pf2 = wf[2]; pf3 = wf[3];
        if (!outrow) return;
        const uint2 h0 = lh[rb + tL], h1 = lh[rb + tl], h2 = lh[rb + tR], h3 = lh[rc + tL];
        constexpr float kcen = kGauss0;
        float a0 = (float)s01.x * kcen, a1 = (float)s01.y * kcen, a2 = (float)s23.x * kcen, a3 = W4 ? (float)s23.y * kcen : 0.0f;
        float total = kcen;
        auto acc = [&](const uint2& h, float wk) {
            const v2h c01 = __builtin_bit_cast(v2h, h.x), c23 = __builtin_bit_cast(v2h, h.y);
            a0 = __builtin_fmaf((float)c01.x, wk, a0);
            a1 = __builtin_fmaf((float)c01.y, wk, a1);
            a2 = __builtin_fmaf((float)c23.x, wk, a2);
            if (W4) a3 = __builtin_fmaf((float)c23.y, wk, a3);
            total += wk;
        };
        acc(h0, w0); acc(h1, w1); acc(h2, w2); acc(h3, w3);              // taps 0 .. 3
        acc(hf[0], wf[0]); acc(hf[1], wf[1]); acc(hf[2], wf[2]); acc(hf[3], wf[3]);   // taps 5 .. 8
        const float r = __builtin_amdgcn_rcpf(total);
        const float y0f = __builtin_fmaf(a0, r, 0.5f), y1f = __builtin_fmaf(a1, r, 0.5f), y2f = __builtin_fmaf(a2, r, 0.5f);
        const float f0 = __builtin_amdgcn_fractf(y0f), f1 = __builtin_amdgcn_fractf(y1f), f2 = __builtin_amdgcn_fractf(y2f);
        const uint32_t o0 = (uint32_t)y0f, o1 = (uint32_t)y1f, o2 = (uint32_t)y2f;
        bool sure = !FLAG || (__builtin_fabsf(f0 - 0.5f) < half_guard && __builtin_fabsf(f1 - 0.5f) < half_guard && __builtin_fabsf(f2 - 0.5f) < half_guard);
        uint32_t o3 = 0u;
        if (W4) {
            const float y3f = __builtin_fmaf(a3, r, 0.5f), f3 = __builtin_amdgcn_fractf(y3f);
            o3 = (uint32_t)y3f;
            if (FLAG) sure = sure && __builtin_fabsf(f3 - 0.5f) < half_guard;
        }
        out_codes = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24); out_idx = (uint32_t)py * (uint32_t)P.W + (uint32_t)px; out_sure = sure;
    };
    // Group g: the centre rows R g .. R g + R - 1, one per wave, in ring slot g & 3.  They read units g - 1, g, g + 1; unit g + 2
    // arrives meanwhile and goes into the slot of unit g - 2.  (Four groups per turn of the loop: the slots are constants and every
    // LDS address is a lane's base + an immediate.)
    auto step = [&](auto s_tag, int g) {
        constexpr int S = decltype(s_tag)::value;
        const bool more = g + 1 < groups;
#ifndef VRT_PAIR_NOPRIO
        // The scheduler serves the oldest wave first: of the workgroups of a compute unit the youngest would be left to finish alone,
        // one wave per SIMD.  A wave's priority falls as it gets on, so that whoever is behind goes first and all end together.
        {
            if (4 * g < groups) __builtin_amdgcn_s_setprio(3); else if (2 * g < groups) __builtin_amdgcn_s_setprio(2);
            else if (4 * g < 3 * groups) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
#endif
        if (more) fetch(g + 2);
        const int f = R * g + wave, py = ys + f - R;
        const bool outrow = g >= 1 && py < ye;
        const bool have = outrow && l >= R && l < 64 - R && px < P.W;
        uint32_t out_codes = 0u, out_idx = 0u;
        bool out_sure = true;
        const bool w4 = fl_w4 != 0u;
        if (f < nrows + R) {                                         // (uniform per wave; every lane computes its texel's weights)
            if (w4) row(std::true_type{}, s_tag, outrow, py, out_codes, out_idx, out_sure);
            else    row(std::false_type{}, s_tag, outrow, py, out_codes, out_idx, out_sure);
        }
        if (more) stash((S + 2) & 3);
        if (have) {
            if (out_sure) reinterpret_cast<uint32_t*>(P.color_out)[out_idx] = out_codes;
            else if (FLAG) { const uint32_t slot = atomicAdd(&fl_n, 1u); if (slot < VRT_PAIR_FIXCAP) fl_px[slot] = out_idx; }
        }
        __syncthreads();
    };
    uint32_t done = 0u;                                               // listed pixels already evaluated the shader's own way
    for (int g = 0; g < groups; g += 4) {
        if (FLAG && g > 0) {
            // The pixels listed so far, now -- under the other workgroups' rows -- rather than all at the end of the launch, where every
            // workgroup would stand in this chain of dependent instructions at once with nothing to hide it.  (The barrier: nobody
            // lists a pixel of the next group before everybody has read the count.)
            const uint32_t n = fl_n < VRT_PAIR_FIXCAP ? fl_n : VRT_PAIR_FIXCAP;
            __syncthreads();
#ifdef VRT_K3_STAMPS
            const uint32_t k3_r0 = (uint32_t)wall_clock64();
#endif
#ifndef VRT_PAIR_EXP_NOREDO
            if (n > done) { denoise_redo<false, false>(P, R, n, false, x0, OW, ys, ye, fl_px, fx_w, fx_c, done); done = n; }
#endif
#ifdef VRT_K3_STAMPS
            k3_redo += (uint32_t)wall_clock64() - k3_r0;
#endif
        }
        step(std::integral_constant<int, 0>{}, g);
        if (g + 1 >= groups) break;
        step(std::integral_constant<int, 1>{}, g + 1);
        if (g + 2 >= groups) break;
        step(std::integral_constant<int, 2>{}, g + 2);
        if (g + 3 >= groups) break;
        step(std::integral_constant<int, 3>{}, g + 3);
    }
#ifdef VRT_K3_STAMPS
    k3_t[2] = (uint32_t)wall_clock64();
#endif
    if (FLAG) {
        // (the loop's last barrier is behind us: fl_n and fl_px are final)
        const uint32_t n = fl_n;
        if (n > done) {                                              // uniform per workgroup
#ifndef VRT_PAIR_NOPRIO
            __builtin_amdgcn_s_setprio(3);
#endif
#ifndef VRT_PAIR_EXP_NOREDO
            denoise_redo<false, false>(P, R, n, n > VRT_PAIR_FIXCAP, x0, OW, ys, ye, fl_px, fx_w, fx_c, done);
#endif
        }
        if (n != 0u && P.fix_counts && threadIdx.x == 0) atomicAdd(&P.fix_counts[(blockIdx.y * gridDim.x + blockIdx.x) & (VRT_DENOISE_SEGS - 1u)], n);
    }
#ifdef VRT_K3_STAMPS
    __syncthreads(); k3_stamp();
#endif
}

// ---- pass 0, a wave to itself (round 4) --------------------------------------------------------------------------------------
// phi = +inf: every edge-stopping weight is exactly 1 and the pass is a 3 x 3 blur of the colour plane with the tap offset 1.  A
// wave owns 62 output columns (its 64 lanes are the columns x0 - 1 .. x0 + 62) and SEG rows: it requests the SEG + 2 rows of its
// column once, all before the first is used, and walks down them with the three rows it needs as floats in registers -- a lane's
// left and right neighbours come through the data-parallel shifts (wave_shr / wave_shl), no LDS, no barrier, nothing shared with
// another wave.  The kernel is separable -- (g, 1, g) x (g, 1, g) with g = G1 = exp(-1/8), G2 = G1^2 --, so a row's horizontal sums
// fma(l, g, fma(r, g, c)) are taken once, when the row arrives, and an output row is fma(h_up, g, fma(h_down, g, h)): four fused
// multiply-adds per channel instead of eight.  That is another cheap form than k_denoise_ver<.., PASS0>'s, under the same guard
// (vrt_denoise_bound.h, denoise_guard_pass0, derives both); the pixels within the guard -- a dozen to a hundred per frame -- go
// through denoise_redo at the end.
#ifndef VRT_P0_SEG
#define VRT_P0_SEG 8
#endif
template <bool FLAG>
__global__ __launch_bounds__(64) void k_denoise_p0(const DenoiseParams P, int segs_per_strip)
{
    __shared__ uint32_t fl_px[64];
    __shared__ float fx_w[7][9];
    __shared__ uint32_t fx_c[7][9];
    constexpr int OW = 62, SEG = VRT_P0_SEG;
    const int x0 = blockIdx.x * OW;
    int ys, ye;
    {
        const int k = (int)blockIdx.y / segs_per_strip, j = (int)blockIdx.y - k * segs_per_strip;
        const int g = k * P.sh.nranks + P.sh.rank;
        int r0 = g * P.sh.strip_rows - P.extend, r1 = (g + 1) * P.sh.strip_rows;
        r1 = (r1 < P.H ? r1 : P.H) + P.extend;
        r0 = r0 < 0 ? 0 : r0; r1 = r1 < P.H ? r1 : P.H;
        ys = r0 + j * SEG;
        ye = ys + SEG < r1 ? ys + SEG : r1;
    }
    if (ys >= ye) return;                                 // uniform
    const int l = threadIdx.x, px = x0 - 1 + l;
    const uint32_t xA = (uint32_t)(px < 0 ? 0 : (px > P.W - 1 ? P.W - 1 : px));
    const uint32_t* const gc = reinterpret_cast<const uint32_t*>(P.color_in);
    uint32_t code[SEG + 2];
#pragma unroll
    for (int q = 0; q < SEG + 2; q++) {                   // rows ys - 1 .. ys + SEG, clamped to the frame (rows beyond ye + 1 are never used)
        int y = ys - 1 + q;
        y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);
        code[q] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(gc + (size_t)y * (size_t)P.W) + xA * 4u);
    }
    const float half_guard = 0.5f - P.guard;
    constexpr float rsum = (float)(1.0 / (1.0 + 4.0 * 0.8824969025845955 + 4.0 * 0.7788007830714049));
    // a row as its horizontal sums, three (four) channels
    struct Row { float h[4]; };
    bool w4 = false;                                      // (uniform) a colour alpha has been met: the fourth channel's sums from here on
    auto load_row = [&](uint32_t cc, Row& o) {
        const uint32_t cl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cc, 0x138, 0xf, 0xf, false);     // wave_shr:1 -- lane l gets lane l - 1's
        const uint32_t cr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cc, 0x130, 0xf, 0xf, false);     // wave_shl:1 -- lane l gets lane l + 1's
        if (__ballot((cc >> 24) != 0u) != 0ull) w4 = true;
        o.h[0] = __builtin_fmaf((float)(cl & 0xFFu), kGauss1, __builtin_fmaf((float)(cr & 0xFFu), kGauss1, (float)(cc & 0xFFu)));
        o.h[1] = __builtin_fmaf((float)((cl >> 8) & 0xFFu), kGauss1, __builtin_fmaf((float)((cr >> 8) & 0xFFu), kGauss1, (float)((cc >> 8) & 0xFFu)));
        o.h[2] = __builtin_fmaf((float)((cl >> 16) & 0xFFu), kGauss1, __builtin_fmaf((float)((cr >> 16) & 0xFFu), kGauss1, (float)((cc >> 16) & 0xFFu)));
        o.h[3] = w4 ? __builtin_fmaf((float)(cl >> 24), kGauss1, __builtin_fmaf((float)(cr >> 24), kGauss1, (float)(cc >> 24))) : 0.0f;
    };
    Row rows[3];
    load_row(code[0], rows[0]);
    load_row(code[1], rows[1]);
    uint32_t n = 0u;                                      // (uniform) pixels listed so far
    const bool col_ok = l >= 1 && l <= OW && px < P.W;
#pragma unroll
    for (int q = 0; q < SEG; q++) {
        load_row(code[q + 2], rows[(q + 2) % 3]);
        const int py = ys + q;
        if (py < ye) {                                    // uniform
            const Row& up = rows[q % 3]; const Row& me = rows[(q + 1) % 3]; const Row& dn = rows[(q + 2) % 3];
            float a[4];
#pragma unroll
            for (int ch = 0; ch < 4; ch++) a[ch] = (ch == 3 && !w4) ? 0.0f : __builtin_fmaf(up.h[ch], kGauss1, __builtin_fmaf(dn.h[ch], kGauss1, me.h[ch]));
            const float y0f = __builtin_fmaf(a[0], rsum, 0.5f), y1f = __builtin_fmaf(a[1], rsum, 0.5f), y2f = __builtin_fmaf(a[2], rsum, 0.5f);
            const float f0 = __builtin_amdgcn_fractf(y0f), f1 = __builtin_amdgcn_fractf(y1f), f2 = __builtin_amdgcn_fractf(y2f);
            const uint32_t o0 = (uint32_t)y0f, o1 = (uint32_t)y1f, o2 = (uint32_t)y2f;
            bool sure = !FLAG || (__builtin_fabsf(f0 - 0.5f) < half_guard && __builtin_fabsf(f1 - 0.5f) < half_guard && __builtin_fabsf(f2 - 0.5f) < half_guard);
            uint32_t o3 = 0u;
            if (w4) {
                const float y3f = __builtin_fmaf(a[3], rsum, 0.5f), f3 = __builtin_amdgcn_fractf(y3f);
                o3 = (uint32_t)y3f;
                if (FLAG) sure = sure && __builtin_fabsf(f3 - 0.5f) < half_guard;
            }
            const uint32_t idx = (uint32_t)py * (uint32_t)P.W + (uint32_t)px;
            if (col_ok) {
                if (sure) reinterpret_cast<uint32_t*>(P.color_out)[idx] = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
            }
            if (FLAG) {
                const uint64_t m = __ballot(col_ok && !sure);
                if (m != 0ull) {                              // (uniform; rare) the listed pixels: the wave's own count, a lane's rank among the listers
                    if (col_ok && !sure) {
                        const uint32_t slot = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (slot < 64u) fl_px[slot] = idx;
                    }
                    n += (uint32_t)__builtin_popcountll(m);
                }
            }
        }
    }
    if (FLAG && n != 0u) {
        __syncthreads();
        if (P.fix_counts && threadIdx.x == 0) atomicAdd(&P.fix_counts[(blockIdx.y * gridDim.x + blockIdx.x) & (VRT_DENOISE_SEGS - 1u)], n);
        denoise_redo<false, true>(P, 1, n, n > 64u, x0, OW, ys, ye, fl_px, fx_w, fx_c);
    }
}

hipError_t launch_denoise_pass(const DenoiseParams& p, hipStream_t s)
{
    int per = p.sh.strip_rows + 2 * p.extend;
    int rows = p.sh.n_local_strips * per;
    dim3 grid((unsigned)((p.W + 63) / 64), (unsigned)((rows + 3) / 4)), block(256);
    // phi = +inf in all three channels <=> pass 0 (denoiser_stage.cpp:148-150)
    bool inf = __builtin_isinf(p.phi_color) && __builtin_isinf(p.phi_normal) && __builtin_isinf(p.phi_pos);
    // LDS tiling needs an integral tap offset, a halo that fits (<= 5 px: 74 x 14 px x 48 B = 48.6 KiB) and blocks of
    // 4 consecutive frame rows (single strip, or strips whose extended height is a multiple of 4)
    float sw = p.step_width;
    int R = (int)sw;
    bool tiled = (float)R == sw && R >= 1 && R <= 5 && (p.sh.nranks == 1 || per % 4 == 0);
    const bool shipped = (p.mode & 1) == VRT_DENOISE_AS_SHIPPED;
    if ((float)R == sw && R == 1 && inf && p.verified && !shipped && !p.no_p0) {
        // pass 0 of the canonical taps, a wave to itself (k_denoise_p0): strips of 62 output columns x VRT_P0_SEG rows.  (VRT_DENOISE_FAST too:
        // its pass 0 has always been exact, and this is 8 us against the literal kernel's 11.7.)
        const int strips = (p.W + 61) / 62;
        const int strip_ext = p.sh.nranks == 1 ? p.H : per;
        const int segs = (strip_ext + VRT_P0_SEG - 1) / VRT_P0_SEG;
        dim3 g2((unsigned)strips, (unsigned)(segs * p.sh.n_local_strips));
        hipLaunchKernelGGL((k_denoise_p0<true>), g2, dim3(64), 0, s, p, segs);
    }
    else
    // (a rank's 16-row strips are too short for it -- R rows above every segment only compute weights --: two passes over rank 0's strips of
    // 2 / 4 / 8 ranks 29.8 / 18.5 / 14.2 us against k_denoise_ver's 27.0 / 19.5 / 13.9, tools/exp_r4_k3_shard.py; bands of 64 rows and more take it)
    if ((float)R == sw && R >= 2 && R <= 5 && p.verified && !inf && !shipped && !p.no_pair && (p.sh.nranks == 1 || per >= 64)) {
        // the verified pass with every weight computed once (k_denoise_pair): strips of 64 - 2 R output columns x seg_rows rows, R waves
        // per workgroup, as many waves in the launch as k_denoise_ver's (p.pair_wgs > 0: that many workgroups instead)
        const int ow = 64 - 2 * R;
        const int strips = (p.W + ow - 1) / ow;
        const int strip_ext = p.sh.nranks == 1 ? p.H : per;
        const int total = p.sh.nranks == 1 ? p.H : rows;
        // (as many waves as k_denoise_ver's launch, and no segment much longer than 48 rows: 4K measured 105 against 109 us for two passes)
        int wgs = (p.pair_wgs > 0 ? p.pair_wgs : 4096 / R) / strips; if (wgs < 1) wgs = 1;
        if (p.pair_wgs <= 0 && wgs < (total + 47) / 48) wgs = (total + 47) / 48;
        int seg_rows = ((total + wgs - 1) / wgs + R - 1) / R * R; if (seg_rows < 2 * R) seg_rows = 2 * R;
        const int segs = (strip_ext + seg_rows - 1) / seg_rows;
        dim3 g2((unsigned)strips, (unsigned)(segs * p.sh.n_local_strips));
#ifdef VRT_PAIR_NOLW
        const size_t l2 = (size_t)(4 * R) * 64 * 24;
#else
        const size_t l2 = (size_t)(4 * R) * 64 * 28;
#endif
        const bool flag = !(p.mode & VRT_DENOISE_FAST);
#define VRT_LAUNCH_PAIR(R_)                                                                                                       \
        if (flag) hipLaunchKernelGGL((k_denoise_pair<true, R_>), g2, dim3(64 * R_), l2, s, p, seg_rows, segs);                     \
        else      hipLaunchKernelGGL((k_denoise_pair<false, R_>), g2, dim3(64 * R_), l2, s, p, seg_rows, segs);
        switch (R) {
        case 2:  VRT_LAUNCH_PAIR(2) break;
        case 3:  VRT_LAUNCH_PAIR(3) break;
        case 4:  VRT_LAUNCH_PAIR(4) break;
        default: VRT_LAUNCH_PAIR(5) break;
        }
#undef VRT_LAUNCH_PAIR
    }
    else if ((float)R == sw && R >= 1 && R <= 5 && p.verified && !(inf && (p.mode & VRT_DENOISE_FAST))) {
        // the verified pass (exact output) or, with VRT_DENOISE_FAST, its cheap half alone: 64-pixel column strips x seg_rows rows of
        // the rank's strips (with the rows either side this pass must also produce), about four workgroups per compute unit
        // (768 ... 2048 measured the same); pass 0: the ring holds the colour plane only -- 34 VGPRs, 9 KB of LDS: eight
        const int strips = (p.W + 63) / 64;
        const int strip_ext = p.sh.nranks == 1 ? p.H : per;                 // rows of a local strip with its extension
        const int total = p.sh.nranks == 1 ? p.H : rows;
        int wgs = (inf ? 2048 : 1024) / strips; if (wgs < 1) wgs = 1;
        int seg_rows = ((total + wgs - 1) / wgs + 3) & ~3; if (seg_rows < 8) seg_rows = 8;
        const int segs = (strip_ext + seg_rows - 1) / seg_rows;
        dim3 g2((unsigned)strips, (unsigned)(segs * p.sh.n_local_strips));
        const int U = (4 + 2 * R + 3) / 4;
        const size_t l2 = (size_t)(64 + 2 * R) * (size_t)(4 * (U + 1)) * (inf ? 4 : 24);
        if (inf) {
            if (R == 1) { if (shipped) hipLaunchKernelGGL((k_denoise_ver<true, true, 1, true>), g2, block, l2, s, p, R, seg_rows, segs);
                          else         hipLaunchKernelGGL((k_denoise_ver<false, true, 1, true>), g2, block, l2, s, p, R, seg_rows, segs); }
            else        { if (shipped) hipLaunchKernelGGL((k_denoise_ver<true, true, 0, true>), g2, block, l2, s, p, R, seg_rows, segs);
                          else         hipLaunchKernelGGL((k_denoise_ver<false, true, 0, true>), g2, block, l2, s, p, R, seg_rows, segs); }
        } else {
            const bool flag = !(p.mode & VRT_DENOISE_FAST);
#define VRT_LAUNCH_VER(SH_, FL_)                                                                                                  \
            switch (R) {                                                                                                          \
            case 2:  hipLaunchKernelGGL((k_denoise_ver<SH_, FL_, 2>), g2, block, l2, s, p, R, seg_rows, segs); break;                 \
            case 3:  hipLaunchKernelGGL((k_denoise_ver<SH_, FL_, 3>), g2, block, l2, s, p, R, seg_rows, segs); break;                 \
            case 5:  hipLaunchKernelGGL((k_denoise_ver<SH_, FL_, 5>), g2, block, l2, s, p, R, seg_rows, segs); break;                 \
            default: hipLaunchKernelGGL((k_denoise_ver<SH_, FL_, 0>), g2, block, l2, s, p, R, seg_rows, segs); break;                 \
            }
            if (flag) { if (shipped) { VRT_LAUNCH_VER(true, true) } else { VRT_LAUNCH_VER(false, true) } }
            else      { if (shipped) { VRT_LAUNCH_VER(true, false) } else { VRT_LAUNCH_VER(false, false) } }
#undef VRT_LAUNCH_VER
        }
    }
    else if (tiled) {
        size_t lds = (size_t)(64 + 2 * R) * (size_t)(4 + 2 * R) * (inf ? 16 : 48);   // pass 0 stages the colour plane only
        if (false) {}
        else if (!inf && (p.mode & VRT_DENOISE_FAST) && p.sh.nranks == 1 && p.extend == 0) {
            const int th = p.tile16 ? 16 : 8;                                // development switch: tile height 8 / 16
            dim3 g2((unsigned)((p.W + 63) / 64), (unsigned)((p.H + th - 1) / th));
            const size_t l2 = (size_t)(64 + 2 * R) * (size_t)(th + 2 * R) * 48;
            if (th == 8) { if (shipped) hipLaunchKernelGGL((k_denoise_fast<true, 8>), g2, block, l2, s, p, R);
                           else         hipLaunchKernelGGL((k_denoise_fast<false, 8>), g2, block, l2, s, p, R); }
            else         { if (shipped) hipLaunchKernelGGL((k_denoise_fast<true, 16>), g2, block, l2, s, p, R);
                           else         hipLaunchKernelGGL((k_denoise_fast<false, 16>), g2, block, l2, s, p, R); }
        }
        else if (!inf && (p.mode & VRT_DENOISE_FAST)) {
            if (shipped) hipLaunchKernelGGL((k_denoise_lds<false, true, true>), grid, block, lds, s, p, R);
            else         hipLaunchKernelGGL((k_denoise_lds<false, false, true>), grid, block, lds, s, p, R);
        }
        else if (inf) { if (shipped) hipLaunchKernelGGL((k_denoise_lds<true, true>), grid, block, lds, s, p, R);
                   else         hipLaunchKernelGGL((k_denoise_lds<true, false>), grid, block, lds, s, p, R); }
        else if (p.packed_ok && !p.no_packed) {                              // (development switch: the tap-by-tap form)
            if (shipped) hipLaunchKernelGGL((k_denoise_lds<false, true, false, true>), grid, block, lds, s, p, R);
            else         hipLaunchKernelGGL((k_denoise_lds<false, false, false, true>), grid, block, lds, s, p, R);
        }
        else     { if (shipped) hipLaunchKernelGGL((k_denoise_lds<false, true>), grid, block, lds, s, p, R);
                   else         hipLaunchKernelGGL((k_denoise_lds<false, false>), grid, block, lds, s, p, R); }
    } else {
        if (inf) hipLaunchKernelGGL(k_denoise<true>, grid, block, 0, s, p);
        else     hipLaunchKernelGGL(k_denoise<false>, grid, block, 0, s, p);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// strip pack / unpack (multi-GPU gather + halo exchange)
// ---------------------------------------------------------------------------------------------

// One workgroup per packed row.  halo == 0: all owned rows in order.  halo > 0: the first (dir < 0) or
// last (dir > 0) `halo` rows of every owned strip.
__global__ __launch_bounds__(256) void k_rows(const RowsParams P)
{
    int r = blockIdx.x;
    int y;
    if (P.halo == 0) {
        y = strip_row(P.sh, 0, r, P.H);
    } else {
        int k = r / P.halo, j = r % P.halo;
        int g = k * P.sh.nranks + P.sh.rank;
        int beg = g * P.sh.strip_rows;
        int end = beg + P.sh.strip_rows; if (end > P.H) end = P.H;
        y = (P.dir < 0) ? beg + j : end - P.halo + j;
        if (y < beg || y >= end) y = -1;
    }
    size_t row_bytes = (size_t)P.W * (size_t)P.bpp;
    const uint8_t* src; uint8_t* dst;
    if (y < 0) {
        if (P.unpack) return;
        // rows that do not exist (partial last strip): zero-fill the packed slot
        dst = P.dst + (size_t)r * row_bytes;
        for (size_t i = threadIdx.x; i < row_bytes; i += blockDim.x) dst[i] = 0;
        return;
    }
    if (P.unpack) { src = P.src + (size_t)r * row_bytes; dst = P.dst + (size_t)y * row_bytes; }
    else          { src = P.src + (size_t)y * row_bytes; dst = P.dst + (size_t)r * row_bytes; }
    if ((row_bytes & 15) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (size_t i = threadIdx.x; i < row_bytes / 16; i += blockDim.x) d4[i] = s4[i];
    } else {
        for (size_t i = threadIdx.x; i < row_bytes; i += blockDim.x) dst[i] = src[i];
    }
}

// The same for up to VRT_ROWS_BATCH images in one launch (the frames of a batch; at the root of a gather: frames x source
// ranks, each with its own strip map): one small launch per image would cost more than the copies.
__global__ __launch_bounds__(256) void k_rows_batch(const RowsBatchParams P)
{
    const int r = blockIdx.x, img = blockIdx.y;
    const ShardMap sh = P.sh[img];
    int y;
    if (P.halo == 0) y = strip_row(sh, 0, r, P.H);
    else {                                                     // halo rows of strip k = r / halo (as k_rows)
        const int k = r / P.halo, j = r % P.halo;
        const int g = k * sh.nranks + sh.rank;
        const int beg = g * sh.strip_rows;
        int end = beg + sh.strip_rows; if (end > P.H) end = P.H;
        y = (P.dir < 0) ? beg + j : end - P.halo + j;
        if (y < beg || y >= end) y = -1;
    }
    const size_t row_bytes = (size_t)P.W * (size_t)P.bpp;
    const uint8_t* src; uint8_t* dst;
    if (y < 0) {
        if (P.unpack) return;
        dst = P.dst[img] + (size_t)r * row_bytes;                  // rows that do not exist: zero-fill the packed slot
        for (size_t i = threadIdx.x; i < row_bytes; i += blockDim.x) dst[i] = 0;
        return;
    }
    if (P.unpack) { src = P.src[img] + (size_t)r * row_bytes; dst = P.dst[img] + (size_t)y * row_bytes; }
    else          { src = P.src[img] + (size_t)y * row_bytes; dst = P.dst[img] + (size_t)r * row_bytes; }
    if ((row_bytes & 15) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* s4 = reinterpret_cast<const u32x4*>(src);
        u32x4* d4 = reinterpret_cast<u32x4*>(dst);
        const size_t n = row_bytes / 16;
        size_t i = threadIdx.x;
        // two 16-byte pieces in flight per thread where the row has them (a 1080p RGBA8 row is 480 pieces for 256 threads);
        // streamed: neither side is read again before the caches have turned over
        for (; i + blockDim.x < n; i += 2 * blockDim.x) {
            const u32x4 a = __builtin_nontemporal_load(&s4[i]), b = __builtin_nontemporal_load(&s4[i + blockDim.x]);
            __builtin_nontemporal_store(a, &d4[i]);
            __builtin_nontemporal_store(b, &d4[i + blockDim.x]);
        }
        for (; i < n; i += blockDim.x) __builtin_nontemporal_store(__builtin_nontemporal_load(&s4[i]), &d4[i]);
    } else {
        for (size_t i = threadIdx.x; i < row_bytes; i += blockDim.x) dst[i] = src[i];
    }
}

hipError_t launch_rows_batch(const RowsBatchParams& p, int rows_total, int images, hipStream_t s)
{
    if (rows_total <= 0 || images <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rows_batch, dim3((unsigned)rows_total, (unsigned)images), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_rows(const RowsParams& p, int rows_total, hipStream_t s)
{
    if (rows_total <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rows, dim3((unsigned)rows_total), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ======================================================================================================
// Presentation / temporal helpers: all three are pure streaming kernels, one
// thread per pixel, one wave = 64 consecutive pixels of a row (256 B coalesced RGBA8 accesses).
// ======================================================================================================

// blit.frag:14-22 with the BlitStage sampler (linear filter, clamp-to-edge; render_image.cpp:61-66): the source
// is centre-cropped to the target's aspect ratio and scaled.  Serves as the plain upscale of a reduced-resolution
// render (voxel_render_settings.cpp:3-13) and as the letterbox copy to a window-sized target.
__global__ __launch_bounds__(256) void k_blit(BlitParams p)
{
    const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= p.tw || py >= p.th) return;
    const float sx = (float)p.sw, sy = (float)p.sh, tx = (float)p.tw, ty = (float)p.th;
    const float scale = fminf(sx / tx, sy / ty);
    const float stx = tx * scale, sty = ty * scale;
    const float vx = ((float)px + 0.5f) / tx, vy = ((float)py + 0.5f) / ty;
    const float spx = (vx * tx) * scale + (sx - stx) / 2.0f, spy = (vy * ty) * scale + (sy - sty) / 2.0f;
    const float u = spx / sx, v = spy / sy;
    const float fx = u * sx - 0.5f, fy = v * sy - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx = fx - x0f, wy = fy - y0f;
    const int x0 = min(max((int)x0f, 0), p.sw - 1), x1 = min(max((int)x0f + 1, 0), p.sw - 1);
    const int y0 = min(max((int)y0f, 0), p.sh - 1), y1 = min(max((int)y0f + 1, 0), p.sh - 1);
    const uchar4* src = reinterpret_cast<const uchar4*>(p.src);
    const uchar4 t00 = src[(size_t)y0 * p.sw + x0], t10 = src[(size_t)y0 * p.sw + x1];
    const uchar4 t01 = src[(size_t)y1 * p.sw + x0], t11 = src[(size_t)y1 * p.sw + x1];
    auto mix = [&](uint32_t c00, uint32_t c10, uint32_t c01, uint32_t c11) -> uint8_t {
        const float f00 = decode_unorm8(c00), f10 = decode_unorm8(c10), f01 = decode_unorm8(c01), f11 = decode_unorm8(c11);
        const float a = f00 + wx * (f10 - f00), b = f01 + wx * (f11 - f01);
        return unorm8(a + wy * (b - a));
    };
    uchar4 o;
    o.x = mix(t00.x, t10.x, t01.x, t11.x); o.y = mix(t00.y, t10.y, t01.y, t11.y);
    o.z = mix(t00.z, t10.z, t01.z, t11.z); o.w = mix(t00.w, t10.w, t01.w, t11.w);
    reinterpret_cast<uchar4*>(p.dst)[(size_t)py * p.tw + px] = o;
}

hipError_t launch_blit(const BlitParams& p, hipStream_t s)
{
    if (p.tw <= 0 || p.th <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_blit, dim3((unsigned)((p.tw + 63) / 64), (unsigned)((p.th + 3) / 4)), dim3(256), 0, s, p);
    return hipGetLastError();
}

// N-frame accumulation of jittered frames (the offline stand-in for the FSR2 temporal pass): exact integer sums of the
// UNORM8 codes, so the result does not depend on the order the frames arrive in.
__global__ __launch_bounds__(256) void k_accumulate(const uchar4* __restrict__ color, uint4* __restrict__ accum, size_t n, int reset)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uchar4 c = color[i];
    uint4 a = reset ? make_uint4(0, 0, 0, 0) : accum[i];
    a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
    accum[i] = a;
}

// mean of `frames` codes, rounded half up: (2*sum + frames) / (2*frames) in integers.
__global__ __launch_bounds__(256) void k_resolve(const uint4* __restrict__ accum, uchar4* __restrict__ out, size_t n, uint32_t frames)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 a = accum[i];
    const uint32_t d = 2u * frames;
    uchar4 o;
    o.x = (uint8_t)min((2u * a.x + frames) / d, 255u); o.y = (uint8_t)min((2u * a.y + frames) / d, 255u);
    o.z = (uint8_t)min((2u * a.z + frames) / d, 255u); o.w = (uint8_t)min((2u * a.w + frames) / d, 255u);
    out[i] = o;
}

hipError_t launch_accumulate(const void* color, void* accum, size_t n, int reset, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       (const uchar4*)color, (uint4*)accum, n, reset);
    return hipGetLastError();
}

hipError_t launch_resolve(const void* accum, void* out, size_t n, uint32_t frames, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       (const uint4*)accum, (uchar4*)out, n, frames);
    return hipGetLastError();
}

} // namespace vrt
