// vrt_internal.h -- structures shared between the C-ABI layer (vrt_api.cpp) and the HIP kernels
// (vrt_device.hip).  Not part of the public interface.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vrt.h"
#include "vrt_traverse.h"
#include "vrt_sky.h"
#include "vrt_tags.h"

namespace vrt {

// Device view of a scene.  Layout in HBM (all hipMalloc'd, read-only during rendering):
//   vol.vox   W*H*D bytes, index x + y*W + z*W*H   (the reference's Texture3D upload order)
//   vol.occ1  one u64 per 4^3 voxels  : bit (x&3) | (y&3)<<2 | (z&3)<<4 set <=> voxel != 0      [n1x*n1y*n1z]
//   vol.occ2  one u64 per 16^3 voxels : bit over the 4x4x4 occ1 words, set <=> word != 0        [n2x*n2y*n2z]
//   vol.occ3  one u64 per 64^3 voxels : same over occ2                                           [n3x*n3y*n3z]
//   vol.df    8 octant clearance fields of (W+2)(H+2)(D+2) bytes each (x-fastest, one-voxel border of zeros): 0 = solid,
//             else min(63, side of the largest empty cube cornered at the voxel and extending towards the octant's signs)
struct DevScene {
    VolumeView vol;
    const vrt_material* palette;
    const float*   sky;   uint32_t sky_w, sky_h;
    const float*   sky_normals;   // 64 x float4: skyColor(n) for the 26 normals a hit can have, index = mask | (sx<0)<<3 |
                                  // (sy<0)<<4 | (sz<0)<<5 (k_sky_normals; rebuilt whenever the sky changes)
    const uint8_t* noise; uint32_t noise_w, noise_h;
    // the sky once more as RGBA8 (unorm8 of every texel's r, g, b; a = 0: what a miss pixel's colour target holds) and the
    // constants of the texel fast path (vrt_sky.h), both made when the sky is set
    const uint32_t* sky8;
    SkyFastConsts  skyk;
};

struct ShardMap {
    int32_t rank, nranks, strip_rows;   // strip_rows multiple of 16
    int32_t n_local_strips;
    int32_t tiles_per_strip;            // strip_rows / 16
};

// The pixel-independent part of main()'s ray generation (voxel_volume.frag:312-319), evaluated once on the host with
// the same fp32 operations the shader performs per fragment.
struct RayGenConsts {
    f3 cd;             // normalize(camDir.xyz)
    f3 planeV;         // camUp.xyz * H / W
    float jx, jy;      // cameraJitter / screenSize * (-2, 2)
    float W, H;
};

inline RayGenConsts raygen_consts(const vrt_push& pc)
{
    RayGenConsts g;
    g.W = (float)pc.screen_size[0]; g.H = (float)pc.screen_size[1];
    g.cd = normalize3(mk3(pc.cam_dir[0], pc.cam_dir[1], pc.cam_dir[2]));
    g.planeV = mk3((pc.cam_up[0] * g.H) / g.W, (pc.cam_up[1] * g.H) / g.W, (pc.cam_up[2] * g.H) / g.W);
    g.jx = (pc.camera_jitter[0] / g.W) * -2.0f;
    g.jy = (pc.camera_jitter[1] / g.H) * 2.0f;
    return g;
}

// Conservative screen rectangle of the volume for one frame's camera, computed on the host in double precision: the rays
// through pixels outside it miss the box [-m, dim + m]^3 (m = one voxel) and so, by a margin thousands of times the
// rounding of the shader's fp32 slab test (frag:109-125), miss the volume: their waves skip the box test altogether and
// write what a miss writes.  Any doubt -- camera inside or near the box, a corner behind the camera, a degenerate basis --
// returns the whole screen.
inline void box_rect(const vrt_push& pc, uint8_t out[4])
{
    out[0] = 0; out[1] = 255; out[2] = 0; out[3] = 255;       // = "no rectangle": the kernel tests nothing
    const double W = pc.screen_size[0], H = pc.screen_size[1];
    if (W > 8128.0 || H > 8128.0) return;                     // (a byte counts units of 32 pixels)
    const double dim[3] = {(double)pc.volume_bounds[0], (double)pc.volume_bounds[1], (double)pc.volume_bounds[2]};
    const double cam[3] = {pc.cam_pos[0], pc.cam_pos[1], pc.cam_pos[2]};
    const double m = 1.0;
    bool inside = true;
    for (int a = 0; a < 3; a++) inside = inside && cam[a] >= -2.0 * m && cam[a] <= dim[a] + 2.0 * m;
    if (inside || !(W >= 1.0) || !(H >= 1.0)) return;
    // ray of screen position (sx, sy): cd + sx U + sy V + J, as main() builds it (frag:312-319)
    double cd[3] = {pc.cam_dir[0], pc.cam_dir[1], pc.cam_dir[2]};
    const double l = sqrt(cd[0] * cd[0] + cd[1] * cd[1] + cd[2] * cd[2]);
    if (!(l > 1e-20) || !(l < 1e20)) return;
    const double U[3] = {pc.cam_right[0], pc.cam_right[1], pc.cam_right[2]};
    const double V[3] = {pc.cam_up[0] * H / W, pc.cam_up[1] * H / W, pc.cam_up[2] * H / W};
    const double C[3] = {cd[0] / l + pc.camera_jitter[0] / W * -2.0, cd[1] / l + pc.camera_jitter[1] / H * 2.0, cd[2] / l};
    // [U V C] (a, b, lambda)^T = P - cam  by Cramer's rule
    auto det3 = [](const double* x, const double* y, const double* z) {
        return x[0] * (y[1] * z[2] - y[2] * z[1]) - y[0] * (x[1] * z[2] - x[2] * z[1]) + z[0] * (x[1] * y[2] - x[2] * y[1]);
    };
    const double det = det3(U, V, C);
    const double scale = (fabs(U[0]) + fabs(U[1]) + fabs(U[2])) * (fabs(V[0]) + fabs(V[1]) + fabs(V[2])) * (fabs(C[0]) + fabs(C[1]) + fabs(C[2]));
    if (!(fabs(det) > 1e-9 * scale) || !(scale > 0.0)) return;
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    for (int k = 0; k < 8; k++) {
        const double P[3] = {((k & 1) ? dim[0] + m : -m) - cam[0], ((k & 2) ? dim[1] + m : -m) - cam[1], ((k & 4) ? dim[2] + m : -m) - cam[2]};
        const double lam = det3(U, V, P) / det, a = det3(P, V, C) / det, b = det3(U, P, C) / det;
        const double reach = fabs(P[0]) + fabs(P[1]) + fabs(P[2]);
        if (!(lam > 1e-3 * reach / (fabs(C[0]) + fabs(C[1]) + fabs(C[2]) + 1e-30))) return;      // behind, or too close to, the camera plane
        const double px = (a / lam + 1.0) * 0.5 * W, py = (b / lam + 1.0) * 0.5 * H;
        if (!(px == px) || !(py == py)) return;
        x0 = px < x0 ? px : x0; x1 = px > x1 ? px : x1; y0 = py < y0 ? py : y0; y1 = py > y1 ? py : y1;
    }
    const double margin = 4.0;                                     // pixels
    auto lo = [](double v) { v = floor(v / 32.0); return v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v); };
    auto hi = [](double v) { v = ceil(v / 32.0); return v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v); };
    out[0] = (uint8_t)lo(x0 - margin); out[1] = (uint8_t)hi(x1 + margin);
    out[2] = (uint8_t)lo(y0 - margin); out[3] = (uint8_t)hi(y1 + margin);
}

// (i + 0.5) / n for i in [0, n) as q = x * r; q += fma(-q, n, x) * r with r = 1 / n: one multiply and two FMAs instead of
// the ~11 instructions of an IEEE division.  The sequence is the correctly rounded quotient for almost every divisor;
// whether it is for THIS one is decided by trying all n numerators (the host does so once per screen size).
__host__ __device__ inline float screen_div_fast(float x, float n, float r)
{
    float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, n, x), r, q);
}
inline bool screen_div_exact(int n)
{
    const float fn = (float)n, r = 1.0f / fn;
    for (int i = 0; i < n; i++) {
        const float x = (float)i + 0.5f;
        volatile float want = x / fn;
        if (screen_div_fast(x, fn, r) != want) return false;
    }
    return true;
}

// One frame of a launch: camera (push block + its hoisted part), output planes and the strip assignment it is traced
// with (the rank of ShardMap; frames of one launch may play different ranks: vrt_render_geometry_slots).
struct FrameSlot {
    RayGenConsts rg;
    vrt_push     pc;
    vrt_frame    fr;
    int32_t      shard_rank;
    uint8_t      box[4];       // screen rectangle outside which no primary ray of this frame can meet the volume, in units of
                               // 32 pixels: columns [box[0], box[1]) x rows [box[2], box[3]) (box_rect(); {0, 255, 0, 255} = all)
};
static_assert(sizeof(FrameSlot) == 256, "FrameSlot is sized for aligned scalar loads");

#define VRT_MAX_BATCH 8      // frames per K1 launch whose slots travel in the kernel arguments (8 x 256 B)
#define VRT_MAX_TABLE 256    // frames per K1 launch whose slots are read from a table in device memory

// Everything a workgroup needs to find its tile, in one 64-byte block of the kernel arguments: a wave fetches it with one
// scalar load and one wait instead of ten loads of one or two dwords, each waited for before the next could be issued.
#define VRT_MAPFLAG_SKY_FAST 0x10000u   // TileMap::flags: GeomParams::sky_fast, where the wave finds it without a load of its own
#define VRT_MAPFLAG_SIX      0x20000u   // every frame of the launch holds exactly the reference's six targets (color8, depth, motion, mask8,
                                        // position, normal8; geometry_stage.hpp:19-27): the stores need no test of their pointers
struct alignas(64) TileMap {
    uint32_t flags;                    // vrt_settings.flags (low 16 bits) | VRT_MAPFLAG_*
    int32_t  n_frames, xcd_turn;
    uint32_t wgs_per_frame, wgs_per_frame_rcp;
    int32_t  tiles_x;       uint32_t tiles_x_rcp;
    int32_t  tiles_y_local; uint32_t tiles_y_rcp;
    uint32_t tps, tps_rcp;
    int32_t  tile;                     // tile_w == tile_h
    int32_t  nranks, strip_rows;
    int32_t  W, H;
};
static_assert(sizeof(TileMap) == 64, "TileMap is one 16-dword scalar load");

struct GeomParams {
    // A launch covers n_frames frames of one scene, one resolution and one set of settings (consecutive camera poses of an
    // animation, or the frames of a multi-GPU batch): workgroup b works on frame b / wgs_per_frame.  Frame f+1's tiles are
    // dispatched while frame f drains, so the ~30 us tail of a frame is paid once per launch instead of once per frame.
    FrameSlot  slot[VRT_MAX_BATCH];
    TileMap    map;            // copies of the fields below that the tile mapping reads (filled last by the host)
    const FrameSlot* table;    // non-null: n_frames (<= VRT_MAX_TABLE) slots in device memory instead of slot[]
    int32_t    n_frames;
    int32_t    xcd_turn;       // 1: tile rows of all frames are dealt to the XCDs in one sequence (block_to_tile)
    uint32_t   wgs_per_frame, wgs_per_frame_rcp;   // chunk * 8 and floor(2^32 / that)
    int32_t    W, H;                               // screen size (common to the frames)
    float      rcp_w, rcp_h;                       // 1 / W, 1 / H
    int32_t    fast_screen_div;                    // screen_div_fast is exact for every pixel centre of this W and H
    DevScene   sc;
    vrt_settings st;
    ShardMap   sh;
    int32_t    tile_w, tile_h; // workgroup tile in pixels: 16x16 (4 waves of 8x8) or 8x8 (one wave)
    int32_t    tiles_x, tiles_y_local, total_tiles, chunk;  // chunk = tiles per XCD slot
    uint32_t   tiles_y_rcp;    // floor(2^32 / tiles_y_local), for xcd_turn
    uint32_t   tiles_x_rcp, tps, tps_rcp;  // floor(2^32 / d) for the two wave-uniform divisions of tile_origin (scalar unit);
                                       // tps = tiles per strip (strip_rows / tile_h)
    uint4*     records;        // per-pixel primary hit record for the shading kernel (full-frame indexing)
    uint32_t*  hit_count;      // K1 -> K2: number of hit pixels (zeroed before K1)
    uint32_t*  hit_list;       // K1 -> K2: their pixel indices, in arrival order
    int32_t    fused_shade;    // 1: primary kernel shades inline (no secondary rays enabled); 2: megakernel; 0: split
    int32_t    no_bounce;      // 1: no ray of the launch can bounce (max_bounces == 0 or no metallic voxel in the scene): the megakernel
                               //    without the bounce loop and its scratch stack
    int32_t    packed_chain;   // 1: the megakernel keeps its bounce chain as one word per hit (MODE 5, color_main_ray_packed); 0: the stack of hits
    int32_t    fast_loop;      // 1: AUTO / DF run the hand-written look-up loop (trace_df_fast; the host checked its preconditions); 2: its counting twins
    int32_t    occ_in_lds;     // 1: stage occ2 + occ3 into LDS, 0: read them through L2
    int32_t    sky_fast;       // 1: waves that cannot hit anything decide their sky texel by vrt_sky.h and write the miss pixel
                               //    without normalising the ray (no diagnostic planes in the launch, W * H < 2^28)
    uint32_t   occ2_bytes, occ3_bytes;   // both multiples of 16
    // Tile tags (k_tile_tags, launched ahead of K1): one word per 8x8-pixel block of every frame of the launch, == tile_gen
    // where a primary ray of the block can meet an occupied 4^3 cell of the volume; word tags_per_frame - 1 of a frame ==
    // tile_gen: the frame's tags say nothing (trace every block).  nullptr: no tags in this launch.
    uint32_t*  tile_tags;
    uint32_t   tile_gen, tags_x, tags_y, tags_per_frame;
    const uint32_t* cells;     // the scene's occupied cells (4^3 voxels; 8^3 = the bricks of a brick scene), x | y << 10 | z << 20
    uint32_t   n_cells, cell_size;
    // Primary rays only (fused_shade == 1: ambient = 1, nothing shadowed, no reflection): the colour of a hit is a function of
    // its material and of which of the 26 normals it has -- colorHit() for all 256 x 64 of them, made by colorHit() itself
    // (k_hit_colors) whenever the settings or the scene's sky change, as the RGBA8 the colour target stores:
    // hit_colors[material << 6 | ncode].  nullptr: every pixel computes its own.
    const uint32_t* hit_colors;
};

#define VRT_DENOISE_SEGS 64    // counters of redone pixels (diagnostics): a workgroup adds to counter (its index % 64)
struct DenoiseParams {
    const uint8_t* color_in;
    const int8_t*  normal;
    const float*   position;
    uint8_t*       color_out;
    int32_t W, H;
    float phi_color, phi_normal, phi_pos, step_width;
    int32_t mode;
    float   kc, kn, kp;        // VRT_DENOISE_FAST: log2(e) / phi per channel (kn also / stepWidth^2)
    float   rc, rn, rp, rs;    // RN(1 / phi) per channel, RN(1 / stepWidth^2): the exact kernel's divisions by pass-uniform values
    int32_t packed_ok;         // 1: phi and stepWidth^2 are in the range the division by reciprocal + residuals is exact for
    int32_t extend;            // rows beyond each owned strip that this pass must also produce
    int32_t tile16;            // development switch (context option "denoise_th16"): the tolerance kernel on 64 x 16 tiles instead of 64 x 8
    int32_t no_packed;         // development switch (context option "denoise_packed" = 0): the exact weighted pass tap by tap
    int32_t no_pair;           // development switch (context option "denoise_pair" = 0): k_denoise_ver for every verified pass
    int32_t no_p0;             // development switch (context option "denoise_p0" = 0): pass 0 through k_denoise_ver
    int32_t pair_wgs;          // development switch (context option "denoise_pair_wgs" > 0): workgroups of a k_denoise_pair launch
    // the verified pass (k_denoise_ver, vrt_denoise_bound.h): scale factors on the integer code distances and on the position
    // distance (log2(e) / phi, the colour's also / 255^2, the normal's / (127^2 stepWidth^2); computed in double), the guard in
    // RGBA8 codes, and -- diagnostics, else NULL -- VRT_DENOISE_SEGS counters of the pixels the pass evaluated twice
    float   vkc, vkn, vkp, guard;
    uint32_t* fix_counts;
    int32_t verified;          // 1: this pass may take k_denoise_ver (host: eligible and the guard is small)
    ShardMap sh;
};

struct RowsParams {
    const uint8_t* src; uint8_t* dst;
    int32_t W, H, bpp;
    ShardMap sh;
    int32_t halo, dir, unpack;
};

#define VRT_ROWS_BATCH 64     // images per k_rows_batch launch (40 B of kernel arguments each)
struct RowsBatchParams {     // strip pack / unpack of many images in one launch: image blockIdx.y, packed row blockIdx.x
    const uint8_t* src[VRT_ROWS_BATCH];
    uint8_t*       dst[VRT_ROWS_BATCH];
    ShardMap       sh[VRT_ROWS_BATCH];
    int32_t W, H, bpp, unpack;
    int32_t halo, dir;       // halo > 0: the first (dir < 0) / last (dir > 0) `halo` rows of every owned strip instead of whole strips
};

struct BlitParams {
    const uint8_t* src; uint8_t* dst;
    int32_t sw, sh, tw, th;
};

// launchers (vrt_device.hip)
hipError_t launch_build_pyramid(const uint8_t* vox, int W, int H, int D, uint64_t* occ1, uint64_t* occ2,
                                uint64_t* occ3, hipStream_t s);
hipError_t launch_build_df(const uint8_t* vox, int W, int H, int D, uint8_t* df, size_t stride, uint8_t* tmp0, uint8_t* tmp1, hipStream_t s, int cap = 0 /* 0: the dense scene's cap */);
hipError_t launch_brick_grid(const uint32_t* grid, int nbx, int nby, int nbz, uint32_t* padded, uint8_t* occ, hipStream_t s);
hipError_t launch_brick_pack(const uint32_t* padded, const uint8_t* coarse, size_t cstride, size_t npad, uint64_t* entry, hipStream_t s);
hipError_t launch_brick_fine(const uint32_t* padded, int pbx, int pby, const uint32_t* coord, uint32_t n_bricks, const uint8_t* pool,
                             uint8_t* fine, hipStream_t s);
hipError_t launch_sky_normals(const DevScene& sc, float* table, hipStream_t s);
hipError_t launch_sky_rgba8(const float* sky, uint32_t* sky8, size_t n, hipStream_t s);
hipError_t debug_brick_counts(unsigned long long out[4]);
hipError_t launch_debug_sky(const DevScene& sc, const float* v, size_t n, uint32_t* out, hipStream_t s);
hipError_t launch_open_cells(const uint8_t* vox, int W, int H, int D, uint8_t* df, size_t stride, uint8_t* tmp0, uint8_t* tmp1, hipStream_t s, int mark = 0);
hipError_t launch_tile_tags(const GeomParams& p, hipStream_t s);
hipError_t launch_hit_colors(const GeomParams& p, uint32_t* table, hipStream_t s);
hipError_t launch_pad_vox(const uint8_t* vox, int W, int H, int D, uint8_t* dst, hipStream_t s);
hipError_t launch_primary(const GeomParams& p, hipStream_t s);
hipError_t launch_shade(const GeomParams& p, hipStream_t s);
hipError_t launch_denoise_pass(const DenoiseParams& p, hipStream_t s);
hipError_t launch_rows(const RowsParams& p, int rows_total, hipStream_t s);
hipError_t launch_rows_batch(const RowsBatchParams& p, int rows_total, int images, hipStream_t s);
hipError_t launch_blit(const BlitParams& p, hipStream_t s);
hipError_t launch_accumulate(const void* color_rgba8, void* accum_u32x4, size_t n, int reset, hipStream_t s);
hipError_t launch_resolve(const void* accum_u32x4, void* out_rgba8, size_t n, uint32_t frames, hipStream_t s);
const char* primary_kernel_name(int traversal, int fused, int occ2_lds);

} // namespace vrt
