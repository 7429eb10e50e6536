"""Deterministic procedural stand-ins for the reference's assets.

Every binary asset of the reference (treehouse.vox, floatingcolored.vox, mandlebulb.vox,
blue_noise_rgba.png, rustig_koppie.hdr) is a git-LFS pointer in the checkout, so BASELINE's configs
are measured on these generators.  All randomness is a counter-based 32-bit hash, so
the volumes are identical on every machine and numpy version.  Volumes are uint8 arrays indexed
[z, y, x] (C order == the reference's x + y*W + z*W*H upload order, voxel_scene.cpp:99).
"""
import numpy as np


def hash32(x):
    """lowbias32 integer hash on uint32 arrays."""
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def rand_u32(seed: int, n: int, stream: int = 0):
    idx = np.arange(n, dtype=np.uint32)
    return hash32(idx ^ hash32(np.uint32([(seed * 0x9E3779B1 + stream * 0x85EBCA77) & 0xFFFFFFFF]))[0])


def rand_unit(seed: int, n: int, stream: int = 0):
    return (rand_u32(seed, n, stream) >> np.uint32(8)).astype(np.float64) / float(1 << 24)


# ---- palette -----------------------------------------------------------------------------------

def default_palette_rgba8() -> np.ndarray:
    """MagicaVoxel's default palette in scene order (entry i colours voxel id i; entry 0 = empty)."""
    lv = [0xFF, 0xCC, 0x99, 0x66, 0x33, 0x00]
    ramp = [0xEE, 0xDD, 0xBB, 0xAA, 0x88, 0x77, 0x55, 0x44, 0x22, 0x11]
    file_order = []
    for i in range(215):
        file_order.append((lv[i // 36], lv[(i // 6) % 6], lv[i % 6], 255))
    for ch in range(3):
        for v in ramp:
            c = [0, 0, 0, 255]; c[ch] = v; file_order.append(tuple(c))
    for v in ramp:
        file_order.append((v, v, v, 255))
    file_order.append((0, 0, 0, 255))
    pal = np.zeros((256, 4), dtype=np.uint8)
    pal[1:] = np.array(file_order[:255], dtype=np.uint8)      # rotate by one (ogt_vox.h:1834-1840)
    pal[0] = (0, 0, 0, 0)
    return pal


def palette_from_rgba8(rgba8: np.ndarray, metallic=None) -> np.ndarray:
    """(256,4) uint8 -> (256,5) float32 [linear r,g,b,a, metallic]  (voxel_scene.cpp:108-117)."""
    out = np.zeros((256, 5), dtype=np.float32)
    out[:, :4] = np.power(rgba8.astype(np.float32) / np.float32(255.0), np.float32(2.2)).astype(np.float32)
    if metallic is not None:
        out[:, 4] = np.asarray(metallic, dtype=np.float32)
    return out


def default_palette(metallic_ids=(), metallic_value=0.8) -> np.ndarray:
    m = np.zeros(256, dtype=np.float32)
    for i in metallic_ids:
        m[i] = metallic_value
    return palette_from_rgba8(default_palette_rgba8(), m)


# ---- sky / noise ---------------------------------------------------------------------------------

def sky_gradient(w: int = 512, h: int = 256, seed: int = 7) -> np.ndarray:
    """Equirectangular HDR-like sky (h, w, 4) float32: horizon-to-zenith gradient + a bright sun lobe."""
    v = (np.arange(h, dtype=np.float32) + 0.5) / h            # 0 = up (uv.y = asin(-y)/pi + 0.5)
    u = (np.arange(w, dtype=np.float32) + 0.5) / w
    elev = (0.5 - v)[:, None] * np.float32(np.pi)             # +pi/2 .. -pi/2
    az = (u[None, :] - 0.5) * np.float32(2 * np.pi)
    t = np.clip(np.sin(elev) * 0.5 + 0.5, 0, 1)
    zenith = np.array([0.25, 0.45, 0.9], dtype=np.float32)
    horizon = np.array([0.9, 0.85, 0.8], dtype=np.float32)
    ground = np.array([0.25, 0.22, 0.2], dtype=np.float32)
    sky = horizon[None, None, :] * (1 - t[..., None]) + zenith[None, None, :] * t[..., None]
    sky = np.broadcast_to(sky, (h, w, 3)).copy()
    sky[(elev < 0)[:, 0]] = ground
    sun_dir = np.array([1.0, 1.0, 1.0]) / np.sqrt(3.0)
    d = np.stack([np.cos(elev) * np.cos(az), np.sin(elev) * np.ones_like(az), np.cos(elev) * np.sin(az)], -1)
    lobe = np.clip((d @ sun_dir), 0, 1) ** 256
    sky += (lobe[..., None] * np.array([8.0, 7.0, 5.0])).astype(np.float32)
    out = np.ones((h, w, 4), dtype=np.float32)
    out[..., :3] = sky
    return out


def blue_noise_standin(n: int = 512, seed: int = 11) -> np.ndarray:
    """(n, n, 4) uint8 hash noise in place of blue_noise_rgba.png (white, not blue: only its determinism matters)."""
    r = rand_u32(seed, n * n, 3)
    out = np.zeros((n, n, 4), dtype=np.uint8)
    out[..., 0] = (r & 0xFF).reshape(n, n)
    out[..., 1] = ((r >> 8) & 0xFF).reshape(n, n)
    out[..., 2] = ((r >> 16) & 0xFF).reshape(n, n)
    out[..., 3] = 255
    return out


# ---- volumes -------------------------------------------------------------------------------------

def single_voxel(n: int = 8, pos=(4, 4, 4), vid: int = 1) -> np.ndarray:
    v = np.zeros((n, n, n), dtype=np.uint8)
    v[pos[2], pos[1], pos[0]] = vid
    return v


def floating_cubes(N: int = 128, seed: int = 1, count: int = 400) -> np.ndarray:
    """Config 1 stand-in for floatingcolored.vox: `count` axis-aligned cubes of edge 2..10, ids 1..255."""
    v = np.zeros((N, N, N), dtype=np.uint8)
    edge = 2 + (rand_u32(seed, count, 0) % np.uint32(9)).astype(np.int64)
    cx = (rand_unit(seed, count, 1) * N).astype(np.int64)
    cy = (rand_unit(seed, count, 2) * N).astype(np.int64)
    cz = (rand_unit(seed, count, 3) * N).astype(np.int64)
    ids = 1 + (rand_u32(seed, count, 4) % np.uint32(255)).astype(np.int64)
    for i in range(count):
        e = int(edge[i])
        x0, y0, z0 = int(cx[i]), int(cy[i]), int(cz[i])
        v[z0:min(N, z0 + e), y0:min(N, y0 + e), x0:min(N, x0 + e)] = ids[i]
    return v


def treehouse(N: int = 256, seed: int = 2) -> np.ndarray:
    """Config 2/3 stand-in for treehouse.vox: ground slab, trunk, shell "rooms" with windows, a sparse
    spherical canopy.  Sky, near, far and occluded pixels all occur from the default camera."""
    s = N / 256.0
    v = np.zeros((N, N, N), dtype=np.uint8)
    z, y, x = np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij", sparse=True)
    c = N // 2
    v[:, : int(8 * s), :] = 1                                              # ground slab
    trunk = ((x - c) ** 2 + (z - c) ** 2 <= (10 * s) ** 2) & (y >= int(8 * s)) & (y < int(128 * s))
    v[np.broadcast_to(trunk, v.shape)] = 2
    rooms = [(-70, 60, -40), (40, 70, 30), (-30, 110, 50), (60, 120, -60), (-80, 130, -70), (10, 40, 80)]
    for k, (ox, oy, oz) in enumerate(rooms):
        e = int(40 * s)
        x0, y0, z0 = c + int(ox * s), int(oy * s), c + int(oz * s)
        x0 = max(0, min(N - e, x0)); y0 = max(0, min(N - e, y0)); z0 = max(0, min(N - e, z0))
        box = np.zeros((e, e, e), dtype=np.uint8)
        box[:] = 3 + k
        t = max(1, int(2 * s))
        box[t:-t, t:-t, t:-t] = 0                                          # hollow shell
        w0, w1 = e // 3, 2 * e // 3
        box[w0:w1, w0:w1, :] = 0                                           # windows through x faces
        box[:, w0:w1, w0:w1] = 0                                           # windows through z faces
        v[z0:z0 + e, y0:y0 + e, x0:x0 + e] = np.where(box > 0, box, v[z0:z0 + e, y0:y0 + e, x0:x0 + e])
    canopy = ((x - c) ** 2 + (y - int(170 * s)) ** 2 + (z - c) ** 2 <= (70 * s) ** 2) & \
             ((x - c) ** 2 + (y - int(170 * s)) ** 2 + (z - c) ** 2 >= (62 * s) ** 2)
    canopy = np.broadcast_to(canopy, v.shape)
    idx = np.flatnonzero(canopy)
    keep = rand_unit(seed, idx.size, 9) >= 0.35                            # 35 % random removal
    leaf_ids = 10 + (rand_u32(seed, idx.size, 10) % np.uint32(6)).astype(np.uint8)
    flat = v.reshape(-1)
    flat[idx[keep]] = leaf_ids[keep]
    return v


def mandelbulb(N: int = 128, power: float = 8.0, iters: int = 8, slab: int = 16, workers: int = 8) -> np.ndarray:
    """Config 4 stand-in for mandlebulb.vox: escape-time Mandelbulb; id by escape iteration, interior 200+.
    Slabs of `slab` z-layers are independent and are computed by `workers` threads (numpy releases the GIL)."""
    v = np.zeros((N, N, N), dtype=np.uint8)
    lin = (np.arange(N, dtype=np.float32) + 0.5) / N * 2.4 - 1.2

    def one(z0):
        zz, yy, xx = np.meshgrid(lin[z0:z0 + slab], lin, lin, indexing="ij")
        cx, cy, cz = xx, yy, zz
        px, py, pz = cx.copy(), cy.copy(), cz.copy()
        esc = np.zeros(cx.shape, dtype=np.int32)
        alive = np.ones(cx.shape, dtype=bool)
        for it in range(iters):
            r = np.sqrt(px * px + py * py + pz * pz)
            out = alive & (r > 2.0)
            esc[out] = it + 1
            alive &= ~out
            r_safe = np.where(r > 1e-9, r, 1e-9)
            theta = np.arccos(np.clip(pz / r_safe, -1, 1)) * power
            phi = np.arctan2(py, px) * power
            rp = np.where(alive, r_safe, 1.0) ** power
            px = np.where(alive, rp * np.sin(theta) * np.cos(phi) + cx, px)
            py = np.where(alive, rp * np.sin(theta) * np.sin(phi) + cy, py)
            pz = np.where(alive, rp * np.cos(theta) + cz, pz)
        inside = alive
        ids = np.zeros(cx.shape, dtype=np.uint8)
        ids[inside] = 200 + (hash32((np.flatnonzero(inside) + z0 * N * N).astype(np.uint32)) % np.uint32(56)).astype(np.uint8)
        near = (~inside) & (esc >= iters - 1)                              # thin shell of late escapers
        ids[near] = (20 + esc[near]).astype(np.uint8)
        v[z0:z0 + slab] = ids

    starts = list(range(0, N, slab))
    if workers > 1 and len(starts) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(workers)) as ex:
            list(ex.map(one, starts))
    else:
        for z0 in starts:
            one(z0)
    return v


def sparse_bricks(N: int = 2048, brick: int = 8, fill: float = 0.015, seed: int = 5) -> np.ndarray:
    """Config 5 stand-in (sparse2048): N^3 volume of brick^3 cells, a fraction `fill` of them occupied.  Occupied bricks
    follow a thresholded value-noise field (trilinear over a hashed lattice of period 8 bricks) so that they cluster;
    each occupied brick is solid with one id in 1..255 (ids 200..255 are the metallic ones of the default test palette)."""
    nb = N // brick
    assert nb * brick == N
    lat = nb // 8 + 2
    g = rand_unit(seed, lat * lat * lat, 0).reshape(lat, lat, lat).astype(np.float32)
    t = (np.arange(nb, dtype=np.float32) + 0.5) / 8.0
    i0 = np.floor(t).astype(np.int64); f = (t - i0).astype(np.float32)
    def lerp_axis(a, axis):
        lo = np.take(a, i0, axis=axis); hi = np.take(a, i0 + 1, axis=axis)
        shape = [1, 1, 1]; shape[axis] = nb
        w = f.reshape(shape)
        return lo * (1 - w) + hi * w
    field = lerp_axis(lerp_axis(lerp_axis(g, 0), 1), 2)                    # nb^3 smooth noise in [0, 1]
    field = field + 0.15 * (rand_unit(seed, nb * nb * nb, 1).reshape(nb, nb, nb).astype(np.float32) - 0.5)
    thr = np.quantile(field, 1.0 - fill)
    occ = field > thr
    ids = (1 + (rand_u32(seed, nb * nb * nb, 2) % np.uint32(255))).astype(np.uint8).reshape(nb, nb, nb)
    cells = np.where(occ, ids, np.uint8(0))
    return np.repeat(np.repeat(np.repeat(cells, brick, axis=0), brick, axis=1), brick, axis=2)


def _brick_cells(N: int, brick: int, fill: float, seed: int):
    """Occupancy and id per brick of sparse_bricks / sparse_brick_scene: (occ bool [nb]^3, ids uint8 [nb]^3)."""
    nb = N // brick
    assert nb * brick == N
    lat = nb // 8 + 2
    g = rand_unit(seed, lat * lat * lat, 0).reshape(lat, lat, lat).astype(np.float32)
    t = (np.arange(nb, dtype=np.float32) + 0.5) / 8.0
    i0 = np.floor(t).astype(np.int64); f = (t - i0).astype(np.float32)
    def lerp_axis(a, axis):
        lo = np.take(a, i0, axis=axis); hi = np.take(a, i0 + 1, axis=axis)
        shape = [1, 1, 1]; shape[axis] = nb
        w = f.reshape(shape)
        return lo * (1 - w) + hi * w
    field = lerp_axis(lerp_axis(lerp_axis(g, 0), 1), 2)
    field = field + 0.15 * (rand_unit(seed, nb * nb * nb, 1).reshape(nb, nb, nb).astype(np.float32) - 0.5)
    thr = np.quantile(field, 1.0 - fill)
    ids = (1 + (rand_u32(seed, nb * nb * nb, 2) % np.uint32(255))).astype(np.uint8).reshape(nb, nb, nb)
    return field > thr, ids


def sparse_brick_scene(N: int = 2048, fill: float = 0.015, seed: int = 5, carve: bool = True):
    """BASELINE configs[4] (synthetic:sparse2048(seed=5)): an N^3 volume generated BRICK-WISE -- never as a dense array (2048^3
    is 8 GiB).  Returns (grid uint32 [nbz, nby, nbx], pool uint8 [n, 8, 8, 8]): grid = 0 for an empty brick, else 1 + its index
    in the pool.  A fraction `fill` of the 8^3 bricks is occupied, clustered by thresholded value noise (as sparse_bricks);
    carve: each occupied brick is its id where a hash of the voxel's position keeps it (about 70 %), so that surfaces are
    rough at the voxel scale; carve=False: solid bricks, dense_from_bricks() of it equals sparse_bricks(N, 8, fill, seed)."""
    occ, ids = _brick_cells(N, 8, fill, seed)
    nb = N // 8
    where = np.flatnonzero(occ.reshape(-1))
    grid = np.zeros(nb * nb * nb, np.uint32)
    grid[where] = np.arange(1, where.size + 1, dtype=np.uint32)
    pool = np.empty((where.size, 8, 8, 8), np.uint8)
    pool[:] = ids.reshape(-1)[where][:, None, None, None]
    if carve and where.size:
        v = np.arange(512, dtype=np.uint32)
        h = hash32(where.astype(np.uint32)[:, None] * np.uint32(512) + v[None, :] + np.uint32((seed * 0x9E3779B1) & 0xFFFFFFFF))
        keep = (h % np.uint32(10)) < np.uint32(7)
        pool = np.where(keep.reshape(-1, 8, 8, 8), pool, np.uint8(0))
    return grid.reshape(nb, nb, nb), pool


def bricks_from_dense(vol: np.ndarray):
    """Dense [z, y, x] volume (dimensions multiples of 8) -> (grid, pool) of vrt_scene_from_bricks."""
    D, H, W = vol.shape
    assert D % 8 == 0 and H % 8 == 0 and W % 8 == 0
    b = vol.reshape(D // 8, 8, H // 8, 8, W // 8, 8).transpose(0, 2, 4, 1, 3, 5).reshape(-1, 8, 8, 8)
    occ = b.reshape(b.shape[0], -1).any(axis=1)
    where = np.flatnonzero(occ)
    grid = np.zeros(b.shape[0], np.uint32)
    grid[where] = np.arange(1, where.size + 1, dtype=np.uint32)
    return grid.reshape(D // 8, H // 8, W // 8), np.ascontiguousarray(b[where])


def dense_from_bricks(grid: np.ndarray, pool: np.ndarray) -> np.ndarray:
    nbz, nby, nbx = grid.shape
    b = np.zeros((grid.size, 8, 8, 8), np.uint8)
    where = np.flatnonzero(grid.reshape(-1))
    b[where] = pool[grid.reshape(-1)[where] - 1]
    return np.ascontiguousarray(b.reshape(nbz, nby, nbx, 8, 8, 8).transpose(0, 3, 1, 4, 2, 5).reshape(nbz * 8, nby * 8, nbx * 8))


def default_camera_for(N_x: int, N_y: int, N_z: int):
    """The reference default camera (8,8,-50)/yaw 90/pitch 0 (voxel_renderer.cpp:20) scaled to the volume."""
    return (N_x / 2.0, N_y / 2.0, -0.8 * N_z), 90.0, 0.0
