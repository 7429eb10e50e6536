"""Brick scenes (vrt_scene_from_bricks: 8^3 brick pool + pointer grid + two-level clearance): the same content handed over
densely and in bricks must render bit for bit alike, and like the oracle; BASELINE configs[4] (2048^3, 1.5 % of the bricks
occupied, 3840x2160, max_steps 6144, 4 bounces, AO 4) at its stated size against oracle bands, the oracle reading the same
bricks (a dense 2048^3 volume is 8 GiB and is never built anywhere)."""
import ctypes as C

import numpy as np
import pytest

from helpers import camera_push, compare_planes, metallic_palette

pytestmark = pytest.mark.gpu

GB = ["color8", "depth", "motion", "mask8", "position", "normal8"]
ALL = GB + ["color_f", "hit_id", "hit_voxel", "hit_mask", "steps_primary", "steps_total", "rays_total"]


def _volumes(vrt):
    rng = np.random.default_rng(77)
    yield "cubes64", vrt.synthetic.floating_cubes(64, seed=5, count=90)
    yield "treehouse64", vrt.synthetic.treehouse(64, seed=3)
    g, p = vrt.synthetic.sparse_brick_scene(96, 0.06, seed=9)
    yield "carved96", vrt.synthetic.dense_from_bricks(g, p)
    v = (rng.random((24, 40, 72)) < 0.01).astype(np.uint8) * np.uint8(201)          # non-cubic, isolated voxels, metallic
    v[0, :, :] = 3; v[:, 0, :] |= 5; v[:, :, 71] = 9                                    # walls on three faces of the volume
    yield "speckle", v


@pytest.mark.parametrize("primary_only", [True, False])
def test_brick_scene_equals_dense_scene_and_oracle(vrt, oracle, engine, primary_only):
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.blue_noise_standin(64)
    for name, vol in _volumes(vrt):
        D, H, W = vol.shape
        grid, pool = vrt.synthetic.bricks_from_dense(vol)
        assert (vrt.synthetic.dense_from_bricks(grid, pool) == vol).all()
        sb = vrt.VoxelScene.from_bricks(engine, grid, pool, pal, sky=sky, noise=noise)
        sd = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
        assert (sb.width, sb.height, sb.depth) == (W, H, D)
        osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
        osb = oracle.OracleScene(None, pal, sky=sky, noise=noise, bricks=(grid, pool))
        for res, pos, yaw, pitch in (((96, 64), None, 90.0, 0.0), ((61, 47), (W * 0.45, H * 0.55, D * 0.4), 40.0, 10.0),
                                     ((80, 56), (W * 1.6, H * 0.9, -0.3 * D), 130.0, -15.0), ((64, 40), (float(W // 2), float(H // 2), -8.0), 90.0, 0.0)):
            st = vrt.VoxelRenderSettings.primary_only(res) if primary_only else vrt.VoxelRenderSettings(targetResolution=res)
            st.fsrSetttings.enable = False
            push = camera_push(vrt, (W, H, D), res, pos=pos, yaw=yaw, pitch=pitch, frame=4, jitter=(0.2, -0.1))
            gb = vrt.GeometryStage(engine, st, sb, debug_planes=True).record(push).numpy()
            gd = vrt.GeometryStage(engine, st, sd, debug_planes=True).record(push).numpy()
            engine.synchronize()
            exp = oracle.render(osn, push, oracle.params_from(st.to_c()), nthreads=8)
            assert not compare_planes(gb, exp, ALL), (name, res, "bricks vs oracle")
            assert not compare_planes(gb, gd, ALL), (name, res, "bricks vs dense scene")
            expb = oracle.render(osb, push, oracle.params_from(st.to_c()), planes=["hit_id", "steps_total"], nthreads=8)
            assert (expb["hit_id"] == exp["hit_id"]).all() and (expb["steps_total"] == exp["steps_total"]).all()   # the oracle's two storages agree
        sb.destroy(); sd.destroy()


def test_brick_scene_validation(vrt, engine):
    pal = metallic_palette(vrt)
    grid = np.zeros((2, 2, 2), np.uint32); pool = np.full((1, 8, 8, 8), 7, np.uint8)
    grid[1, 0, 1] = 1
    sc = vrt.VoxelScene.from_bricks(engine, grid, pool, pal)
    assert (sc.width, sc.height, sc.depth) == (16, 16, 16) and 0 < sc.memory_bytes() < (1 << 20)
    st = vrt.VoxelRenderSettings.primary_only((32, 32), vrt.TRAVERSAL_DENSE)
    with pytest.raises(vrt._capi.VrtError):                      # only AUTO applies to a brick scene
        vrt.GeometryStage(engine, st, sc).record(camera_push(vrt, (16, 16, 16), (32, 32)))
    with pytest.raises(vrt._capi.VrtError):
        sc.download()
    bad = grid.copy(); bad[0, 0, 0] = 2                         # points past the pool
    with pytest.raises(vrt._capi.VrtError):
        vrt.VoxelScene.from_bricks(engine, bad, pool, pal)
    bad = grid.copy(); bad[0, 0, 0] = 1                         # two entries share a brick
    with pytest.raises(vrt._capi.VrtError):
        vrt.VoxelScene.from_bricks(engine, bad, pool, pal)
    with pytest.raises(vrt._capi.VrtError):                      # a pool brick nobody references
        vrt.VoxelScene.from_bricks(engine, np.zeros((2, 2, 2), np.uint32), pool, pal)
    empty = vrt.VoxelScene.from_bricks(engine, np.zeros((3, 2, 1), np.uint32), np.zeros((0, 8, 8, 8), np.uint8), pal)
    gb = vrt.GeometryStage(engine, vrt.VoxelRenderSettings.primary_only((32, 24)), empty, debug_planes=True).record(camera_push(vrt, (8, 16, 24), (32, 24)))
    engine.synchronize()
    assert (gb.numpy()["hit_id"] == 0).all()
    sc.destroy(); empty.destroy()


@pytest.fixture(scope="module")
def sparse2048(vrt):
    return vrt.synthetic.sparse_brick_scene(2048, 0.015, seed=5)


def test_config5_sparse2048_4k(vrt, oracle, engine, sparse2048):
    """BASELINE configs[4] at its stated workload: synthetic:sparse2048(seed=5), 3840x2160, max_steps = 6144, max_bounces = 4,
    ao_samples = 4 with the blue-noise sequence of frag:80-89 (512^2 tile)."""
    grid, pool = sparse2048
    N = 2048
    assert grid.shape == (256, 256, 256) and abs(pool.shape[0] / grid.size - 0.015) < 1e-4
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(256, 128), vrt.synthetic.blue_noise_standin(512)
    sc = vrt.VoxelScene.from_bricks(engine, grid, pool, pal, sky=sky, noise=noise)
    mem = sc.memory_bytes()
    assert mem < 2 * (1 << 30), mem                              # 8 GiB dense, 77 GiB as a dense scene with its clearance fields
    osn = oracle.OracleScene(None, pal, sky=sky, noise=noise, bricks=(grid, pool))
    res = (3840, 2160)
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.traceSettings.maxRaySteps = 6144
    st.traceSettings.maxReflections = 4
    st.occlusionSettings.numSamples = 4
    pos, yaw, pitch = vrt.synthetic.default_camera_for(N, N, N)
    cam = vrt.CameraController(position=(pos[0] + 0.3, pos[1] + 0.2, pos[2]), yaw=yaw, pitch=pitch)
    push = vrt.make_push(cam, (N, N, N), res, frame=17)
    names = GB + ["hit_id", "hit_voxel", "rays_total", "steps_total"]
    g = vrt.GeometryStage(engine, st, sc, debug_planes=True).record(push)
    engine.synchronize()
    g = g.numpy()
    hit = g["hit_id"] != 0
    assert 0.3 < hit.mean() < 0.999 and int(g["steps_total"].max()) > 2048 and int(g["rays_total"].max()) > 6
    # the hit cell really holds the reported id
    hv = g["hit_voxel"].astype(np.int64)[hit]
    b = grid[hv[:, 2] >> 3, hv[:, 1] >> 3, hv[:, 0] >> 3]
    assert (b != 0).all() and (pool[b - 1, hv[:, 2] & 7, hv[:, 1] & 7, hv[:, 0] & 7] == g["hit_id"][hit]).all()
    for r0 in (4, 1076, 2150):                                   # bands of 6 rows against the oracle reading the same bricks
        exp = oracle.render_band(osn, push, oracle.params_from(st.to_c()), r0, r0 + 6, planes=names, nthreads=16)
        assert not compare_planes({n: g[n][r0:r0 + 6] for n in names}, exp, names), r0
    # the sharded frame (8 simulated ranks, 16-row strips) equals the unsharded one
    stage = vrt.GeometryStage(engine, st, sc)
    merged = {n: np.zeros_like(g[n]) for n in GB}
    rows = np.arange(res[1])
    for rank in range(8):
        pn = stage.record(push, vrt.make_shard(rank, 8, 16)).numpy()
        engine.synchronize()
        own = ((rows // 16) % 8) == rank
        for n in GB:
            merged[n][own] = pn[n][own]
    assert not compare_planes(merged, g, GB)
    sc.destroy()
