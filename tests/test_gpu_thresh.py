"""Long runs by threshold (vrt_traverse.h df_prim_loop / brick_march_thresh; context option thresh_runs): every axis steps on its
own while its sideDist is below the lane's threshold -- the same fp32 additions per lane as the shader's merged loop, so every
plane a caller of the product gets must come out bit for bit as with thresh_runs = 0 and as the oracle computes it.
The loop keeps no iteration count and is entered only by waves whose rays provably cannot reach the budget: the cases below sit
on both sides of that bound (budgets from 32 upwards against rays of 20 ... 350 cells), on ties and axis-parallel rays (two
axes holding the same value step in ONE iteration of the shader's loop), inside the volume, on secondary rays (shadow and bounce
rays take the same loop; AO rays never do), on brick scenes, and on a randomised sweep."""
import ctypes as C

import numpy as np
import pytest

from helpers import camera_push, compare_planes, metallic_palette

pytestmark = pytest.mark.gpu

GB = ["color8", "depth", "motion", "mask8", "position", "normal8"]
PRODUCT = GB + ["color_f", "hit_id", "hit_mask", "rays_total"]      # no iteration counts: the launch the threshold loop serves


def _render(vrt, engine, sc, st, push, thresh, planes=PRODUCT):
    W, H = st.renderResolution()
    gb = vrt.GeometryBuffer(engine, W, H, planes)
    stc, fr = st.to_c(), gb.to_c()
    with engine.options(thresh_runs=thresh):
        vrt._capi.check(vrt.lib().vrt_render_geometry(engine.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
        engine.synchronize()
    return gb.numpy()


def _check(vrt, oracle, engine, sc, osn, st, push, what, planes=PRODUCT):
    on = _render(vrt, engine, sc, st, push, 1, planes)
    off = _render(vrt, engine, sc, st, push, 0, planes)
    exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=planes, nthreads=8)
    bad = compare_planes(on, exp, planes)
    assert not bad, (what, "threshold runs vs oracle", bad)
    assert not compare_planes(off, exp, planes), (what, "merged runs vs oracle")
    return exp


def _scene(vrt, oracle, engine, vol, sky=(64, 32), noise=64):
    pal = metallic_palette(vrt)
    s, n = vrt.synthetic.sky_gradient(*sky), vrt.synthetic.blue_noise_standin(noise)
    return vrt.VoxelScene.from_dense(engine, vol, pal, sky=s, noise=n), oracle.OracleScene(vol, pal, sky=s, noise=n)


@pytest.mark.parametrize("max_steps", [32, 33, 48, 64, 100, 150, 192, 200, 260, 512, 1024])
def test_budgets_either_side_of_the_bound(vrt, oracle, engine, max_steps):
    """a 24 x 24 x 200 tunnel with a wall at its far end: rays along it cross ~190 planes, rays across it ~24; whether a WAVE's
    bound stays below the budget changes from block to block and from budget to budget, and a ray that runs out of budget must
    do so on exactly the reference's iteration"""
    vol = np.zeros((200, 24, 24), np.uint8); vol[190:, :, :] = 3
    vol[60, 5:9, 5:9] = 201; vol[120, 14:20, 2:6] = 7
    sc, osn = _scene(vrt, oracle, engine, vol)
    for res, pos, yaw, pitch in (((48, 48), (12.0, 12.0, -3.0), 90.0, 0.0), ((64, 40), (11.3, 12.6, 20.2), 88.0, 1.5),
                                 ((40, 56), (-10.0, 12.0, 100.0), 0.0, 0.0), ((56, 32), (30.0, 40.0, 230.0), 250.0, -35.0)):
        st = vrt.VoxelRenderSettings.primary_only(res); st.traceSettings.maxRaySteps = max_steps
        push = camera_push(vrt, (24, 24, 200), res, pos=pos, yaw=yaw, pitch=pitch, frame=3, jitter=(0.1, 0.3))
        exp = _check(vrt, oracle, engine, sc, osn, st, push, (max_steps, res))
    sc.destroy()


def test_ties_axis_parallel_and_lattice_cameras(vrt, oracle, engine):
    vol = np.zeros((48, 48, 48), np.uint8)
    vol[40, :, :] = 5; vol[20, 12:36, 12:36] = 9; vol[8:30, 30:34, 30:34] = 7; vol[30, 20:22, 20:22] = 200
    sc, osn = _scene(vrt, oracle, engine, vol)
    res = (48, 40)
    st = vrt.VoxelRenderSettings.primary_only(res)
    # exactly axis-aligned view directions (inf deltas on the centre row / column), from outside and from lattice points inside
    for pos, d in (((24.0, 24.0, -16.0), (0.0, 0.0, 1.0)), ((2.0, 25.0, 3.0), (1.0, 0.0, 0.0)), ((25.0, 46.0, 3.0), (0.0, -1.0, 0.0)),
                   ((24.0, 24.0, 2.0), (0.0, 0.0, 1.0))):
        push = camera_push(vrt, (48, 48, 48), res, pos=pos)
        push.cam_dir[:] = list(d) + [0.0]
        _check(vrt, oracle, engine, sc, osn, st, push, ("axis", pos, d))
    # diagonals through lattice points: two or three sideDists equal again and again
    for pos, yaw, pitch in (((0.0, 0.0, 0.0), 45.0, 0.0), ((-8.0, 24.0, -8.0), 45.0, 0.0), ((4.0, 4.0, 4.0), 45.0, 35.264389),
                            ((47.0, 1.0, 0.0), 135.0, 0.0)):
        push = camera_push(vrt, (48, 48, 48), res, pos=pos, yaw=yaw, pitch=pitch)
        _check(vrt, oracle, engine, sc, osn, st, push, ("diagonal", pos, yaw, pitch))
    sc.destroy()


@pytest.mark.parametrize("ao,shadows,bounces", [(0, True, 0), (0, False, 5), (4, True, 5), (2, True, 2)])
def test_secondary_rays(vrt, oracle, engine, ao, shadows, bounces):
    """shadow and bounce rays take the threshold loop under the same bound (AO rays keep their counting loop); partly filled waves"""
    vol = vrt.synthetic.treehouse(64, seed=11)
    sc, osn = _scene(vrt, oracle, engine, vol)
    for res, pos, yaw, pitch, steps in (((96, 64), None, 90.0, 0.0, 512), ((72, 48), (30.2, 40.7, 20.1), 60.0, -20.0, 512),
                                        ((80, 56), (100.0, 80.0, -30.0), 120.0, -25.0, 200), ((64, 64), (32.0, 60.0, 32.0), 0.0, -89.0, 96)):
        st = vrt.VoxelRenderSettings(targetResolution=res)
        st.fsrSetttings.enable = False
        st.occlusionSettings.numSamples = ao
        st.traceSettings.shadows = shadows; st.traceSettings.maxReflections = bounces; st.traceSettings.maxRaySteps = steps
        push = camera_push(vrt, (64, 64, 64), res, pos=pos, yaw=yaw, pitch=pitch, frame=9, jitter=(-0.2, 0.15))
        _check(vrt, oracle, engine, sc, osn, st, push, (ao, shadows, bounces, res))
    sc.destroy()


def test_large_volume_long_rays_take_the_counting_loop(vrt, oracle, engine):
    """384 cells deep, budget 256: the bound of rays along the long axis exceeds the budget (they must run out of it where the
    reference does), rays across it stay below"""
    vol = np.zeros((384, 16, 40), np.uint8)
    vol[380:, :, :] = 4; vol[100:110, 0:3, :] = 6; vol[250, 8:12, 10:30] = 220
    sc, osn = _scene(vrt, oracle, engine, vol)
    res = (80, 48)
    for steps in (256, 400, 1024):
        st = vrt.VoxelRenderSettings.primary_only(res); st.traceSettings.maxRaySteps = steps
        for pos, yaw, pitch in (((20.0, 8.0, -10.0), 90.0, 0.0), ((-30.0, 8.0, 150.0), 20.0, 0.0), ((20.3, 9.1, 5.2), 85.0, 3.0)):
            push = camera_push(vrt, (40, 16, 384), res, pos=pos, yaw=yaw, pitch=pitch)
            exp = _check(vrt, oracle, engine, sc, osn, st, push, (steps, pos))
    assert (exp["hit_id"] != 0).any()
    sc.destroy()


@pytest.mark.parametrize("primary_only", [True, False])
def test_brick_scenes(vrt, oracle, engine, primary_only):
    """brick_march_thresh: the generic two-level march's long runs by threshold at the wave's smallest clearance"""
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.blue_noise_standin(64)
    g, p = vrt.synthetic.sparse_brick_scene(96, 0.06, seed=21)
    for vol in (vrt.synthetic.dense_from_bricks(g, p), vrt.synthetic.treehouse(64, seed=5), vrt.synthetic.floating_cubes(64, seed=8, count=40)):
        D, H, W = vol.shape
        grid, pool = vrt.synthetic.bricks_from_dense(vol)
        sb = vrt.VoxelScene.from_bricks(engine, grid, pool, pal, sky=sky, noise=noise)
        osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
        for res, pos, yaw, pitch, steps in (((96, 64), None, 90.0, 0.0, 512), ((61, 47), (W * 0.45, H * 0.55, D * 0.4), 40.0, 10.0, 512),
                                            ((80, 56), (W * 1.6, H * 0.9, -0.3 * D), 130.0, -15.0, 64), ((64, 40), (float(W // 2), float(H // 2), -8.0), 90.0, 0.0, 33),
                                            ((72, 40), (-20.0, H * 0.5, D * 0.5), 0.0, 0.0, 1500)):
            st = vrt.VoxelRenderSettings.primary_only(res) if primary_only else vrt.VoxelRenderSettings(targetResolution=res)
            st.fsrSetttings.enable = False
            st.traceSettings.maxRaySteps = steps
            push = camera_push(vrt, (W, H, D), res, pos=pos, yaw=yaw, pitch=pitch, frame=4, jitter=(0.2, -0.1))
            _check(vrt, oracle, engine, sb, osn, st, push, ("bricks", vol.shape, res, steps), planes=GB + ["hit_id", "hit_mask", "rays_total"])
        sb.destroy()


def test_random_sweep(vrt, oracle, engine):
    rng = np.random.default_rng(20260)
    pal = metallic_palette(vrt)
    for case in range(60):
        n = int(rng.choice([24, 40, 64]))
        kind = int(rng.integers(0, 3))
        vol = (vrt.synthetic.floating_cubes(n, seed=int(rng.integers(1, 1 << 30)), count=int(rng.integers(1, 80))) if kind == 0 else
               vrt.synthetic.treehouse(n if n != 24 else 32, seed=int(rng.integers(1, 1 << 30))) if kind == 1 else
               (rng.random((n, n, n)) < rng.uniform(0.0, 0.05)).astype(np.uint8) * np.uint8(rng.integers(1, 256)))
        D, H, W = vol.shape
        sky, noise = vrt.synthetic.sky_gradient(int(rng.choice([7, 64])), int(rng.choice([5, 32]))), vrt.synthetic.blue_noise_standin(16)
        sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
        osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
        res = (int(rng.integers(8, 130)), int(rng.integers(8, 90)))
        st = vrt.VoxelRenderSettings(targetResolution=res)
        st.fsrSetttings.enable = False
        st.occlusionSettings.numSamples = int(rng.integers(0, 3))
        st.traceSettings.shadows = bool(rng.integers(0, 2)); st.traceSettings.maxReflections = int(rng.integers(0, 4))
        st.traceSettings.maxRaySteps = int(rng.choice([32, 40, 64, 100, 200, 512, 1024]))
        mode = int(rng.integers(0, 3))
        pos = ((W / 2 + rng.uniform(-1, 1), H / 2 + rng.uniform(-1, 1), -rng.uniform(0.2, 2.0) * D) if mode == 0 else
               tuple(rng.uniform(0, 1, 3) * np.array([W, H, D])) if mode == 1 else
               (float(rng.integers(0, W + 1)), float(rng.integers(0, H + 1)), -float(rng.integers(0, 40))))
        yaw = float(rng.choice([90.0, 0.0, 45.0, rng.uniform(0, 360)])); pitch = float(rng.choice([0.0, 45.0, -30.0, rng.uniform(-89, 89)]))
        push = camera_push(vrt, (W, H, D), res, pos=pos, yaw=yaw, pitch=pitch, frame=int(rng.integers(0, 50)), jitter=(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5))))
        _check(vrt, oracle, engine, sc, osn, st, push, ("sweep", case))
        sc.destroy()
