"""The hand-written look-up loop (vrt_traverse.h trace_df_fast; chosen by the host for AUTO / DF when the budgets are <= 1024
and no hit_voxel plane is asked for) against the oracle and against the general loop (context option fast_loop = 0): primary rays and the
secondary rays of the megakernel / split kernels (partly filled waves), ties, axis-parallel rays, cameras inside the volume
and on lattice points, rays that miss the box, budgets of 1 ... 1024, ragged frame sizes, batches, and a randomised sweep."""
import os

import numpy as np
import pytest

from helpers import camera_push, compare_planes, metallic_palette

pytestmark = pytest.mark.gpu

GB = ["color8", "depth", "motion", "mask8", "position", "normal8"]
PLANES = GB + ["color_f", "hit_id", "hit_mask", "steps_primary", "steps_total", "rays_total"]     # everything but hit_voxel
# without the count planes a launch marches through the fields WITH open cells (rays end where nothing solid is left in their
# octant): what a caller of the product gets; with them, through the fields without -- the loop's own bookkeeping is pinned
PRODUCT = [p for p in PLANES if p not in ("steps_primary", "steps_total")]


def _render(vrt, engine, sc, st, push, fast, planes=PLANES):
    W, H = st.renderResolution()
    gb = vrt.GeometryBuffer(engine, W, H, planes)
    stc, fr = st.to_c(), gb.to_c()
    import ctypes as C
    with engine.options(fast_loop=fast):
        vrt._capi.check(vrt.lib().vrt_render_geometry(engine.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
        engine.synchronize()
    return gb.numpy()


def _check(vrt, oracle, engine, vol, pal, st, push, sky=None, noise=None):
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    fast = _render(vrt, engine, sc, st, push, True)
    slow = _render(vrt, engine, sc, st, push, False)
    exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=PLANES, nthreads=8)
    bad = compare_planes(fast, exp, PLANES)
    assert not bad, ("fast vs oracle", bad)
    assert not compare_planes(slow, exp, PLANES), "general kernel vs oracle"
    for f in (True, False):
        bad = compare_planes(_render(vrt, engine, sc, st, push, f, PRODUCT), exp, PRODUCT)
        assert not bad, ("open cells, fast loop" if f else "open cells, general loop", bad)
    sc.destroy()
    return exp


def test_fast_loop_floating_cubes(vrt, oracle, engine):
    vol = vrt.synthetic.floating_cubes(64, seed=3, count=120)
    for res, pos, yaw, pitch in (((96, 64), None, 90.0, 0.0), ((64, 64), (32.4, 30.2, 31.7), 37.0, 12.0), ((128, 72), (200.0, 90.0, -40.0), 120.0, -20.0),
                                 ((64, 40), (-30.0, -20.0, -50.0), 300.0, 60.0)):
        st = vrt.VoxelRenderSettings.primary_only(res)
        push = camera_push(vrt, (64, 64, 64), res, pos=pos, yaw=yaw, pitch=pitch, frame=7, jitter=(0.25, -0.3))
        exp = _check(vrt, oracle, engine, vol, metallic_palette(vrt), st, push, sky=vrt.synthetic.sky_gradient(64, 32))
    assert True


def test_fast_loop_ties_and_axis_parallel(vrt, oracle, engine):
    vol = np.zeros((16, 16, 16), np.uint8)
    vol[12, :, :] = 5
    vol[8, 4:12, 4:12] = 9
    vol[4, 7:9, 7:9] = 200
    vol[0:12, 11:14, 11:14] = 7
    res = (32, 32)
    st = vrt.VoxelRenderSettings.primary_only(res)
    push = camera_push(vrt, (16, 16, 16), res, pos=(8.0, 8.0, -8.0))
    push.cam_dir[:] = [0.0, 0.0, 1.0, 0.0]                       # exactly axis-aligned: 1/0 = inf deltas on the centre column / row
    exp = _check(vrt, oracle, engine, vol, metallic_palette(vrt), st, push)
    assert np.isin(exp["hit_mask"], [3, 5, 6, 7]).any()
    # from inside, on lattice points, along +x / -y / +z exactly
    for d in ((1.0, 0.0, 0.0), (0.0, -1.0, 0.0), (0.0, 0.0, 1.0)):
        push = camera_push(vrt, (16, 16, 16), res, pos=(2.0, 9.0, 1.0))
        push.cam_dir[:] = list(d) + [0.0]
        _check(vrt, oracle, engine, vol, metallic_palette(vrt), st, push)


@pytest.mark.parametrize("max_steps", [1, 2, 37, 150, 192, 200, 260, 1024])
def test_fast_loop_budgets(vrt, oracle, engine, max_steps):
    vol = np.zeros((200, 24, 24), np.uint8); vol[190:, :, :] = 3
    res = (48, 48)
    st = vrt.VoxelRenderSettings.primary_only(res)
    st.traceSettings.maxRaySteps = max_steps
    push = camera_push(vrt, (24, 24, 200), res, pos=(12.2, 12.4, -5.0))
    _check(vrt, oracle, engine, vol, metallic_palette(vrt), st, push)


def test_fast_loop_long_rays_odd_volume(vrt, oracle, engine):
    """Up to ~700 iterations per ray through a sparse 300 x 40 x 600 volume (the recovery of positions from the sideDist
    travelled since the start of the ray is at its least precise for long rays)."""
    rng = np.random.default_rng(5)
    vol = (rng.random((600, 40, 300)) < 0.0004).astype(np.uint8) * np.uint8(9)
    vol[590:, :, :] = 4
    res = (160, 48)
    st = vrt.VoxelRenderSettings.primary_only(res)
    st.traceSettings.maxRaySteps = 1024
    for pos, yaw, pitch in (((150.3, 20.2, -30.0), 90.0, 0.0), ((-40.0, 20.0, 10.0), 50.0, 2.0), ((5.5, 35.5, 5.5), 60.0, -3.0)):
        push = camera_push(vrt, (300, 40, 600), res, pos=pos, yaw=yaw, pitch=pitch)
        exp = _check(vrt, oracle, engine, vol, metallic_palette(vrt), st, push)
    assert int(exp["steps_primary"].max()) > 400


def test_fast_loop_batch_and_table(vrt, oracle, engine):
    """19 poses in one launch (slots from the table in device memory, rotating XCD regions) == single calls == oracle."""
    import ctypes as C
    vol = vrt.synthetic.treehouse(64, seed=4)
    pal = metallic_palette(vrt)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal)
    osn = oracle.OracleScene(vol, pal)
    res = (200, 120)
    st = vrt.VoxelRenderSettings.primary_only(res)
    pushes = [camera_push(vrt, (64, 64, 64), res, frame=f, pos=(30.0 + 1.5 * f, 31.0 + 0.5 * f, -50.0 + 3.0 * f), yaw=90.0 - 2.0 * f) for f in range(19)]
    stage = vrt.GeometryStage(engine, st, sc)
    batch = [g.numpy() for g in stage.record_batch(pushes)]
    engine.synchronize()
    for f in (0, 7, 18):
        exp = oracle.render(osn, pushes[f], oracle.params_from(st.to_c()), planes=GB, nthreads=8)
        assert not compare_planes(batch[f], exp, GB), f
    sc.destroy()


@pytest.mark.parametrize("split", [False, True])
def test_fast_loop_secondary_rays(vrt, oracle, engine, split):
    """AO 4 x 64, shadow ray, <= 5 bounces through the fast loop: hit lanes only are in EXEC while the secondary rays are
    traced (the loop parks the other lanes), frame sizes that leave partly filled waves."""
    vol = vrt.synthetic.floating_cubes(64, seed=21, count=150)
    sky, noise = vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.blue_noise_standin(64)
    for res, pos in (((100, 60), None), ((61, 47), (30.2, 33.1, 20.4)), ((96, 64), (90.0, 70.0, -30.0))):
        st = vrt.VoxelRenderSettings(targetResolution=res)
        st.fsrSetttings.enable = False
        st.traceSettings.splitKernels = split
        push = camera_push(vrt, (64, 64, 64), res, pos=pos, yaw=90.0 if pos is None else 115.0, frame=11, jitter=(0.1, 0.2))
        exp = _check(vrt, oracle, engine, vol, metallic_palette(vrt), st, push, sky=sky, noise=noise)
        assert int(exp["rays_total"].max()) > 6


def test_fast_loop_random_sweep(vrt, oracle, engine):
    # VRT_SWEEP_CASES / VRT_SWEEP_SEED: longer runs of the same sweep (DESIGN.md quotes one of 20 000 cases)
    rng = np.random.default_rng(int(os.environ.get("VRT_SWEEP_SEED", "2024")))
    for case in range(int(os.environ.get("VRT_SWEEP_CASES", "80"))):
        kind = int(rng.integers(0, 4))
        if kind == 0:   vol = vrt.synthetic.floating_cubes(int(rng.integers(16, 72)), seed=int(rng.integers(1, 1 << 30)), count=int(rng.integers(1, 200)))
        elif kind == 1: vol = vrt.synthetic.sparse_bricks(int(rng.choice([32, 48, 64])), int(rng.choice([2, 4, 8])), float(rng.uniform(0.005, 0.3)), seed=int(rng.integers(1, 1 << 30)))
        elif kind == 2: vol = vrt.synthetic.treehouse(int(rng.choice([32, 64])), seed=int(rng.integers(1, 1 << 30)))
        else:           vol = (rng.random((int(rng.integers(5, 70)), int(rng.integers(5, 70)), int(rng.integers(5, 70)))) < rng.uniform(0.0, 0.1)).astype(np.uint8) * np.uint8(rng.integers(1, 256))
        D, H, W = vol.shape
        res = (int(rng.integers(1, 160)), int(rng.integers(1, 120)))
        st = vrt.VoxelRenderSettings.primary_only(res)
        st.traceSettings.maxRaySteps = int(rng.choice([1, 7, 64, 512, 1000]))
        if case % 2:                                              # secondary rays: AO / shadow / bounces, megakernel or split
            st.occlusionSettings.numSamples = int(rng.integers(0, 5))
            st.traceSettings.shadows = bool(rng.integers(0, 2))
            st.traceSettings.maxReflections = int(rng.integers(0, 6))
            st.traceSettings.aoSteps = int(rng.choice([1, 16, 64]))
            st.traceSettings.splitKernels = bool(rng.integers(0, 4) == 0)
        mode = int(rng.integers(0, 4))
        if mode == 0:   pos = (W / 2 + rng.uniform(-1, 1), H / 2 + rng.uniform(-1, 1), -rng.uniform(0.2, 2.0) * D)
        elif mode == 1: pos = tuple(rng.uniform(0, 1, 3) * np.array([W, H, D]))
        elif mode == 2: pos = (float(rng.integers(0, W + 1)), float(rng.integers(0, H + 1)), -float(rng.integers(0, 40)))
        else:           pos = tuple(rng.uniform(-2, 3, 3) * np.array([W, H, D]))
        yaw = float(rng.choice([90.0, 0.0, 45.0, rng.uniform(0, 360)])); pitch = float(rng.choice([0.0, 45.0, -30.0, rng.uniform(-89, 89)]))
        push = vrt.make_push(vrt.CameraController(position=pos, yaw=yaw, pitch=pitch), (W, H, D), res, frame=int(rng.integers(0, 100)),
                             jitter=(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5))))
        sky, noise = vrt.synthetic.sky_gradient(int(rng.choice([1, 7, 64])), int(rng.choice([1, 5, 32]))), vrt.synthetic.blue_noise_standin(int(rng.choice([1, 16, 64])))
        sc = vrt.VoxelScene.from_dense(engine, vol, metallic_palette(vrt), sky=sky, noise=noise)
        fast = _render(vrt, engine, sc, st, push, True)
        exp = oracle.render(oracle.OracleScene(vol, metallic_palette(vrt), sky=sky, noise=noise), push, oracle.params_from(st.to_c()), planes=PLANES, nthreads=8)
        bad = compare_planes(fast, exp, PLANES)
        assert not bad, (case, kind, vol.shape, res, st.traceSettings.maxRaySteps, pos, yaw, pitch, bad[:2])
        bad = compare_planes(_render(vrt, engine, sc, st, push, True, PRODUCT), exp, PRODUCT)            # the product march: open cells
        assert not bad, ("open cells", case, kind, vol.shape, res, st.traceSettings.maxRaySteps, pos, yaw, pitch, bad[:2])
        sc.destroy()


def test_secondary_ray_loops_agree(vrt, oracle, engine):
    """AO rays spend their own clearance (df_any_loop), shadow and bounce rays prefetch their neighbours' rows: every plane,
    the count planes included, equals what the wave-minimum loop without prefetch gives, and the oracle's."""
    vol = vrt.synthetic.treehouse(64, seed=11)
    vol[20:24, 30:34, 20:24] = 230                                # something metallic to bounce off
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(32, 16), vrt.synthetic.blue_noise_standin(32)
    scenes = {}
    for own, pf in (("1", "1"), ("0", "0"), ("1", "0"), ("0", "1")):
        with engine.options(df_own=int(own), df_prefetch=int(pf)):
            scenes[(own, pf)] = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    for ci, (pos, yaw, pitch, ao, ao_steps, max_steps) in enumerate((((32.3, 30.1, -40.0), 90.0, 0.0, 4, 64, 512), ((10.0, 50.0, 10.0), 45.0, -30.0, 3, 16, 200),
                                                                      ((32.0, 20.0, 32.0), 10.0, 5.0, 5, 1, 37), ((80.0, 70.0, -30.0), 130.0, -25.0, 2, 64, 1000))):
        res = (120, 88)
        st = vrt.VoxelRenderSettings(targetResolution=res)
        st.fsrSetttings.enable = False
        st.occlusionSettings.numSamples = ao; st.traceSettings.aoSteps = ao_steps; st.traceSettings.maxRaySteps = max_steps
        st.traceSettings.shadows = True; st.traceSettings.maxReflections = 3
        push = camera_push(vrt, (64, 64, 64), res, pos=pos, yaw=yaw, pitch=pitch, frame=ci)
        exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=PLANES, nthreads=8)
        for key, sc in scenes.items():
            bad = compare_planes(_render(vrt, engine, sc, st, push, True), exp, PLANES)           # through the fields without open cells
            assert not bad, (key, ci, "counts", bad[:2])
            bad = compare_planes(_render(vrt, engine, sc, st, push, True, PRODUCT), exp, PRODUCT)   # the product march
            assert not bad, (key, ci, "product", bad[:2])
    for sc in scenes.values():
        sc.destroy()
