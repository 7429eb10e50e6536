"""BASELINE configs 4 and 5 (and the maximum-size paths) as parity cases: bands of rows against the oracle, the whole
frame through properties that do not depend on its size -- every traversal mode must give the same planes, the sharded
frame must equal the unsharded one."""
import numpy as np
import pytest

from helpers import compare_planes, metallic_palette

pytestmark = pytest.mark.gpu

GB = ["color8", "depth", "motion", "mask8", "position", "normal8"]


def _check_band(vrt, oracle, gb_np, osn, push, st, r0, r1, names):
    exp = oracle.render_band(osn, push, oracle.params_from(st.to_c()), r0, r1, planes=names, nthreads=8)
    got = {n: gb_np[n][r0:r1] for n in names}
    return compare_planes(got, exp, names)


def test_config4_mandelbulb_4k_bounces(vrt, oracle, engine):
    """Config 4: escape-time Mandelbulb, 3840x2160, max_bounces = 2, ids 200..255 metallic (0.8)."""
    N = 160                                              # the 512^3 original differs only in scale; built in seconds
    vol = vrt.synthetic.mandelbulb(N)
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(256, 128), vrt.synthetic.blue_noise_standin(128)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    res = (3840, 2160)
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.traceSettings.maxReflections = 2
    cam = vrt.CameraController(position=(N * 0.5 + 0.3, N * 0.5 + 0.2, -0.45 * N))
    push = vrt.make_push(cam, (N, N, N), res, frame=5)
    names = GB + ["hit_id", "rays_total"]
    frames = {}
    for trav in ("DF", "DENSE"):
        st.traceSettings.traversal = getattr(vrt, "TRAVERSAL_" + trav)
        gb = vrt.GeometryStage(engine, st, sc, debug_planes=True).record(push)
        engine.synchronize()
        frames[trav] = gb.numpy()
    assert not compare_planes(frames["DF"], frames["DENSE"], names + ["steps_total"])
    g = frames["DF"]
    assert (g["hit_id"] >= 200).mean() > 0.02 and (g["rays_total"] > 6).any()          # metallic hits and their bounces
    for r0 in (0, 1072, 2152):                                                         # top, middle, bottom bands of 8 rows
        assert not _check_band(vrt, oracle, g, osn, push, st, r0, r0 + 8, names)


def test_config5_sparse_bricks_long_budget(vrt, oracle, engine):
    """Config 5: 8^3 bricks, 1.5 % occupied, max_steps = 6144, max_bounces = 4, ao_samples = 4 (frag:80-89 sequence)."""
    N = 256
    vol = vrt.synthetic.sparse_bricks(N, 8, 0.015, seed=5)
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(256, 128), vrt.synthetic.blue_noise_standin(512)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    res = (1920, 1080)
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.traceSettings.maxRaySteps = 6144
    st.traceSettings.maxReflections = 4
    st.occlusionSettings.numSamples = 4
    pos, yaw, pitch = vrt.synthetic.default_camera_for(N, N, N)
    cam = vrt.CameraController(position=(pos[0] + 0.3, pos[1] + 0.2, pos[2]), yaw=yaw, pitch=pitch)
    push = vrt.make_push(cam, (N, N, N), res, frame=17)
    names = GB + ["hit_id", "rays_total", "steps_total"]
    frames = {}
    for trav in ("DF", "BITMASK"):
        st.traceSettings.traversal = getattr(vrt, "TRAVERSAL_" + trav)
        gb = vrt.GeometryStage(engine, st, sc, debug_planes=True).record(push)
        engine.synchronize()
        frames[trav] = gb.numpy()
    assert not compare_planes(frames["DF"], frames["BITMASK"], names)
    g = frames["DF"]
    assert 0.05 < (g["hit_id"] != 0).mean() < 0.95
    for r0 in (200, 536, 900):
        assert not _check_band(vrt, oracle, g, osn, push, st, r0, r0 + 6, names)
    # sharded == unsharded at this size (3 simulated ranks on one GPU)
    st.traceSettings.traversal = vrt.TRAVERSAL_DF
    full = vrt.GeometryBuffer(engine, res[0], res[1])
    stage = vrt.GeometryStage(engine, st, sc)
    merged = {n: np.zeros_like(g[n]) for n in GB}
    for rank in range(3):
        sh = vrt.make_shard(rank, 3, 16)
        part = stage.record(push, sh)
        engine.synchronize()
        pn = part.numpy()
        rows = np.arange(res[1])
        own = ((rows // 16) % 3) == rank
        for n in GB:
            merged[n][own] = pn[n][own]
    assert not compare_planes(merged, g, GB)


def test_volume_past_the_32bit_field_limit(vrt, oracle, engine):
    """Eight clearance fields of an 832^3 volume total 4.6 GiB: the traversal switches to 64-bit field indexing
    (vrt_traverse.h df_small).  Same planes as the dense traversal; bands against the oracle."""
    N = 832
    vol = vrt.synthetic.sparse_bricks(N, 8, 0.004, seed=9)
    pal = metallic_palette(vrt)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal)
    res = (640, 360)
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.traceSettings.maxRaySteps = 3000
    st.traceSettings.maxReflections = 1
    st.occlusionSettings.numSamples = 1
    pos, yaw, pitch = vrt.synthetic.default_camera_for(N, N, N)
    cam = vrt.CameraController(position=(pos[0] + 0.3, pos[1] + 0.2, pos[2]), yaw=yaw, pitch=pitch)
    push = vrt.make_push(cam, (N, N, N), res, frame=2)
    names = GB + ["hit_id", "hit_voxel", "steps_total", "rays_total"]
    frames = {}
    for trav in ("DF", "DENSE"):
        st.traceSettings.traversal = getattr(vrt, "TRAVERSAL_" + trav)
        gb = vrt.GeometryStage(engine, st, sc, debug_planes=True).record(push)
        engine.synchronize()
        frames[trav] = gb.numpy()
    assert not compare_planes(frames["DF"], frames["DENSE"], names)
    g = frames["DF"]
    assert (g["hit_id"] != 0).mean() > 0.05 and int(g["steps_total"].max()) > 832
    osn = oracle.OracleScene(vol, pal)
    for r0 in (40, 176, 300):
        assert not _check_band(vrt, oracle, g, osn, push, st, r0, r0 + 4, names)
    sc.destroy()
