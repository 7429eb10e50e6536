"""BASELINE configs 3, 4 and 5 at their stated workloads (and the maximum-size paths) as parity cases: bands of rows against the oracle, the whole
frame through properties that do not depend on its size -- every traversal mode must give the same planes, the sharded
frame must equal the unsharded one."""
import numpy as np
import pytest

from helpers import camera_push, compare_planes, metallic_palette

pytestmark = pytest.mark.gpu

GB = ["color8", "depth", "motion", "mask8", "position", "normal8"]


def _check_band(vrt, oracle, gb_np, osn, push, st, r0, r1, names):
    exp = oracle.render_band(osn, push, oracle.params_from(st.to_c()), r0, r1, planes=names, nthreads=8)
    got = {n: gb_np[n][r0:r1] for n in names}
    return compare_planes(got, exp, names)


@pytest.fixture(scope="module")
def mandelbulb512(vrt):
    """synthetic:mandelbulb(N=512, power=8, iters=8): ~10 s on 8 threads, generated once per session."""
    return vrt.synthetic.mandelbulb(512)


def test_config3_treehouse_1080p_shadow_denoise(vrt, oracle, engine):
    """BASELINE configs[2] at its stated workload: treehouse 256^3, 1920x1080, primary + shadow ray (ao_samples = 0,
    max_bounces = 0), denoiser iterations 1 and 2.  Whole frame: every traversal mode gives the same G-buffer and the same
    denoised image; bands of rows: G-buffer and denoised rows against the oracle."""
    vol = vrt.synthetic.treehouse(256, seed=2)
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(512, 256), vrt.synthetic.blue_noise_standin(512)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    res = (1920, 1080)
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = 0
    st.traceSettings.maxReflections = 0
    pos, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
    push = vrt.make_push(vrt.CameraController(position=pos, yaw=yaw, pitch=pitch), (256, 256, 256), res)
    names = GB + ["hit_id", "rays_total"]
    frames, den = {}, {}
    for trav in ("DF", "DENSE", "BITMASK", "JUMP", "DFJ"):
        st.traceSettings.traversal = getattr(vrt, "TRAVERSAL_" + trav)
        gb = vrt.GeometryStage(engine, st, sc, debug_planes=True).record(push)
        for it in (1, 2):
            st.denoiserSettings.iterations = it
            den[trav, it] = vrt.DenoiserStage(engine, st).record(gb.color, gb.normal, gb.position).cpu().numpy()
        engine.synchronize()
        frames[trav] = gb.numpy()
    g = frames["DF"]
    for trav in ("DENSE", "BITMASK", "JUMP", "DFJ"):
        assert not compare_planes(frames[trav], g, names), trav
        assert (den[trav, 1] == den["DF", 1]).all() and (den[trav, 2] == den["DF", 2]).all(), trav
    hit = g["hit_id"] != 0
    assert 0.15 < hit.mean() < 0.5 and int(g["rays_total"].max()) == 2 and (g["rays_total"][~hit] == 1).all()   # one shadow ray per hit
    assert (den["DF", 1] != g["color8"]).any() and (den["DF", 2] != den["DF", 1]).any()
    # split kernels (K1 -> records -> K2) give the same frame at full size
    st.traceSettings.traversal = vrt.TRAVERSAL_DF
    st.traceSettings.splitKernels = True
    gs = vrt.GeometryStage(engine, st, sc, debug_planes=True).record(push)
    engine.synchronize()
    assert not compare_planes(gs.numpy(), g, names)
    st.traceSettings.splitKernels = False
    # oracle: G-buffer bands, and the denoised rows of each band (the filter reaches 1 + 3 rows: the oracle denoises the
    # band plus that halo, cropped frames clamp at the crop's edge, so only rows with their full halo inside are compared)
    halo = 4
    for r0, r1 in ((0, 8), (536, 548), (1072, 1080)):
        a, b = max(0, r0 - halo), min(res[1], r1 + halo)
        exp = oracle.render_band(osn, push, oracle.params_from(st.to_c()), a, b, planes=names, nthreads=8)
        assert not compare_planes({n: g[n][a:b] for n in names}, exp, names), (r0, r1)
        for it in (1, 2):
            od = oracle.denoise(exp["color8"], exp["normal8"], exp["position"], iterations=it)
            assert (od[r0 - a:r1 - a] == den["DF", it][r0:r1]).all(), (r0, r1, it)
    sc.destroy()


def test_config4_mandelbulb512_4k_bounces(vrt, oracle, engine, mandelbulb512):
    """BASELINE configs[3] at its stated workload: synthetic:mandelbulb(N=512), 3840x2160, max_bounces = 2, ids 200..255
    metallic (0.8); AO 4 and the shadow ray stay at the reference defaults."""
    N = 512
    vol = mandelbulb512
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
    # the sharded frame (8 simulated ranks, 16-row strips: the layout of the 8-GPU run) equals the unsharded one
    st.traceSettings.traversal = vrt.TRAVERSAL_DF
    stage = vrt.GeometryStage(engine, st, sc)
    merged = {n: np.zeros_like(g[n]) for n in GB}
    rows = np.arange(res[1])
    for rank in range(8):
        part = stage.record(push, vrt.make_shard(rank, 8, 16))
        engine.synchronize()
        pn = part.numpy()
        own = ((rows // 16) % 8) == rank
        for n in GB:
            merged[n][own] = pn[n][own]
    assert not compare_planes(merged, g, GB)
    sc.destroy()


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


def test_count_planes_fields_are_built_on_demand_and_can_be_dropped(vrt, engine):
    """A launch with a count plane marches a second set of clearance fields (without open cells), built on first use: the scene
    grows by that much, VoxelScene.trim() gives it back, the next such launch builds it again with the same counts."""
    vol = vrt.synthetic.treehouse(64, seed=3)
    sc = vrt.VoxelScene.from_dense(engine, vol, metallic_palette(vrt), sky=vrt.synthetic.sky_gradient(32, 16), noise=vrt.synthetic.blue_noise_standin(32))
    st = vrt.VoxelRenderSettings.primary_only((128, 96))
    push = camera_push(vrt, (64, 64, 64), (128, 96))
    base = sc.memory_bytes()
    vrt.GeometryStage(engine, st, sc).record(push); engine.synchronize()
    assert sc.memory_bytes() == base                                  # ordinary launches never build it
    a = vrt.GeometryStage(engine, st, sc, debug_planes=True).record(push); engine.synchronize()
    steps = a.steps_primary.clone()
    grown = sc.memory_bytes()
    assert grown > base + 8 * 66 ** 3
    sc.trim()
    assert sc.memory_bytes() == base
    b = vrt.GeometryStage(engine, st, sc, debug_planes=True).record(push); engine.synchronize()
    assert (b.steps_primary == steps).all() and sc.memory_bytes() == grown
    sc.destroy()
