"""Parity of the HIP path (through the C-ABI) with the CPU oracle on the same seeded inputs.

Bar: every G-buffer plane, every debug plane and the fp32 pre-quantisation colour are BIT-EXACT
(the numeric spec pins atan/asin/exp/normalize and forbids FMA contraction, so no tolerance is needed);
hit voxel ids / cells / masks are integers and trivially so.  Tolerance stated per test: 0 ulp.
"""
import os

import numpy as np
import pytest

from helpers import camera_push, compare_planes, metallic_palette

pytestmark = pytest.mark.gpu

GB = ["color8", "depth", "motion", "mask8", "position", "normal8"]
DBG = ["color_f", "hit_id", "hit_voxel", "hit_mask", "steps_primary", "steps_total", "rays_total"]


def _scene_pair(vrt, oracle, engine, vol, pal, sky=None, noise=None):
    sky = sky if sky is not None else vrt.synthetic.sky_gradient(64, 32)
    noise = noise if noise is not None else vrt.synthetic.blue_noise_standin(64)
    gs = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    return gs, osn


def _render_both(vrt, oracle, engine, gs, osn, settings, push, shard=None):
    stage = vrt.GeometryStage(engine, settings, gs, debug_planes=True)
    gb = stage.record(push, shard)
    engine.synchronize()
    got = gb.numpy()
    exp = oracle.render(osn, push, oracle.params_from(settings.to_c()), nthreads=8)
    return got, exp


@pytest.mark.parametrize("trav", ["DENSE", "BITMASK", "JUMP", "DF", "DFJ"])
@pytest.mark.parametrize("mode", ["primary_only", "shadow_only", "default_ao_shadow_bounce"])
def test_geometry_bit_exact_floating_cubes(vrt, oracle, engine, trav, mode):
    vol = vrt.synthetic.floating_cubes(64, seed=1, count=120)
    pal = metallic_palette(vrt)
    gs, osn = _scene_pair(vrt, oracle, engine, vol, pal)
    res = (96, 64)
    st = vrt.VoxelRenderSettings.primary_only(res, getattr(vrt, "TRAVERSAL_" + trav))
    if mode == "shadow_only":
        st.traceSettings.shadows = True
    elif mode == "default_ao_shadow_bounce":
        st.traceSettings.shadows = True
        st.traceSettings.maxReflections = 5
        st.occlusionSettings.numSamples = 4
    push = camera_push(vrt, (64, 64, 64), res, frame=3)
    got, exp = _render_both(vrt, oracle, engine, gs, osn, st, push)
    names = GB + DBG if trav not in ("JUMP", "DFJ") else GB + ["color_f", "hit_id", "hit_voxel", "hit_mask", "rays_total"]
    bad = compare_planes(got, exp, names)
    assert not bad, bad
    assert (exp["hit_id"] != 0).mean() > 0.2 and (exp["hit_id"] == 0).mean() > 0.05     # both cases exercised
    if mode == "default_ao_shadow_bounce":
        assert (exp["rays_total"] > 6).any()                                             # some metallic bounces


@pytest.mark.parametrize("trav", ["DENSE", "BITMASK", "JUMP", "DF", "DFJ"])
def test_geometry_bit_exact_odd_sizes_and_views(vrt, oracle, engine, trav):
    # non-multiple-of-4 volume, ragged frame (not a multiple of the 16x16 tile), cameras inside / above / tilted
    rng = np.random.default_rng(12)
    vol = (rng.random((37, 22, 51)) < 0.04).astype(np.uint8) * rng.integers(1, 256, (37, 22, 51)).astype(np.uint8)
    pal = metallic_palette(vrt, ids=range(128, 256))
    gs, osn = _scene_pair(vrt, oracle, engine, vol, pal)
    res = (45, 37)
    for pos, yaw, pitch in [((25.3, 11.2, -30.0), 90.0, 0.0), ((25.3, 11.2, 18.4), 40.0, -20.0),
                            ((-20.0, 40.0, -10.0), 30.0, -35.0), ((25.5, 60.0, 18.5), 90.0, -89.0),
                            ((80.0, 11.0, 18.0), 180.0, 0.0)]:
        st = vrt.VoxelRenderSettings.primary_only(res, getattr(vrt, "TRAVERSAL_" + trav))
        st.traceSettings.shadows = True
        st.occlusionSettings.numSamples = 2
        st.traceSettings.maxReflections = 3
        push = camera_push(vrt, (51, 22, 37), res, pos=pos, yaw=yaw, pitch=pitch, frame=33, jitter=(0.25, -0.4))
        got, exp = _render_both(vrt, oracle, engine, gs, osn, st, push)
        names = GB + DBG if trav not in ("JUMP", "DFJ") else GB + ["color_f", "hit_id", "hit_voxel", "hit_mask", "rays_total"]
        bad = compare_planes(got, exp, names)
        assert not bad, (pos, bad)


@pytest.mark.parametrize("trav", ["DENSE", "BITMASK", "JUMP", "DF", "DFJ"])
def test_exact_ties_and_axis_parallel_rays(vrt, oracle, engine, trav):
    # camera on a lattice point looking down an axis: centre rays are axis-parallel (1/0 = inf deltas),
    # diagonal pixels hit exact sideDist ties (multi-axis masks, diagonal normals)
    vol = np.zeros((16, 16, 16), np.uint8)
    vol[12, :, :] = 5
    vol[8, 4:12, 4:12] = 9
    vol[4, 7:9, 7:9] = 200
    vol[0:12, 11:14, 11:14] = 7          # pillar on the x = y diagonal: entered by a simultaneous x+y step
    pal = metallic_palette(vrt)
    gs, osn = _scene_pair(vrt, oracle, engine, vol, pal)
    res = (32, 32)
    st = vrt.VoxelRenderSettings.primary_only(res, getattr(vrt, "TRAVERSAL_" + trav))
    st.traceSettings.shadows = True
    st.traceSettings.maxReflections = 2
    push = camera_push(vrt, (16, 16, 16), res, pos=(8.0, 8.0, -8.0))
    push.cam_dir[:] = [0.0, 0.0, 1.0, 0.0]          # exactly axis-aligned (cos(radians(90)) is -4.4e-8 in fp32)
    got, exp = _render_both(vrt, oracle, engine, gs, osn, st, push)
    names = GB + ["color_f", "hit_id", "hit_voxel", "hit_mask", "rays_total"]
    bad = compare_planes(got, exp, names)
    assert not bad, bad
    assert np.isin(exp["hit_mask"], [3, 5, 6, 7]).any()          # tie masks occurred


@pytest.mark.parametrize("trav", ["DENSE", "BITMASK", "JUMP", "DF", "DFJ"])
def test_step_budget_exhaustion(vrt, oracle, engine, trav):
    # a wall that most rays reach only after more DDA iterations than max_steps allows
    vol = np.zeros((200, 24, 24), np.uint8); vol[190:, :, :] = 3
    pal = metallic_palette(vrt)
    gs, osn = _scene_pair(vrt, oracle, engine, vol, pal)
    res = (48, 48)
    for max_steps in (150, 192, 200, 260):
        st = vrt.VoxelRenderSettings.primary_only(res, getattr(vrt, "TRAVERSAL_" + trav))
        st.traceSettings.maxRaySteps = max_steps
        push = camera_push(vrt, (24, 24, 200), res, pos=(12.2, 12.4, -5.0))
        got, exp = _render_both(vrt, oracle, engine, gs, osn, st, push)
        bad = compare_planes(got, exp, GB + ["color_f", "hit_id", "hit_voxel", "hit_mask"])
        assert not bad, (max_steps, bad)


def test_split_kernels_equal_megakernel(vrt, oracle, engine):
    """K1 -> hit records -> K2 over the compacted hit list (VRT_FLAG_SPLIT_KERNELS) == the default megakernel."""
    vol = vrt.synthetic.floating_cubes(64, seed=21, count=150)
    pal = metallic_palette(vrt)
    gs, osn = _scene_pair(vrt, oracle, engine, vol, pal)
    res = (130, 90)
    outs = []
    for split in (False, True):
        st = vrt.VoxelRenderSettings(targetResolution=res)
        st.fsrSetttings.enable = False
        st.traceSettings.splitKernels = split
        push = camera_push(vrt, (64, 64, 64), res, frame=7)
        got, exp = _render_both(vrt, oracle, engine, gs, osn, st, push)
        assert not compare_planes(got, exp, GB + DBG), split
        outs.append(got)
    assert not compare_planes(outs[0], outs[1], GB + DBG)


def test_treehouse_1080p_properties(vrt, oracle, engine):
    """BASELINE config 2 at full size: size-independent properties + oracle parity on sampled rows."""
    vol = vrt.synthetic.treehouse(256, seed=2)
    pal = metallic_palette(vrt)
    gs, osn = _scene_pair(vrt, oracle, engine, vol, pal, sky=vrt.synthetic.sky_gradient(512, 256),
                          noise=vrt.synthetic.blue_noise_standin(512))
    res = (1920, 1080)
    pos, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
    push = camera_push(vrt, (256, 256, 256), res, pos=pos, yaw=yaw, pitch=pitch)
    outs = {}
    for trav in ("DENSE", "BITMASK", "JUMP", "DF", "DFJ"):
        st = vrt.VoxelRenderSettings.primary_only(res, getattr(vrt, "TRAVERSAL_" + trav))
        stage = vrt.GeometryStage(engine, st, gs, debug_planes=True)
        gb = stage.record(push)
        engine.synchronize()
        outs[trav] = gb.numpy()
    o = outs["DENSE"]
    # traversal modes agree bit-for-bit at full size
    for trav in ("BITMASK", "JUMP", "DF", "DFJ"):
        names = GB + ["color_f", "hit_id", "hit_voxel", "hit_mask"] + (["steps_primary"] if trav not in ("JUMP", "DFJ") else [])
        assert not compare_planes(outs[trav], o, names), trav
    # the hit cell really holds the reported id; misses report 0 everywhere
    hit = o["hit_id"] != 0
    hv = o["hit_voxel"].astype(np.int64)
    assert (vol[hv[..., 2][hit], hv[..., 1][hit], hv[..., 0][hit]] == o["hit_id"][hit]).all()
    assert (o["depth"][~hit] == 0).all() and (o["mask8"][~hit] == 0).all() and (o["mask8"][hit] == 230).all()
    assert 0.15 < hit.mean() < 0.99
    # hit position lies on the reported face of the reported cell
    # (box-entry voxels that are solid are the canonical rule-A pixels: d is not a face distance there)
    # and multi-axis tie masks make d = length(mask*(side-delta)) a diagonal length, not a ray distance (frag:191)
    face = hit & (o["steps_primary"] > 1) & np.isin(o["hit_mask"], [1, 2, 4])
    p = o["position"][..., :3][face]
    lo, hi = hv[face].astype(np.float32) - 2e-2, hv[face].astype(np.float32) + 1 + 2e-2   # sideDist sums up to 512 fp32 additions
    assert ((p >= lo) & (p <= hi)).all()
    # oracle parity on a band of rows (the oracle renders 40 rows in a second or two)
    exp = oracle.render(osn, push, oracle.params_from(st.to_c()), rows=(520, 560))
    sub = {k: v[520:560] for k, v in o.items()}
    esub = {k: v[520:560] for k, v in exp.items()}
    assert not compare_planes(sub, esub, GB + DBG)


def test_vox_file_load_matches_from_dense(vrt, oracle, engine, tmp_path):
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    data = open(os.path.join(gold, "vox_multi.vox"), "rb").read()
    exp = np.load(os.path.join(gold, "vox_multi.npz"))
    p = tmp_path / "multi.vox"
    p.write_bytes(data)
    sc = vrt.VoxelScene(engine, str(p))
    assert (sc.width, sc.height, sc.depth) == exp["voxels"].shape[::-1]
    vox, pal = sc.download()
    assert (vox == exp["voxels"]).all() and np.allclose(pal, exp["palette"], rtol=1e-6)
    # and it renders identically to the oracle fed with the reference parser's volume
    osn = oracle.OracleScene(exp["voxels"], pal)
    res = (40, 24)
    st = vrt.VoxelRenderSettings.primary_only(res, vrt.TRAVERSAL_BITMASK)     # exact step counts (JUMP reports bounds)
    W, H, D = sc.width, sc.height, sc.depth
    push = camera_push(vrt, (W, H, D), res, pos=(W / 2 + 0.3, H / 2 + 0.1, -1.2 * D))
    stage = vrt.GeometryStage(engine, st, sc, debug_planes=True)
    gb = stage.record(push); engine.synchronize()
    e = oracle.render(osn, push, oracle.params_from(st.to_c()))
    assert not compare_planes(gb.numpy(), e, GB + DBG)
    # error behaviour mirrors the reference's exceptions (voxel_scene.cpp:42,46,50)
    with pytest.raises(RuntimeError, match="Failed to read voxel scene"):
        vrt.VoxelScene(engine, str(tmp_path / "missing.vox"))
    bad = tmp_path / "bad.vox"; bad.write_bytes(b"VOY " + data[4:])
    with pytest.raises(RuntimeError, match="Could not parse voxel scene"):
        vrt.VoxelScene(engine, str(bad))
    with pytest.raises(RuntimeError, match="does not contain an instance"):
        vrt.VoxelScene.from_memory(engine, open(os.path.join(gold, "vox_err_no_instance.vox"), "rb").read())


def test_sky_and_noise_from_files(vrt, oracle, engine, tmp_path):
    """Texture2D path: sky from a Radiance .hdr and blue noise from a PNG, decoded by the library."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_image_io import _rgbe_encode, _hdr_file, _expected_float
    from PIL import Image
    rng = np.random.default_rng(8)
    rgbe = _rgbe_encode(rng.random((16, 32, 3)) * 4.0)
    (tmp_path / "sky.hdr").write_bytes(_hdr_file(rgbe, True))
    noise = rng.integers(0, 256, (64, 64, 4), dtype=np.uint8)
    Image.fromarray(noise, "RGBA").save(tmp_path / "noise.png")
    vol = vrt.synthetic.floating_cubes(32, seed=5, count=40)
    pal = metallic_palette(vrt)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal)
    sc.set_sky(str(tmp_path / "sky.hdr")); sc.set_blue_noise(str(tmp_path / "noise.png"))
    osn = oracle.OracleScene(vol, pal, sky=_expected_float(rgbe), noise=noise)
    res = (64, 48)
    st = vrt.VoxelRenderSettings(targetResolution=res); st.fsrSetttings.enable = False
    push = camera_push(vrt, (32, 32, 32), res, frame=2)
    stage = vrt.GeometryStage(engine, st, sc, debug_planes=True)
    gb = stage.record(push); engine.synchronize()
    exp = oracle.render(osn, push, oracle.params_from(st.to_c()), nthreads=4)
    assert not compare_planes(gb.numpy(), exp, GB + DBG)
    with pytest.raises(RuntimeError, match="Could not load image"):
        sc.set_sky(str(tmp_path / "nope.hdr"))


def test_argument_validation(vrt, engine):
    vol = vrt.synthetic.single_voxel()
    sc = vrt.VoxelScene.from_dense(engine, vol, vrt.synthetic.default_palette())
    st = vrt.VoxelRenderSettings.primary_only((16, 16))
    stage = vrt.GeometryStage(engine, st, sc)
    push = camera_push(vrt, (9, 8, 8), (16, 16))                 # wrong volume bounds
    with pytest.raises(vrt.VrtError, match="volume_bounds"):
        stage.record(push)
    st.traceSettings.maxReflections = 9
    with pytest.raises(vrt.VrtError, match="max_bounces"):
        stage.record(camera_push(vrt, (8, 8, 8), (16, 16)))


@pytest.mark.parametrize("res", [(1, 1), (7, 5), (9, 17), (65, 3), (8, 8), (130, 1)])
def test_tiny_and_ragged_frames(vrt, oracle, engine, res):
    """Frames smaller than a tile, one pixel wide / high, and sizes that leave partial tiles on both edges: geometry and
    the denoiser (whose LDS tile then hangs over the frame on every side) against the oracle."""
    vol = vrt.synthetic.floating_cubes(32, seed=8, count=40)
    pal = metallic_palette(vrt)
    gs, osn = _scene_pair(vrt, oracle, engine, vol, pal)
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = 2
    st.denoiserSettings.iterations = 3
    push = camera_push(vrt, (32, 32, 32), res, frame=1)
    got, exp = _render_both(vrt, oracle, engine, gs, osn, st, push)
    assert not compare_planes(got, exp, GB + DBG)
    stage = vrt.GeometryStage(engine, st, gs)
    gb = stage.record(push)
    den = vrt.DenoiserStage(engine, st).record(gb.color, gb.normal, gb.position)
    engine.synchronize()
    e = oracle.denoise(exp["color8"], exp["normal8"], exp["position"], iterations=3)
    assert (den.cpu().numpy() == e).all()


def test_host_pointers_are_refused(vrt, engine):
    """A host pointer in vrt_frame must come back as an error code, not as a GPU fault."""
    import ctypes as C
    vol = vrt.synthetic.floating_cubes(16, seed=1, count=5)
    sc = vrt.VoxelScene.from_dense(engine, vol, vrt.synthetic.default_palette())
    st = vrt.VoxelRenderSettings.primary_only((32, 16))
    push = camera_push(vrt, (16, 16, 16), (32, 16))
    host = np.zeros((16, 32, 4), np.uint8)
    fr = vrt._capi.Frame()
    fr.color8 = host.ctypes.data
    stc = st.to_c()
    rc = vrt.lib().vrt_render_geometry(engine.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None)
    assert rc != 0 and b"not device memory" in vrt.lib().vrt_last_error()


@pytest.mark.parametrize("mode", ["primary_only", "default", "split"])
def test_batch_of_frames_equals_single_calls(vrt, oracle, engine, mode):
    """vrt_render_geometry_batch: 11 poses (one launch, slots read from the table in device memory; one launch per frame in
    split mode) must give exactly the planes of 11 single calls; three of
    them are also checked against the oracle.  Sharded: the strips of a simulated rank only."""
    vol = vrt.synthetic.treehouse(64, seed=4)
    pal = metallic_palette(vrt)
    gs, osn = _scene_pair(vrt, oracle, engine, vol, pal)
    res = (200, 120)
    st = vrt.VoxelRenderSettings.primary_only(res) if mode == "primary_only" else vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.traceSettings.splitKernels = mode == "split"
    pushes = [camera_push(vrt, (64, 64, 64), res, frame=f, pos=(30.0 + 1.5 * f, 31.0 + 0.5 * f, -50.0 + 3.0 * f), yaw=90.0 - 2.0 * f,
                          jitter=(0.1 * f - 0.5, 0.25)) for f in range(11)]
    stage = vrt.GeometryStage(engine, st, gs, debug_planes=True)
    names = GB + ["color_f", "hit_id", "hit_voxel", "hit_mask", "steps_primary", "steps_total", "rays_total"]
    for shard in (None, vrt.make_shard(1, 3, 16)):
        batch = [g.numpy() for g in stage.record_batch(pushes, shard)]
        engine.synchronize()
        for f, push in enumerate(pushes):
            single = stage.record(push, shard)
            engine.synchronize()
            sn = single.numpy()
            own = np.ones(res[1], bool) if shard is None else ((np.arange(res[1]) // 16) % 3) == 1     # a rank writes its strips only
            bad = compare_planes({n: batch[f][n][own] for n in names}, {n: sn[n][own] for n in names}, names)
            assert not bad, (f, shard is not None, bad)
        if shard is None:
            for f in (0, 7, 10):
                exp = oracle.render(osn, pushes[f], oracle.params_from(st.to_c()), nthreads=8)
                assert not compare_planes(batch[f], exp, names), f
    assert any((b["hit_id"] != batch[0]["hit_id"]).any() for b in batch[1:])           # the poses really differ


@pytest.mark.parametrize("n", [5, 19])
def test_frames_with_their_own_strip_assignment_equal_single_calls(vrt, engine, n):
    """vrt_render_geometry_slots: every frame of the launch plays another rank of a 3-rank split of 136 rows (8.5 strips:
    the ranks own 48, 48 and 40 rows); 5 frames travel in the kernel arguments, 19 in the device table.  The launch is
    repeated more often than the table ring is long, with the poses in another order each time."""
    vol = vrt.synthetic.treehouse(64, seed=4)
    gs = vrt.VoxelScene.from_dense(engine, vol, metallic_palette(vrt), sky=vrt.synthetic.sky_gradient(64, 32), noise=vrt.synthetic.blue_noise_standin(64))
    res = (200, 136)
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.traceSettings.maxReflections = 1
    pushes = [camera_push(vrt, (64, 64, 64), res, frame=f, pos=(30.0 + 1.5 * f, 31.0 + 0.5 * f, -50.0 + 3.0 * f), yaw=90.0 - 2.0 * f)
              for f in range(n)]
    ranks = [(2 * f + 1) % 3 for f in range(n)]
    stage = vrt.GeometryStage(engine, st, gs)
    launch = stage.prepare_batch(n, shards=[vrt.make_shard(r, 3, 16) for r in ranks])
    names = GB
    singles = {}
    for rep in range(6):
        order = [(f + rep) % n for f in range(n)]               # slot k renders pose order[k] with the assignment of slot k
        batch = [g.numpy() for g in launch([pushes[f] for f in order])]
        engine.synchronize()
        if rep not in (0, 5):
            continue
        for k, f in enumerate(order):
            if (f, ranks[k]) not in singles:
                singles[(f, ranks[k])] = stage.record(pushes[f], vrt.make_shard(ranks[k], 3, 16)).numpy()
                engine.synchronize()
            sn = singles[(f, ranks[k])]
            own = ((np.arange(res[1]) // 16) % 3) == ranks[k]
            bad = compare_planes({m: batch[k][m][own] for m in names}, {m: sn[m][own] for m in names}, names)
            assert not bad, (rep, k, f, bad)
    with pytest.raises(ValueError):
        stage.prepare_batch(n, shards=[vrt.make_shard(0, 3, 16)] * (n - 1))
    bad_mix = [vrt.make_shard(0, 3, 16)] * (n - 1) + [vrt.make_shard(0, 2, 16)]
    with pytest.raises(vrt.VrtError):
        stage.prepare_batch(n, shards=bad_mix)(pushes)


def test_two_contexts_share_a_scene(vrt, oracle, engine):
    """One context per frame in flight (MAX_FRAMES_IN_FLIGHT = 2, engine.hpp:19): two contexts with their own streams render
    alternate poses of one scene concurrently; every frame must be the oracle's."""
    import torch
    vol = vrt.synthetic.treehouse(48, seed=9)
    pal = metallic_palette(vrt)
    gs, osn = _scene_pair(vrt, oracle, engine, vol, pal)
    engine.synchronize()
    res = (160, 96)
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = 2
    slots = [vrt.Engine(0, use_torch_stream=False) for _ in range(2)]
    torch.cuda.synchronize()
    stages = [vrt.GeometryStage(e, st, gs) for e in slots]
    pushes = [camera_push(vrt, (48, 48, 48), res, frame=f, pos=(22.0 + f, 25.0, -40.0 + 1.5 * f)) for f in range(6)]
    torch.cuda.synchronize()                                   # the planes were zero-filled on torch's stream
    outs = []
    for f, push in enumerate(pushes):
        e = slots[f % 2]
        if f >= 2:
            e.synchronize()                                    # the slot's previous frame is about to be overwritten
            outs.append(stages[f % 2]._buffer.numpy())
        stages[f % 2].record(push)
    for k in (0, 1):
        slots[(len(pushes) + k) % 2].synchronize()
        outs.append(stages[(len(pushes) + k) % 2]._buffer.numpy())
    for f, push in enumerate(pushes):
        exp = oracle.render(osn, push, oracle.params_from(st.to_c()), nthreads=8)
        assert not compare_planes(outs[f], exp, GB), f
    for e in slots:
        e.destroy()
