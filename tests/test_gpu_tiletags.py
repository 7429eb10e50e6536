"""Tile tags (k_tile_tags: the 8x8-pixel blocks no occupied 4^3 cell projects onto write what a miss writes without tracing) and
open cells (a ray ends as a miss where nothing solid is left in its octant): the frames they produce against the frames
without either (context options tile_tags = 0 at launch -- which also switches the sky-texel fast path off --, open_cells = 0 at scene build) and against the oracle -- every traversal mode
(8x8 and 16x16 workgroup tiles), cameras at all kinds of angles and distances, jitter, ragged sizes, batches in kernel
arguments and in the table, sharded launches, split kernels -- and a check that blocks really are skipped."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import camera_push, compare_planes, metallic_palette

pytestmark = pytest.mark.gpu

GB = ["color8", "depth", "motion", "mask8", "position", "normal8"]
PRODUCT = GB + ["color_f", "hit_id", "hit_mask", "rays_total"]


def _render(vrt, engine, sc, st, push, tags, planes=PRODUCT, shard=None, flags=None):
    W, H = st.renderResolution()
    gb = vrt.GeometryBuffer(engine, W, H, planes)
    stc, fr = st.to_c(), gb.to_c()
    if flags is not None:
        stc.flags = flags
    with engine.options(tile_tags=tags, sky_fast=tags):
        vrt._capi.check(vrt.lib().vrt_render_geometry(engine.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr),
                                                      C.byref(shard) if shard is not None else None))
        engine.synchronize()
    return gb.numpy()


CAMERAS = [  # position in units of the volume's size, yaw, pitch
    ((0.5, 0.5, -0.8), 90.0, 0.0), ((0.15, 0.8, -0.25), 70.0, -25.0), ((1.2, 1.2, 1.2), 225.0, -35.0), ((0.5, 1.2, 0.5), 90.0, -89.0),
    ((-0.03, 0.5, 0.4), 10.0, 5.0), ((2.4, 0.55, -2.0), 128.0, -3.0), ((0.5, 0.5, -0.05), 90.0, 0.0), ((0.5, -0.6, 0.5), 90.0, 80.0),
    ((-0.5, 0.3, 0.5), 200.0, 0.0), ((1.02, 1.02, -0.02), 135.0, -20.0), ((0.5, 0.5, -6.0), 90.0, 0.0),
]


@pytest.mark.parametrize("trav", ["AUTO", "DF", "DENSE", "BITMASK", "JUMP", "DFJ"])
def test_tags_and_open_cells_change_nothing(vrt, oracle, engine, trav):
    rng = np.random.default_rng(11)
    vols = [vrt.synthetic.treehouse(48, seed=4), vrt.synthetic.floating_cubes(40, seed=8, count=25),
            (rng.random((36, 20, 52)) < 0.004).astype(np.uint8) * np.uint8(77)]
    for vi, vol in enumerate(vols):
        D, H, W = vol.shape
        pal = metallic_palette(vrt)
        sky, noise = vrt.synthetic.sky_gradient(32, 16), vrt.synthetic.blue_noise_standin(32)
        with engine.options(open_cells=0):
            closed = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
        sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
        osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
        for ci, (p, yaw, pitch) in enumerate(CAMERAS):
            res = [(96, 64), (131, 77), (64, 40)][(ci + vi) % 3]
            st = vrt.VoxelRenderSettings.primary_only(res, getattr(vrt, "TRAVERSAL_" + trav))
            if (ci + vi) % 2:
                st.occlusionSettings.numSamples = 2; st.traceSettings.shadows = True; st.traceSettings.maxReflections = 2
                st.traceSettings.splitKernels = ci % 4 == 1
            push = camera_push(vrt, (W, H, D), res, pos=(p[0] * W, p[1] * H, p[2] * D), yaw=yaw, pitch=pitch, frame=ci,
                               jitter=(0.3, -0.2) if ci % 3 == 0 else (0.0, 0.0))
            ref = _render(vrt, engine, closed, st, push, False)
            for name, scene, tags in (("open cells", sc, False), ("open cells + tile tags", sc, True), ("tile tags", closed, True)):
                bad = compare_planes(_render(vrt, engine, scene, st, push, tags), ref, PRODUCT)
                assert not bad, (name, trav, vi, ci, bad[:2])
            if ci % 4 == 0:
                exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=PRODUCT, nthreads=8)
                assert not compare_planes(ref, exp, PRODUCT), ("oracle", trav, vi, ci)
        sc.destroy(); closed.destroy()


def test_tags_skip_blocks_and_open_cells_end_rays(vrt, engine):
    """The development counters (VRT_FLAG_DEBUG_PLANES: steps_primary = iterations the product march took) show the work
    going away: fewer iterations with open cells, fewer again with tags, the same hits."""
    vol = vrt.synthetic.treehouse(64, seed=2)
    res = (160, 96)
    st = vrt.VoxelRenderSettings.primary_only(res)
    push = camera_push(vrt, (64, 64, 64), res)
    with engine.options(open_cells=0):
        closed = vrt.VoxelScene.from_dense(engine, vol, metallic_palette(vrt))
    sc = vrt.VoxelScene.from_dense(engine, vol, metallic_palette(vrt))
    planes = ["hit_id", "steps_primary", "steps_total", "rays_total"]
    a = _render(vrt, engine, closed, st, push, False, planes, flags=1)
    b = _render(vrt, engine, sc, st, push, False, planes, flags=1)
    c = _render(vrt, engine, sc, st, push, True, planes, flags=1)
    assert (a["hit_id"] == b["hit_id"]).all() and (a["hit_id"] == c["hit_id"]).all()
    sa, sb, scount = (int(x["steps_primary"].astype(np.int64).sum()) for x in (a, b, c))
    assert sb < 0.9 * sa and scount < 0.98 * sb, (sa, sb, scount)
    # the count planes of an ordinary launch still report the reference's iterations, whatever the scene was built with
    d = _render(vrt, engine, sc, st, push, True, planes)
    e = _render(vrt, engine, closed, st, push, False, planes)
    assert (d["steps_primary"] == e["steps_primary"]).all() and (d["steps_total"] == e["steps_total"]).all()
    sc.destroy(); closed.destroy()


@pytest.mark.parametrize("nranks,strip_rows", [(1, 16), (3, 16), (2, 32)])
def test_tags_in_batches_and_sharded_launches(vrt, engine, nranks, strip_rows):
    vol = vrt.synthetic.floating_cubes(48, seed=5, count=30)
    sc = vrt.VoxelScene.from_dense(engine, vol, metallic_palette(vrt), sky=vrt.synthetic.sky_gradient(32, 16))
    res = (120, 100)
    st = vrt.VoxelRenderSettings.primary_only(res)
    for n in (3, 11):                                              # slots in the kernel arguments / in the device table
        pushes = [camera_push(vrt, (48, 48, 48), res, pos=(24.0 + 3.0 * k, 20.0 + k, -50.0 + 4.0 * k), yaw=90.0 - 4.0 * k, pitch=2.0 * k)
                  for k in range(n)]
        got = {}
        for tags in (False, True):
            frames = [np.zeros((res[1], res[0], 4), np.uint8) for _ in range(n)]
            with engine.options(tile_tags=tags, sky_fast=tags):
                for rank in range(nranks):
                    stage = vrt.GeometryStage(engine, st, sc)
                    gbs = stage.prepare_batch(n, vrt.make_shard(rank, nranks, strip_rows))(pushes)
                    engine.synchronize()
                    rows = vrt.distributed.packed_row_map(res[1], rank, nranks, strip_rows)
                    rows = rows[rows >= 0]
                    for k in range(n):
                        frames[k][rows] = gbs[k].color.cpu().numpy()[rows]
            got[tags] = frames
        for k in range(n):
            assert (got[True][k] == got[False][k]).all(), (n, k)
            single = _render(vrt, engine, sc, st, pushes[k], False, ["color8"])["color8"]
            assert (got[True][k] == single).all(), (n, k)
    sc.destroy()


def test_open_bricks_and_brick_tags_change_nothing(vrt, engine):
    """Brick scenes: bit 7 of a coarse byte (no occupied brick left in the octant) ends a ray; the tile tags come from the
    occupied bricks.  Against the dense scene without open cells or tags, and the counters show work going away."""
    vol = vrt.synthetic.treehouse(64, seed=6)
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(32, 16), vrt.synthetic.blue_noise_standin(32)
    grid, pool = vrt.synthetic.bricks_from_dense(vol)
    with engine.options(open_cells=0):
        closed = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
        bclosed = vrt.VoxelScene.from_bricks(engine, grid, pool, pal, sky=sky, noise=noise)
    bsc = vrt.VoxelScene.from_bricks(engine, grid, pool, pal, sky=sky, noise=noise)
    for ci, (p, yaw, pitch) in enumerate(CAMERAS):
        res = [(96, 64), (131, 77)][ci % 2]
        st = vrt.VoxelRenderSettings.primary_only(res)
        if ci % 2:
            st.occlusionSettings.numSamples = 2; st.traceSettings.shadows = True; st.traceSettings.maxReflections = 2
        push = camera_push(vrt, (64, 64, 64), res, pos=(p[0] * 64, p[1] * 64, p[2] * 64), yaw=yaw, pitch=pitch, frame=ci)
        ref = _render(vrt, engine, closed, st, push, False)
        for name, scene, tags in (("bricks", bclosed, False), ("open bricks", bsc, False), ("open bricks + tags", bsc, True)):
            bad = compare_planes(_render(vrt, engine, scene, st, push, tags), ref, PRODUCT)
            assert not bad, (name, ci, bad[:2])
    st = vrt.VoxelRenderSettings.primary_only((160, 96))
    push = camera_push(vrt, (64, 64, 64), (160, 96))
    planes = ["hit_id", "steps_primary", "steps_total", "rays_total"]
    a, b, c = (_render(vrt, engine, s, st, push, t, planes, flags=1) for s, t in ((bclosed, False), (bsc, False), (bsc, True)))
    sa, sb, sc_ = (int(x["steps_primary"].astype(np.int64).sum()) for x in (a, b, c))
    assert sb < sa and sc_ < sb, (sa, sb, sc_)
    d = _render(vrt, engine, bsc, st, push, True, planes)
    e = _render(vrt, engine, closed, st, push, False, planes)
    assert (d["steps_primary"] == e["steps_primary"]).all() and (d["steps_total"] == e["steps_total"]).all()
    for s in (closed, bclosed, bsc):
        s.destroy()


def test_tags_with_arbitrary_camera_bases(vrt, oracle, engine):
    """The push block is the caller's: scaled, skewed and mirrored camera bases, a direction that is not unit length, large
    jitter.  Tags on and off must agree (and agree with the oracle)."""
    vol = vrt.synthetic.floating_cubes(40, seed=21, count=30)
    pal = metallic_palette(vrt)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=vrt.synthetic.sky_gradient(16, 8))
    osn = oracle.OracleScene(vol, pal, sky=vrt.synthetic.sky_gradient(16, 8))
    res = (112, 80)
    st = vrt.VoxelRenderSettings.primary_only(res)
    rng = np.random.default_rng(5)
    for case in range(24):
        push = camera_push(vrt, (40, 40, 40), res, pos=(20.0 + rng.uniform(-30, 30), 20.0 + rng.uniform(-30, 30), -40.0 - rng.uniform(0, 60)),
                           yaw=90.0 + rng.uniform(-25, 25), pitch=rng.uniform(-20, 20), jitter=(rng.uniform(-3, 3), rng.uniform(-3, 3)))
        right = np.array(list(push.cam_right)[:3], np.float32); up = np.array(list(push.cam_up)[:3], np.float32); d = np.array(list(push.cam_dir)[:3], np.float32)
        kind = case % 6
        if kind == 0:   right *= 2.5                               # a wide, anamorphic view
        elif kind == 1: up = up * 0.3 + right * 0.4                # skewed
        elif kind == 2: right = -right                             # mirrored
        elif kind == 3: d *= 7.0                                   # direction not normalised (the shader normalises it)
        elif kind == 4: right *= 0.2; up *= 0.2                    # a long lens
        else:           up = -up * 1.7
        push.cam_right[:3] = [float(x) for x in right]; push.cam_up[:3] = [float(x) for x in up]; push.cam_dir[:3] = [float(x) for x in d]
        a = _render(vrt, engine, sc, st, push, True)
        b = _render(vrt, engine, sc, st, push, False)
        bad = compare_planes(a, b, PRODUCT)
        assert not bad, (case, kind, bad[:2])
        if case % 3 == 0:
            exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=PRODUCT, nthreads=8)
            assert not compare_planes(a, exp, PRODUCT), ("oracle", case)
    sc.destroy()


def test_tags_at_the_edges_of_what_the_bound_admits(vrt, oracle, engine):
    """The tags' projection is used only while its own error bound is below 1.5 px (csrc/vrt_tags.h; tests/test_tile_tag_bound.py
    checks the bound against exact arithmetic).  Here the renders: camera bases whose direction leans into the plane of right and
    up by factors from 1e-1 to 1e-6 (the determinant shrinks, the bound grows until the tags switch themselves off), cameras
    whose plane cuts through or grazes the volume, frames 8 pixels wide and 3840 x 2160 -- tags on, off and the oracle agree."""
    vol = vrt.synthetic.floating_cubes(40, seed=33, count=40)
    pal = metallic_palette(vrt)
    sky = vrt.synthetic.sky_gradient(16, 8)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky)
    osn = oracle.OracleScene(vol, pal, sky=sky)
    planes = GB + ["hit_id"]
    cases = []
    for e in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6):                  # nearly coplanar bases
        push = camera_push(vrt, (40, 40, 40), (104, 72), pos=(21.0, 19.0, -55.0), yaw=88.0, pitch=3.0)
        r = np.array(list(push.cam_right)[:3], np.float64); u = np.array(list(push.cam_up)[:3], np.float64); d = np.array(list(push.cam_dir)[:3], np.float64)
        d2 = 0.8 * r + 0.5 * u + e * d
        push.cam_dir[:3] = [float(x) for x in d2]
        cases.append(("coplanar %g" % e, (104, 72), push))
    for dist in (0.5, 2.0, 6.0, 20.5):                               # the camera plane grazing / cutting the cells
        cases.append(("grazing %g" % dist, (104, 72), camera_push(vrt, (40, 40, 40), (104, 72), pos=(-dist, 20.3, 20.1), yaw=90.0, pitch=0.0)))
        cases.append(("grazing up %g" % dist, (104, 72), camera_push(vrt, (40, 40, 40), (104, 72), pos=(20.2, 40.0 + dist, 19.7), yaw=10.0, pitch=-3.0)))
    cases.append(("8 wide", (8, 120), camera_push(vrt, (40, 40, 40), (8, 120), pos=(20.0, 20.0, -60.0))))
    cases.append(("8 high", (200, 8), camera_push(vrt, (40, 40, 40), (200, 8), pos=(20.0, 20.0, -60.0))))
    cases.append(("4K", (3840, 2160), camera_push(vrt, (40, 40, 40), (3840, 2160), pos=(20.3, 21.1, -70.0), yaw=91.0, pitch=-1.0, jitter=(0.25, -0.4))))
    for name, res, push in cases:
        st = vrt.VoxelRenderSettings.primary_only(res)
        a = _render(vrt, engine, sc, st, push, True, planes)
        b = _render(vrt, engine, sc, st, push, False, planes)
        bad = compare_planes(a, b, planes)
        assert not bad, (name, bad[:2])
        if res[0] * res[1] < 100000:
            exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=planes, nthreads=8)
            assert not compare_planes(a, exp, planes), ("oracle", name)
        else:                                                      # 4K: bands against the oracle
            for r0 in (0, 1076, 2152):
                exp = oracle.render_band(osn, push, oracle.params_from(st.to_c()), r0, r0 + 8, planes=planes, nthreads=8)
                assert not compare_planes({k: v[r0:r0 + 8] for k, v in a.items()}, exp, planes), ("oracle", name, r0)
    sc.destroy()
