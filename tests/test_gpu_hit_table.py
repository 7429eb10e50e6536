"""Launches without secondary rays take a hit's colour from a table (GeomParams::hit_colors: colorHit() of every material and
every one of the 26 normals, made by colorHit() itself whenever the settings or the scene's sky have changed; context option
hit_table): bit for bit what every pixel computes on its own, and what the oracle computes -- across changes of the light,
the ambient intensity and the sky between launches of one context (the table must follow), metallic palettes, ties (edge and
corner normals) and two scenes taking turns."""
import ctypes as C

import numpy as np
import pytest

from helpers import camera_push, compare_planes, metallic_palette

pytestmark = pytest.mark.gpu

GB = ["color8", "depth", "motion", "mask8", "position", "normal8"]


def _render(vrt, engine, sc, st, push, table):
    W, H = st.renderResolution()
    gb = vrt.GeometryBuffer(engine, W, H, GB)
    stc, fr = st.to_c(), gb.to_c()
    with engine.options(hit_table=table):
        vrt._capi.check(vrt.lib().vrt_render_geometry(engine.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
        engine.synchronize()
    return gb.numpy()


def _check(vrt, oracle, engine, sc, osn, st, push, what):
    exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=GB, nthreads=8)
    bad = compare_planes(_render(vrt, engine, sc, st, push, 1), exp, GB)
    assert not bad, (what, "table vs oracle", bad)
    assert not compare_planes(_render(vrt, engine, sc, st, push, 0), exp, GB), (what, "per pixel vs oracle")
    return exp


def test_table_follows_settings_sky_and_scene(vrt, oracle, engine):
    pal = metallic_palette(vrt)
    vol_a = vrt.synthetic.treehouse(64, seed=4)
    vol_b = vrt.synthetic.floating_cubes(48, seed=6, count=60)
    skies = [vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.sky_gradient(7, 5)[::-1].copy(), vrt.synthetic.sky_gradient(128, 64) * np.float32(0.5)]
    sa = vrt.VoxelScene.from_dense(engine, vol_a, pal, sky=skies[0])
    sb = vrt.VoxelScene.from_dense(engine, vol_b, vrt.synthetic.default_palette(), sky=skies[1])
    res = (96, 64)
    st = vrt.VoxelRenderSettings.primary_only(res)
    pa = camera_push(vrt, (64, 64, 64), res, yaw=70.0, pitch=-10.0, frame=2, jitter=(0.1, -0.2))
    pb = camera_push(vrt, (48, 48, 48), res, pos=(70.0, 60.0, -30.0), yaw=120.0, pitch=-25.0)
    seen = set()
    for step in range(8):
        # one thing changes per step: the light, the ambient intensity, the sky of scene a, the scene
        if step == 1: st.lightSettings.direction = (0.3, -0.8, 0.52)
        if step == 2: st.lightSettings.intensity = 1.7
        if step == 3: st.occlusionSettings.intensity = 0.35      # (ambient_intensity: the factor on the sky term of a hit)
        if step == 4: sa.set_sky(skies[2])
        if step == 5: st.lightSettings.color = (0.9, 0.5, 0.2, 1.0)
        if step == 6: sa.set_sky(skies[1])
        sky_now = skies[0] if step < 4 else (skies[2] if step < 6 else skies[1])
        osa = oracle.OracleScene(vol_a, pal, sky=sky_now)
        osb = oracle.OracleScene(vol_b, vrt.synthetic.default_palette(), sky=skies[1])
        ea = _check(vrt, oracle, engine, sa, osa, st, pa, ("a", step))
        eb = _check(vrt, oracle, engine, sb, osb, st, pb, ("b", step))
        seen.add(ea["color8"].tobytes()); seen.add(eb["color8"].tobytes())
    assert len(seen) >= 8                                      # the changes did change the pictures
    sa.destroy(); sb.destroy()


def test_edge_and_corner_normals(vrt, oracle, engine):
    """cameras on lattice points looking along diagonals: hits whose mask has two or three bits (normals with components
    1 / sqrt 2 and 1 / sqrt 3), and rays that start inside a solid voxel (rule A: the zero normal, which has no table entry)"""
    vol = np.zeros((32, 32, 32), np.uint8)
    vol[20:, :, :] = 5; vol[:, 24:, :] = 9; vol[8:12, 8:12, 8:12] = 201; vol[0:3, 0:3, 0:3] = 77
    pal = metallic_palette(vrt)
    sky = vrt.synthetic.sky_gradient(64, 32)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky)
    osn = oracle.OracleScene(vol, pal, sky=sky)
    res = (64, 48)
    st = vrt.VoxelRenderSettings.primary_only(res)
    masks = set()
    for pos, yaw, pitch in (((0.0, 0.0, -8.0), 45.0, 0.0), ((-6.0, -6.0, -6.0), 45.0, 35.264389), ((4.0, 4.0, 4.0), 45.0, 35.264389),
                            ((1.5, 1.5, 1.5), 30.0, 10.0), ((16.0, 16.0, -10.0), 90.0, 0.0), ((31.0, 0.0, 0.0), 135.0, 0.0)):
        push = camera_push(vrt, (32, 32, 32), res, pos=pos, yaw=yaw, pitch=pitch)
        exp = _check(vrt, oracle, engine, sc, osn, st, push, (pos, yaw, pitch))
        n = exp["normal8"].reshape(-1, exp["normal8"].shape[-1])[:, :3]
        masks.update(int((row != 0).sum()) for row in np.unique(n, axis=0))
    assert {1, 2}.issubset(masks) or {1, 3}.issubset(masks)
    sc.destroy()
