"""Round 4: (1) the megakernel's bounce chain as one word per hit (k_primary MODE 5 / 6 / 7, color_main_ray_packed; context option
packed_bounces) against the stack of hits it replaces and against the oracle -- every plane, the count planes included (the
secondary rays of a metallic bounce are traced before it is known whether the chain ends; where it does not, they must leave the
counts again: voxel_volume.frag:281-303 with lastIdx = -1); (2) the counting twins of the look-up loops (VRT_FLAG_MARCHED_COUNTS,
VRT_FLAG_LOOKUP_COUNTS): the same frame as the product launch, iteration counts that equal the merged loops' where no two axes
tie, and look-up counts between their obvious bounds."""
import ctypes as C

import numpy as np
import pytest

from helpers import camera_push, compare_planes, metallic_palette

pytestmark = pytest.mark.gpu

GB = ["color8", "depth", "motion", "mask8", "position", "normal8"]
ALL = GB + ["color_f", "hit_id", "hit_mask", "steps_primary", "steps_total", "rays_total"]


def _render(vrt, engine, sc, st, push, planes, flags=0, **opts):
    W, H = st.renderResolution()
    gb = vrt.GeometryBuffer(engine, W, H, planes)
    stc, fr = st.to_c(), gb.to_c()
    stc.flags |= flags
    with engine.options(**opts):
        vrt._capi.check(vrt.lib().vrt_render_geometry(engine.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
        engine.synchronize()
    return gb.numpy()


def _mirror_hall(n=40):
    """a hall whose walls, floor and pillars are all metallic (ids >= 200) with a few matt blocks: chains of every length,
    chains that end in the sky through the open roof, and chains of max_bounces metallic hits"""
    vol = np.zeros((n, n, n), np.uint8)
    vol[:, 0, :] = 210; vol[:, :, 0] = 220; vol[:, :, n - 1] = 230; vol[0, :, :] = 240; vol[n - 1, :, :] = 250      # (z, y, x) order
    vol[8:30, 1:20, 12:15] = 205; vol[10:14, 1:8, 24:30] = 7; vol[25:28, 1:25, 25:28] = 215; vol[18:22, 1:5, 5:9] = 9
    return vol


@pytest.mark.parametrize("bounces", [0, 1, 2, 3, 5, 8])
def test_packed_chain_equals_stack_and_oracle(vrt, oracle, engine, bounces):
    vol = _mirror_hall()
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.blue_noise_standin(64)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    res = (72, 56)
    for ao, sh, pos, yaw, pitch in ((2, True, (20.3, 30.2, 20.4), 40.0, -35.0), (0, True, (5.5, 12.5, 33.1), -20.0, -10.0),
                                    (3, False, (33.2, 8.4, 6.3), 130.0, 5.0), (4, True, (20.0, 60.0, 20.0), 90.0, -89.0)):
        st = vrt.VoxelRenderSettings(targetResolution=res)
        st.fsrSetttings.enable = False
        st.occlusionSettings.numSamples = ao
        st.traceSettings.shadows = sh
        st.traceSettings.maxReflections = bounces
        push = camera_push(vrt, (40, 40, 40), res, pos=pos, yaw=yaw, pitch=pitch, frame=11)
        exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=ALL, nthreads=8)
        packed = _render(vrt, engine, sc, st, push, ALL, packed_bounces=1)
        stack = _render(vrt, engine, sc, st, push, ALL, packed_bounces=0)
        assert not compare_planes(packed, exp, ALL), ("packed chain vs oracle", bounces, ao, sh, compare_planes(packed, exp, ALL))
        assert not compare_planes(stack, exp, ALL), ("stack of hits vs oracle", bounces, ao, sh)
        # the product launch (no count planes: open cells, threshold runs, tags)
        p2 = _render(vrt, engine, sc, st, push, GB + ["color_f", "hit_id"], packed_bounces=1)
        assert not compare_planes(p2, exp, GB + ["color_f", "hit_id"]), ("packed chain, product launch", bounces, ao, sh)
        if bounces >= 2:
            assert int(exp["rays_total"].max()) > 2 + ao + int(sh), "the case should hold chains"
    sc.destroy()


def test_packed_chain_on_bricks(vrt, oracle, engine):
    n = 64
    vol = np.zeros((n, n, n), np.uint8)
    vol[:, 0:2, :] = 210; vol[:, :, 0:2] = 220; vol[0:2, :, :] = 240
    vol[20:40, 2:30, 30:34] = 205; vol[10:14, 2:8, 44:50] = 7; vol[45:50, 2:40, 10:14] = 230
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.blue_noise_standin(64)
    grid, pool = vrt.synthetic.bricks_from_dense(vol)
    sb = vrt.VoxelScene.from_bricks(engine, grid, pool, pal, sky=sky, noise=noise)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    res = (64, 48)
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.traceSettings.maxReflections = 4
    push = camera_push(vrt, (n, n, n), res, pos=(50.3, 40.2, 55.4), yaw=225.0, pitch=-25.0, frame=3)
    exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=ALL, nthreads=8)
    for pk in (2, 0):                                          # (brick scenes keep the stack of hits unless the option says 2)
        got = _render(vrt, engine, sb, st, push, ALL, packed_bounces=pk)
        assert not compare_planes(got, exp, ALL), ("brick scene, packed" if pk else "brick scene, stack", compare_planes(got, exp, ALL))
    sb.destroy()


def test_counting_twins(vrt, oracle, engine):
    """flag 16: the product march's own iteration counts -- through the threshold loop's counting twin (steps per axis) they must
    equal the merged loops' counts on a frame without ties, and stay an upper bound of them on lattice cameras; flag 16 | 32: the
    bytes the march asks for"""
    vol = vrt.synthetic.treehouse(96, seed=2)
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.blue_noise_standin(64)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    res = (160, 96)
    CNT = ["steps_primary", "steps_total", "rays_total"]
    for name, st in (("primary", vrt.VoxelRenderSettings.primary_only(res)), ("defaults", vrt.VoxelRenderSettings(targetResolution=res))):
        st.fsrSetttings.enable = False
        for pos, yaw, pitch, ties in (((48.37, 48.21, -70.0), 90.0, 0.0, False), ((20.3, 60.7, -30.2), 70.0, -20.0, False),
                                      ((0.0, 0.0, 0.0), 45.0, 0.0, True)):
            push = camera_push(vrt, (96, 96, 96), res, pos=pos, yaw=yaw, pitch=pitch, frame=2)
            exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=GB + ["hit_id"], nthreads=8)
            twin = _render(vrt, engine, sc, st, push, GB + ["hit_id"] + CNT, flags=16)
            merged = _render(vrt, engine, sc, st, push, GB + ["hit_id"] + CNT, flags=16, thresh_runs=0)
            looks = _render(vrt, engine, sc, st, push, GB + ["hit_id"] + CNT, flags=16 | 32)
            for got in (twin, merged, looks):
                assert not compare_planes(got, exp, GB + ["hit_id"]), (name, pos, "a counting launch renders the product's frame")
            assert (twin["rays_total"] == merged["rays_total"]).all()
            # a ray that HITS takes the same events in both loops: the threshold loop's per-axis steps are the iterations unless two
            # axes tie (then they count twice).  A ray that misses ends at the first OPEN cell it happens to look at, and the two
            # loops look at different cells (a lane's own clearance against the wave's smallest): either may stop first -- which
            # is why the bench counts the loop it times.  Both stay below the reference loop's count.
            a, b = twin["steps_total"].astype(np.int64), merged["steps_total"].astype(np.int64)
            hit = twin["hit_id"] != 0
            if name == "primary":
                assert (a[hit] >= b[hit]).all(), (name, pos, "per-axis steps are never fewer than iterations")
                if not ties:
                    assert (a[hit] == b[hit]).all(), (name, pos, int((a[hit] != b[hit]).sum()), "no ties: the threshold loop's steps ARE the iterations")
                    assert (twin["steps_primary"][hit] == merged["steps_primary"][hit]).all()
            ref = _render(vrt, engine, sc, st, push, CNT)
            assert (a <= ref["steps_total"].astype(np.int64) + 2 * twin["rays_total"]).all() or ties, (name, pos, "the march never takes more than the reference's loop")
            lk = looks["steps_total"].astype(np.int64)
            rays = twin["rays_total"].astype(np.int64)
            traced = b > 0
            # (an AO ray's look-ups are counted by the lane that makes them -- the rays come from the wave's pool -- so only sums are bounded)
            assert (lk[traced] >= 1).all() and lk.sum() <= 3 * (np.maximum(a, b).sum() + 2 * rays.sum()), (name, pos, "look-ups: at least one per traced pixel, at most three bytes per iteration + id")
            if not ties:
                assert lk.sum() < b.sum(), (name, pos, "the clearance runs are what keeps the march from asking per iteration")
    sc.destroy()


def test_counting_on_bricks(vrt, oracle, engine):
    """brick scenes: the march that is timed (brick_march_thresh) counts its own steps and bytes under the flags; on rays that hit,
    steps = the counting loop's iterations (no ties from this camera)"""
    vol = vrt.synthetic.treehouse(64, seed=4)
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.blue_noise_standin(64)
    grid, pool = vrt.synthetic.bricks_from_dense(vol)
    sb = vrt.VoxelScene.from_bricks(engine, grid, pool, pal, sky=sky, noise=noise)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    res = (96, 64)
    CNT = ["steps_primary", "steps_total", "rays_total"]
    st = vrt.VoxelRenderSettings.primary_only(res)
    push = camera_push(vrt, (64, 64, 64), res, pos=(32.37, 30.21, -50.0), frame=2)
    exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=GB + ["hit_id"] + CNT, nthreads=8)
    twin = _render(vrt, engine, sb, st, push, GB + ["hit_id"] + CNT, flags=16)
    merged = _render(vrt, engine, sb, st, push, GB + ["hit_id"] + CNT, flags=16, thresh_runs=0)
    looks = _render(vrt, engine, sb, st, push, GB + ["hit_id"] + CNT, flags=16 | 32)
    for got in (twin, merged, looks):
        assert not compare_planes(got, exp, GB + ["hit_id"])
    hit = twin["hit_id"] != 0
    assert hit.any() and (twin["steps_primary"][hit] == merged["steps_primary"][hit]).all()
    assert (twin["steps_primary"][hit] == exp["steps_primary"][hit]).all(), "a ray that hits takes the reference's iterations"
    lk = looks["steps_primary"].astype(np.int64)
    assert (lk[hit] >= 10).all() and (lk[hit] % 1 == 0).all()         # at least one brick word + fine byte + id
    sb.destroy()
