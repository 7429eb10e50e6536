"""CPU checks (oracle only, no GPU) of the two claims the march-skipping structures of the HIP path rest on:
  * open cells (vrt_device.hip launch_open_cells): a ray standing on an empty voxel with no solid voxel in the box between the
    voxel and the volume's corner in the octant of its direction hits nothing, whatever its direction within the octant;
  * tile tags (k_tile_tags): the pixel of a primary ray that hits a voxel lies inside the screen rectangle of that voxel's
    4^3 cell grown by one voxel (+ 2 pixels) under the projection [U V C] (a, b, lambda)^T = p - cam."""
import numpy as np
import pytest

from helpers import camera_push


def open_cells(vol, sx, sy, sz):
    """open[z, y, x]: no solid voxel at (x + a sx, y + b sy, z + c sz), a, b, c >= 0."""
    e = vol == 0
    for axis, s in ((2, sx), (1, sy), (0, sz)):
        e = np.flip(np.logical_and.accumulate(np.flip(e, axis), axis), axis) if s > 0 else np.logical_and.accumulate(e, axis)
    return e


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_a_ray_on_an_open_cell_hits_nothing(vrt, oracle, seed):
    rng = np.random.default_rng(seed)
    vol = vrt.synthetic.treehouse(32, seed=seed) if seed != 3 else (rng.random((20, 28, 36)) < 0.01).astype(np.uint8) * np.uint8(9)
    D, H, W = vol.shape
    osn = oracle.OracleScene(vol, vrt.synthetic.default_palette())
    checked = 0
    for o in range(8):
        sx, sy, sz = (1 if o & 1 else -1), (1 if o & 2 else -1), (1 if o & 4 else -1)
        op = open_cells(vol, sx, sy, sz)
        assert not (op & (vol != 0)).any()
        zs, ys, xs = np.nonzero(op)
        if xs.size == 0:
            continue
        for k in rng.choice(xs.size, size=min(400, xs.size), replace=False):
            start = np.array([xs[k], ys[k], zs[k]], np.float32) + rng.uniform(0.01, 0.99, 3).astype(np.float32)
            d = rng.uniform(0.0, 1.0, 3) * np.array([sx, sy, sz])
            if rng.integers(0, 4) == 0:
                d[rng.integers(0, 3)] = 0.0                          # a ray that does not move along one axis
            if not np.any(d):
                continue
            d = (d / np.linalg.norm(d)).astype(np.float32)
            h = oracle.trace_ray(osn, start, d, 4096)
            assert h.material == 0, (o, start, d, list(h.voxel))
            checked += 1
    assert checked > 500
    # ... and the flag is not vacuous: some rays from cells that are NOT open do hit
    zs, ys, xs = np.nonzero((vol == 0) & ~open_cells(vol, 1, 1, 1))
    hits = 0
    for k in rng.choice(xs.size, size=min(300, xs.size), replace=False):
        d = rng.uniform(0.05, 1.0, 3); d = (d / np.linalg.norm(d)).astype(np.float32)
        hits += oracle.trace_ray(osn, np.array([xs[k], ys[k], zs[k]], np.float32) + 0.5, d, 4096).material != 0
    assert hits > 0


def test_a_hit_pixel_lies_in_the_rectangle_of_its_cell(vrt, oracle):
    vol = vrt.synthetic.treehouse(48, seed=5)
    osn = oracle.OracleScene(vol, vrt.synthetic.default_palette(), sky=vrt.synthetic.sky_gradient(16, 8))
    res = (96, 72)
    W, H = res
    for pos, yaw, pitch, jitter in (((24.3, 24.2, -40.0), 90.0, 0.0, (0.0, 0.0)), ((70.0, 60.0, -20.0), 135.0, -30.0, (0.4, -0.3)),
                                    ((-30.0, 30.0, 24.0), 0.0, -5.0, (0.0, 0.0)), ((24.0, 120.0, 24.0), 90.0, -89.0, (-0.5, 0.5))):
        push = camera_push(vrt, (48, 48, 48), res, pos=pos, yaw=yaw, pitch=pitch, jitter=jitter)
        st = vrt.VoxelRenderSettings.primary_only(res)
        out = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=["hit_id", "hit_voxel"], nthreads=8)
        cd = np.array(list(push.cam_dir)[:3], np.float64); cd /= np.linalg.norm(cd)
        U = np.array(list(push.cam_right)[:3], np.float64)
        V = np.array(list(push.cam_up)[:3], np.float64) * H / W
        Cv = cd + np.array([push.camera_jitter[0] / W * -2.0, push.camera_jitter[1] / H * 2.0, 0.0])
        Minv = np.linalg.inv(np.stack([U, V, Cv], axis=1))
        cam = np.array(list(push.cam_pos)[:3], np.float64)
        ys, xs = np.nonzero(out["hit_id"])
        assert xs.size > 200
        for y, x in zip(ys[::7], xs[::7]):
            cell = (out["hit_voxel"][y, x].astype(np.int64) // 4) * 4
            corners = np.array([[cell[0] - 1 + 6 * (k & 1), cell[1] - 1 + 6 * ((k >> 1) & 1), cell[2] - 1 + 6 * (k >> 2)] for k in range(8)], np.float64)
            abl = (Minv @ (corners - cam).T).T
            assert (abl[:, 2] > 0).all()
            fx, fy = (abl[:, 0] / abl[:, 2] + 1) * 0.5 * W, (abl[:, 1] / abl[:, 2] + 1) * 0.5 * H
            assert fx.min() - 2 <= x + 0.5 <= fx.max() + 2 and fy.min() - 2 <= y + 0.5 <= fy.max() + 2, (pos, x, y, fx.min(), fx.max(), fy.min(), fy.max())
