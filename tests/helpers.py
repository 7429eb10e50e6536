"""Shared helpers for the parity tests: build the same scene for the HIP path and the oracle."""
import numpy as np


def camera_push(vrt, dims, resolution, pos=None, yaw=90.0, pitch=0.0, frame=0, jitter=(0.0, 0.0)):
    W, H, D = dims
    if pos is None:
        pos = (W / 2.0 + 0.37, H / 2.0 + 0.21, -0.8 * D)
    cam = vrt.CameraController(position=pos, yaw=yaw, pitch=pitch)
    return vrt.make_push(cam, dims, resolution, frame, jitter)


def metallic_palette(vrt, ids=range(200, 256), value=0.8):
    return vrt.synthetic.default_palette(metallic_ids=ids, metallic_value=value)


def compare_planes(got: dict, exp: dict, names, exact=True):
    bad = []
    for n in names:
        g, e = np.asarray(got[n]), np.asarray(exp[n])
        if g.dtype != e.dtype:
            g = g.view(e.dtype) if g.dtype.itemsize == e.dtype.itemsize else g.astype(e.dtype)
        if g.dtype.kind == "f":
            same = (g.view(np.uint32) == e.view(np.uint32)) | (np.isnan(g) & np.isnan(e))
        else:
            same = g == e
        if not same.all():
            idx = np.argwhere(~same)
            bad.append((n, int((~same).sum()), idx[0].tolist(), g[tuple(idx[0])].tolist() if g.ndim else g, e[tuple(idx[0])].tolist()))
    return bad
