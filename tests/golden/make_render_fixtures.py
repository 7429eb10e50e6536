"""Writes tests/golden/render_*.npz: golden vectors of the render path, produced by the CPU oracle (oracle/) on seeded
synthetic inputs.  The reference itself ships no vectors for this path and cannot run here, so these pin
the ORACLE against drift between rounds -- tests/test_golden_render.py checks that today's oracle and the HIP path
both reproduce them bit for bit.  Regenerate only on purpose:  python tests/golden/make_render_fixtures.py"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxel_raytracing_amd as vrt                     # host-side helpers only (synthetic scenes, camera); no GPU needed
from oracle import oracle
from helpers import camera_push, metallic_palette

HERE = os.path.dirname(os.path.abspath(__file__))


def cubes64():
    """64^3 procedural scene at 64x64 px, reference defaults (AO 4, shadow, 5 bounces)."""
    vol = vrt.synthetic.floating_cubes(64, seed=1, count=120)
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.blue_noise_standin(64)
    st = vrt.VoxelRenderSettings(targetResolution=(64, 64))
    st.fsrSetttings.enable = False
    push = camera_push(vrt, (64, 64, 64), (64, 64), frame=3)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    out = oracle.render(osn, push, oracle.params_from(st.to_c()), nthreads=4)
    den = oracle.denoise(out["color8"], out["normal8"], out["position"])
    keep = {k: out[k] for k in ("color8", "depth", "mask8", "position", "normal8", "hit_id", "hit_voxel", "hit_mask",
                                "steps_primary", "steps_total", "rays_total", "color_f")}
    keep["denoised8"] = den
    keep["crc_hit_id"] = np.array([zlib.crc32(out["hit_id"].tobytes())], np.uint32)
    np.savez_compressed(os.path.join(HERE, "render_cubes64.npz"), **keep)
    return keep


def denoise16():
    """16x16 synthetic G-buffer with a normal / position edge; passes 0 and 0+1, both UBO modes."""
    rng = np.random.default_rng(12345)
    color = rng.integers(0, 256, size=(16, 16, 4), dtype=np.uint8)
    normal = np.zeros((16, 16, 4), np.int8); normal[:, :8, 0] = 127; normal[:, 8:, 1] = 127      # edge between x = 7 and 8
    position = np.zeros((16, 16, 4), np.float32)
    yy, xx = np.meshgrid(np.arange(16, dtype=np.float32), np.arange(16, dtype=np.float32), indexing="ij")
    position[..., 0] = xx * 0.25; position[..., 1] = yy * 0.25; position[:, 8:, 2] = 3.0          # depth step at the edge
    out = {"color": color, "normal": normal, "position": position}
    for mode in (0, 1):
        for it in (1, 2, 3):
            out[f"out_mode{mode}_iter{it}"] = oracle.denoise(color, normal, position, iterations=it, mode=mode)
    np.savez_compressed(os.path.join(HERE, "render_denoise16.npz"), **out)
    return out


def quantisation():
    """UNORM8 / SNORM8 conversion table + the pinned transcendental functions on a grid."""
    L = oracle.lib()
    x = np.concatenate([np.linspace(-1.5, 1.5, 3001, dtype=np.float32), np.array([np.nan, np.inf, -np.inf, 0.5 / 255, 1.5 / 255], np.float32)])
    un = np.array([L.vo_unorm8(float(v)) for v in x], np.uint8)
    sn = np.array([L.vo_snorm8(float(v)) for v in x], np.int8)
    g = np.linspace(-4.0, 4.0, 801, dtype=np.float32)
    at = np.array([[L.vo_atan2f(float(a), float(b)) for b in g[::40]] for a in g[::40]], np.float32)
    asn = np.array([L.vo_asinf(float(v)) for v in np.linspace(-1.2, 1.2, 481, dtype=np.float32)], np.float32)
    ex = np.array([L.vo_expf(float(v)) for v in np.linspace(-100.0, 5.0, 2101, dtype=np.float32)], np.float32)
    np.savez_compressed(os.path.join(HERE, "render_numeric.npz"), x=x, unorm8=un, snorm8=sn, atan2=at, asin=asn, exp=ex)


def config1_inputs():
    """BASELINE configs[0] at its stated size (SURVEY 8(d) "Config 1"): floating_cubes(N=128, seed=1), 256x256, primary rays
    only, the reference default camera scaled to the volume: (N/2, N/2, -0.8 N), yaw 90, pitch 0."""
    N = 128
    vol = vrt.synthetic.floating_cubes(N, seed=1)
    pal = vrt.synthetic.default_palette()
    st = vrt.VoxelRenderSettings.primary_only((256, 256))
    pos, yaw, pitch = vrt.synthetic.default_camera_for(N, N, N)
    push = vrt.make_push(vrt.CameraController(position=pos, yaw=yaw, pitch=pitch), (N, N, N), (256, 256))
    return vol, pal, st, push


def config1():
    vol, pal, st, push = config1_inputs()
    out = oracle.render(oracle.OracleScene(vol, pal), push, oracle.params_from(st.to_c()), nthreads=8)
    keep = {k: out[k] for k in ("hit_id", "hit_mask", "hit_voxel", "steps_primary", "color8", "depth", "normal8")}
    keep["crc_hit_id"] = np.array([zlib.crc32(out["hit_id"].tobytes())], np.uint32)
    keep["step_sum"] = np.array([int(out["steps_primary"].astype(np.int64).sum())], np.int64)
    np.savez_compressed(os.path.join(HERE, "render_config1.npz"), **keep)
    return keep


if __name__ == "__main__":
    oracle.build()
    cubes64(); denoise16(); quantisation(); config1()
    for f in sorted(os.listdir(HERE)):
        if f.startswith("render_"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
