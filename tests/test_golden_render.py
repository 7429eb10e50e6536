"""Committed golden vectors of the render path (tests/golden/render_*.npz, made by tests/golden/make_render_fixtures.py
from the oracle): the oracle must still reproduce them (CPU), and so must the HIP path through the C-ABI (GPU)."""
import os
import zlib

import numpy as np
import pytest

from helpers import camera_push, compare_planes, metallic_palette

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PLANES = ["color8", "depth", "mask8", "position", "normal8", "hit_id", "hit_voxel", "hit_mask", "steps_primary",
          "steps_total", "rays_total", "color_f"]


def _cubes64_inputs(vrt):
    vol = vrt.synthetic.floating_cubes(64, seed=1, count=120)
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.blue_noise_standin(64)
    st = vrt.VoxelRenderSettings(targetResolution=(64, 64))
    st.fsrSetttings.enable = False
    push = camera_push(vrt, (64, 64, 64), (64, 64), frame=3)
    return vol, pal, sky, noise, st, push


def test_oracle_reproduces_golden_frame(vrt, oracle):
    g = np.load(os.path.join(GOLD, "render_cubes64.npz"))
    vol, pal, sky, noise, st, push = _cubes64_inputs(vrt)
    out = oracle.render(oracle.OracleScene(vol, pal, sky=sky, noise=noise), push, oracle.params_from(st.to_c()), nthreads=4)
    assert not compare_planes(out, g, PLANES)
    assert zlib.crc32(out["hit_id"].tobytes()) == int(g["crc_hit_id"][0])
    den = oracle.denoise(out["color8"], out["normal8"], out["position"])
    assert (den == g["denoised8"]).all()
    assert (g["hit_id"] != 0).mean() > 0.1 and int(g["rays_total"].max()) > 6       # the frame has hits and bounces


def test_oracle_reproduces_golden_denoise_and_numeric(oracle):
    g = np.load(os.path.join(GOLD, "render_denoise16.npz"))
    for mode in (0, 1):
        for it in (1, 2, 3):
            out = oracle.denoise(g["color"], g["normal"], g["position"], iterations=it, mode=mode)
            assert (out == g[f"out_mode{mode}_iter{it}"]).all(), (mode, it)
    # the edge is preserved by the weighted passes: columns 7 and 8 stay further apart than a plain blur leaves them
    n = np.load(os.path.join(GOLD, "render_numeric.npz"))
    L = oracle.lib()
    assert (np.array([L.vo_unorm8(float(v)) for v in n["x"]], np.uint8) == n["unorm8"]).all()
    assert (np.array([L.vo_snorm8(float(v)) for v in n["x"]], np.int8) == n["snorm8"]).all()
    ex = np.array([L.vo_expf(float(v)) for v in np.linspace(-100.0, 5.0, 2101, dtype=np.float32)], np.float32)
    assert (ex.view(np.uint32) == n["exp"].view(np.uint32)).all()
    asn = np.array([L.vo_asinf(float(v)) for v in np.linspace(-1.2, 1.2, 481, dtype=np.float32)], np.float32)
    assert (asn.view(np.uint32) == n["asin"].view(np.uint32)).all()
    gq = np.linspace(-4.0, 4.0, 801, dtype=np.float32)[::40]
    at = np.array([[L.vo_atan2f(float(a), float(b)) for b in gq] for a in gq], np.float32)
    assert (at.view(np.uint32) == n["atan2"].view(np.uint32)).all()
    # spot values of the table: UNORM8 rounds half up, SNORM8 clamps at -1
    x = n["x"]
    assert n["unorm8"][np.nanargmin(np.abs(x - 1.0))] == 255 and n["unorm8"][np.nanargmin(np.abs(x + 0.3))] == 0
    assert n["snorm8"][np.nanargmin(np.abs(x + 1.5))] == -127 and n["unorm8"][np.isnan(x)][0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("trav", ["DF", "DENSE", "BITMASK", "JUMP", "DFJ"])
def test_hip_reproduces_golden_frame(vrt, engine, trav):
    g = np.load(os.path.join(GOLD, "render_cubes64.npz"))
    vol, pal, sky, noise, st, push = _cubes64_inputs(vrt)
    st.traceSettings.traversal = getattr(vrt, "TRAVERSAL_" + trav)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    gb = vrt.GeometryStage(engine, st, sc, debug_planes=True).record(push)
    engine.synchronize()
    names = PLANES if trav not in ("JUMP", "DFJ") else [p for p in PLANES if not p.startswith("steps_")]
    assert not compare_planes(gb.numpy(), g, names)
    den = vrt.DenoiserStage(engine, st).record(gb.color, gb.normal, gb.position)
    engine.synchronize()
    assert (den.cpu().numpy() == g["denoised8"]).all()


@pytest.mark.gpu
def test_hip_reproduces_golden_denoise(vrt, engine):
    import torch
    g = np.load(os.path.join(GOLD, "render_denoise16.npz"))
    dev = engine.torch_device
    c, n, p = (torch.from_numpy(g[k]).to(dev) for k in ("color", "normal", "position"))
    for mode in (0, 1):
        for it in (1, 2, 3):
            st = vrt.VoxelRenderSettings(targetResolution=(16, 16))
            st.denoiserSettings.iterations = it
            st.denoiserSettings.mode = mode
            out = vrt.DenoiserStage(engine, st).record(c, n, p)
            engine.synchronize()
            assert (out.cpu().numpy() == g[f"out_mode{mode}_iter{it}"]).all(), (mode, it)


# ---- BASELINE configs[0] at its stated workload: floating_cubes 128^3 @ 256x256, primary rays only ----------------------

C1_PLANES = ["hit_id", "hit_mask", "hit_voxel", "steps_primary", "color8", "depth", "normal8"]


def _config1_inputs():
    import sys
    sys.path.insert(0, GOLD)
    from make_render_fixtures import config1_inputs
    return config1_inputs()


def test_oracle_reproduces_config1(vrt, oracle):
    g = np.load(os.path.join(GOLD, "render_config1.npz"))
    vol, pal, st, push = _config1_inputs()
    out = oracle.render(oracle.OracleScene(vol, pal), push, oracle.params_from(st.to_c()), planes=C1_PLANES, nthreads=8)
    assert not compare_planes(out, g, C1_PLANES)
    assert zlib.crc32(out["hit_id"].tobytes()) == int(g["crc_hit_id"][0])
    assert int(out["steps_primary"].astype(np.int64).sum()) == int(g["step_sum"][0])
    assert 0.1 < (g["hit_id"] != 0).mean() < 0.9


@pytest.mark.gpu
@pytest.mark.parametrize("trav", ["DF", "DENSE", "BITMASK", "JUMP", "DFJ"])
def test_hip_reproduces_config1(vrt, engine, trav):
    g = np.load(os.path.join(GOLD, "render_config1.npz"))
    vol, pal, st, push = _config1_inputs()
    st.traceSettings.traversal = getattr(vrt, "TRAVERSAL_" + trav)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal)
    gb = vrt.GeometryStage(engine, st, sc, debug_planes=True).record(push)
    engine.synchronize()
    out = gb.numpy()
    names = C1_PLANES if trav not in ("JUMP", "DFJ") else [p for p in C1_PLANES if not p.startswith("steps_")]
    assert not compare_planes(out, g, names)
    assert zlib.crc32(out["hit_id"].tobytes()) == int(g["crc_hit_id"][0])
    if trav not in ("JUMP", "DFJ"):
        assert int(out["steps_primary"].astype(np.int64).sum()) == int(g["step_sum"][0])
