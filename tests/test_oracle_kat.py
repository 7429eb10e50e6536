"""Known-answer tests that pin the CPU oracle (hand-computed; the reference ships no vectors for this path:
SURVEY.md section 4 / 8(c)).  Each case cites the shader lines whose arithmetic was worked out by hand."""
import math

import numpy as np
import pytest


def _scene(oracle, vol, pal=None):
    if pal is None:
        pal = np.zeros((256, 5), np.float32); pal[:, :3] = 0.5
    return oracle.OracleScene(vol, pal)


def test_axis_ray_single_voxel(oracle):
    # 8^3 volume, voxel id 7 at (4,4,4); ray from (4.5,4.5,-10) along +z.
    # boxIntersection (frag:109-125): tz = 10 -> p0 = start + 10.1*dir = (4.5,4.5,0.1); mapPos = (4,4,0)
    # sideDist.z = (1*(0-0.1)+0.5+0.5)*1 = 0.9 (frag:144); 4 z-steps -> 5 fetches; d = 4.9-1 = 3.9
    vol = np.zeros((8, 8, 8), np.uint8); vol[4, 4, 4] = 7
    h = oracle.trace_ray(_scene(oracle, vol), (4.5, 4.5, -10.0), (0.0, 0.0, 1.0))
    assert h.material == 7 and list(h.voxel) == [4, 4, 4] and h.mask == 4 and h.steps == 5
    assert list(h.normal) == [0.0, 0.0, -1.0]
    assert h.p0[2] == pytest.approx(0.1, abs=1e-6)
    assert h.pos[2] == pytest.approx(4.0, abs=1e-5) and h.pos[0] == 4.5 and h.pos[1] == 4.5
    assert math.isinf(h.delta[0]) and math.isinf(h.side[0])          # axis-parallel: 1/0 = inf, never stepped


def test_miss_away_from_box(oracle):
    vol = np.zeros((8, 8, 8), np.uint8); vol[4, 4, 4] = 7
    h = oracle.trace_ray(_scene(oracle, vol), (4.5, 4.5, -10.0), (0.0, 0.0, -1.0))
    assert h.material == 0 and h.steps == 0 and h.mask == 0
    assert list(h.pos) == [0.0, 0.0, 0.0] and list(h.normal) == [0.0, 0.0, 0.0]     # canonical rule B


def test_exact_diagonal_tie(oracle):
    # dir = (1,1,0)/sqrt2 from (-2,-2,4.5): sideDist.x == sideDist.y at every step, so both axes step
    # together (frag:164 lessThanEqual ties) and the hit has a two-axis mask and a diagonal normal.
    vol = np.zeros((8, 8, 8), np.uint8); vol[4, 4, 4] = 3
    s = np.float32(1.0) / np.sqrt(np.float32(2.0))
    h = oracle.trace_ray(_scene(oracle, vol), (-2.0, -2.0, 4.5), (float(s), float(s), 0.0))
    assert h.material == 3 and list(h.voxel) == [4, 4, 4] and h.mask == 3 and h.steps == 5
    assert h.normal[0] == h.normal[1] and h.normal[2] == 0.0
    assert h.normal[0] == pytest.approx(-1 / math.sqrt(2), abs=1e-7)


def test_first_voxel_solid_entry_mask(oracle):
    # the very first sampled voxel (box-entry voxel) is solid: mask = entry axis (canonical rule A)
    vol = np.zeros((8, 8, 8), np.uint8); vol[0, 4, 4] = 9
    h = oracle.trace_ray(_scene(oracle, vol), (4.5, 4.5, -3.0), (0.0, 0.0, 1.0))
    assert h.material == 9 and h.steps == 1 and h.mask == 4 and list(h.voxel) == [4, 4, 0]
    assert list(h.normal) == [0.0, 0.0, -1.0]


def test_camera_inside_solid(oracle):
    vol = np.zeros((8, 8, 8), np.uint8); vol[4, 4, 4] = 5
    h = oracle.trace_ray(_scene(oracle, vol), (4.5, 4.5, 4.5), (0.0, 0.6, 0.8))
    assert h.material == 5 and h.steps == 1 and h.mask == 0
    assert list(h.normal) == [0.0, 0.0, 0.0] and list(h.pos) == [4.5, 4.5, 4.5]     # d = 0


def test_step_exhaustion(oracle):
    vol = np.zeros((64, 8, 8), np.uint8); vol[60, 4, 4] = 1
    sc = _scene(oracle, vol)
    assert oracle.trace_ray(sc, (4.5, 4.5, -1.0), (0.0, 0.0, 1.0), max_steps=61).material == 1
    h = oracle.trace_ray(sc, (4.5, 4.5, -1.0), (0.0, 0.0, 1.0), max_steps=60)       # needs 61 fetches
    assert h.material == 0 and h.steps == 60


def test_generic_ray_matches_float64_dda(oracle):
    # generic direction: compare the visited hit voxel with an independent float64 DDA
    rng = np.random.default_rng(5)
    vol = (rng.random((16, 16, 16)) < 0.03).astype(np.uint8) * 4
    sc = _scene(oracle, vol)
    for _ in range(200):
        o = np.array([rng.uniform(2, 14), rng.uniform(2, 14), -5.0])
        d = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), 1.0]); d /= np.linalg.norm(d)
        h = oracle.trace_ray(sc, o, d)
        # float64 reference march with tiny steps
        t0 = (0 - o[2]) / d[2] + 0.1
        hit = None
        for t in np.arange(t0, t0 + 40, 0.002):
            p = o + t * d
            ip = np.floor(p).astype(int)
            if (ip < 0).any() or (ip >= 16).any():
                break
            if vol[ip[2], ip[1], ip[0]]:
                hit = ip; break
        if hit is None:
            continue                      # fine march may clip a corner the DDA legitimately passes through
        assert h.material == 4
        assert np.abs(np.array(list(h.voxel)) - hit).max() <= 1      # corner clipping tolerance of the marcher


def test_primary_ray_center_and_corner(oracle, vrt):
    from helpers import camera_push
    push = camera_push(vrt, (8, 8, 8), (4, 2), pos=(4.0, 4.0, -10.0))
    # yaw 90, pitch 0: dir ~ +z, right = +x, up = -y (camera_controller.cpp:15-28); H/W = 0.5
    s, d = oracle.primary_ray(push, 0, 0)
    assert list(s) == [4.0, 4.0, -10.0]
    v = np.array([-0.75, 0.5 * 0.5, 1.0]); v /= np.linalg.norm(v)      # sx = -0.75, sy = -0.5 -> +y (up on screen)
    assert np.allclose(d, v, atol=1e-6)


def test_math_accuracy(oracle):
    l = oracle.lib()
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = float(np.float32(rng.normal())), float(np.float32(rng.normal()))
        assert abs(l.vo_atan2f(y, x) - math.atan2(y, x)) < 2e-6
        u = float(np.float32(rng.uniform(-1, 1)))
        assert abs(l.vo_asinf(u) - math.asin(u)) < 2e-6
        e = float(np.float32(-rng.uniform(0, 80)))
        assert abs(l.vo_expf(e) - math.exp(e)) <= 3e-7 * math.exp(e) + 1e-38
    assert l.vo_expf(0.0) == 1.0 and l.vo_expf(-0.0) == 1.0 and l.vo_expf(-1000.0) == 0.0
    assert l.vo_expf(float("-inf")) == 0.0
    assert l.vo_atan2f(0.0, 0.0) == 0.0


def test_quantisation_table(oracle):
    l = oracle.lib()
    assert [l.vo_unorm8(x) for x in (-1.0, 0.0, 0.5, 0.9, 1.0, 7.0)] == [0, 0, 128, 230, 255, 255]
    assert l.vo_unorm8(float("nan")) == 0
    assert [l.vo_snorm8(x) for x in (-2.0, -1.0, -0.70710677, 0.0, 0.70710677, 1.0)] == [-127, -127, -90, 0, 90, 127]


def test_denoise_pass_params(oracle):
    p = oracle.DenoiseParams()
    oracle.lib().vo_denoise_pass_params(0, 20.4, 0.01, 0.1, 2.0, p)
    assert math.isinf(p.phi_color) and math.isinf(p.phi_normal) and math.isinf(p.phi_pos) and p.step_width == 1.0
    oracle.lib().vo_denoise_pass_params(2, 20.4, 0.01, 0.1, 2.0, p)
    assert p.phi_color == pytest.approx(10.2) and p.step_width == 5.0


def test_denoise_pass0_is_gaussian_blur(oracle):
    # pass 0: phi = +inf -> all edge weights exactly 1 -> plain 3x3 blur with exp(-(x^2+y^2)/8) weights
    rng = np.random.default_rng(3)
    H, W = 6, 7
    color = rng.integers(0, 256, (H, W, 4), dtype=np.uint8); color[..., 3] = 0
    normal = rng.integers(-127, 128, (H, W, 4)).astype(np.int8)
    pos = rng.normal(size=(H, W, 4)).astype(np.float32)
    out = oracle.denoise(color, normal, pos, iterations=1)
    k = np.array([[math.exp(-(x * x + y * y) / 8) for x in (-1, 0, 1)] for y in (-1, 0, 1)])
    c = color.astype(np.float64) / 255
    pad = np.pad(c, ((1, 1), (1, 1), (0, 0)), mode="edge")
    ref = np.zeros_like(c)
    for dy in range(3):
        for dx in range(3):
            ref += k[dy, dx] * pad[dy:dy + H, dx:dx + W]
    ref = np.floor(np.clip(ref / k.sum(), 0, 1) * 255 + 0.5)
    assert np.abs(out.astype(np.int64) - ref.astype(np.int64)).max() <= 1       # fp32 vs fp64 rounding at .5 ties
    # as-shipped std140 aliasing: 3 taps (-1,-1)*0.7788, (1,-1)*1, (0,0)*0.7788
    out2 = oracle.denoise(color, normal, pos, iterations=1, mode=1)
    k2 = [(-1, -1, math.exp(-0.25)), (1, -1, 1.0), (0, 0, math.exp(-0.25))]
    ref2 = sum(w * pad[1 + dy:1 + dy + H, 1 + dx:1 + dx + W] for dx, dy, w in k2) / sum(w for _, _, w in k2)
    ref2 = np.floor(np.clip(ref2, 0, 1) * 255 + 0.5)
    assert np.abs(out2.astype(np.int64) - ref2.astype(np.int64)).max() <= 1


def test_shading_known_answer(oracle, vrt):
    # one lit voxel face, AO off, no shadow: color = (N.L*lightColor*I + 1*ambientIntensity*sky(N)) * albedo (frag:236-258)
    from helpers import camera_push
    vol = np.zeros((8, 8, 8), np.uint8); vol[4, 0:8, 0:8] = 2        # slab facing -z
    pal = np.zeros((256, 5), np.float32); pal[2, :3] = (0.2, 0.4, 0.8)
    sky = np.zeros((1, 1, 4), np.float32); sky[0, 0] = (0.5, 0.25, 0.125, 1.0)
    sc = oracle.OracleScene(vol, pal, sky=sky)
    st = vrt.VoxelRenderSettings.primary_only((2, 2))
    st.lightSettings.direction = (0.0, 0.0, -1.0)
    push = camera_push(vrt, (8, 8, 8), (2, 2), pos=(4.3, 4.2, -2.0))
    out = oracle.render(sc, push, oracle.params_from(st.to_c()))
    assert (out["hit_id"] == 2).all() and (out["hit_mask"] == 4).all()
    exp = (1.0 + np.array([0.5, 0.25, 0.125])) * np.array([0.2, 0.4, 0.8])
    assert np.allclose(out["color_f"][0, 0], exp, rtol=1e-6)
    assert (out["normal8"][..., 2] == -127).all() and (out["mask8"] == 230).all()
    assert np.allclose(out["depth"], np.linalg.norm(out["position"][..., :3] - np.array([4.3, 4.2, -2.0], np.float32), axis=-1), rtol=1e-6)


def test_oracle_brick_storage_equals_dense(vrt, oracle):
    """vo_scene's brick storage (8^3 brick pool + pointer grid; getVoxel reads through it) is the same texture: a frame of the
    same content rendered from both storages is identical, plane by plane."""
    from helpers import camera_push, compare_planes, metallic_palette
    grid, pool = vrt.synthetic.sparse_brick_scene(64, 0.08, seed=3)
    vol = vrt.synthetic.dense_from_bricks(grid, pool)
    g2, p2 = vrt.synthetic.bricks_from_dense(vol)
    assert (g2 == grid).all() and (p2 == pool).all() and 0 < pool.shape[0] < grid.size
    pal = metallic_palette(vrt)
    st = vrt.VoxelRenderSettings(targetResolution=(48, 40))
    st.fsrSetttings.enable = False
    push = camera_push(vrt, (64, 64, 64), (48, 40), frame=2)
    a = oracle.render(oracle.OracleScene(vol, pal), push, oracle.params_from(st.to_c()), nthreads=4)
    b = oracle.render(oracle.OracleScene(None, pal, bricks=(grid, pool)), push, oracle.params_from(st.to_c()), nthreads=4)
    assert not compare_planes(a, b, list(a.keys()))
    assert (a["hit_id"] != 0).mean() > 0.05
