"""K3 (a-trous denoiser) through the C-ABI vs the oracle: bit-exact RGBA8 output."""
import numpy as np
import pytest

from helpers import camera_push, metallic_palette

pytestmark = pytest.mark.gpu


def _gbuffer(vrt, engine, res=(112, 72)):
    vol = vrt.synthetic.floating_cubes(48, seed=4, count=60)
    sc = vrt.VoxelScene.from_dense(engine, vol, metallic_palette(vrt), sky=vrt.synthetic.sky_gradient(64, 32),
                                   noise=vrt.synthetic.blue_noise_standin(64))
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = 2
    stage = vrt.GeometryStage(engine, st, sc)
    gb = stage.record(camera_push(vrt, (48, 48, 48), res))
    engine.synchronize()
    return st, gb


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("iterations,step", [(0, 2.0), (1, 2.0), (2, 2.0), (3, 2.0), (5, 1.0), (2, 1.5), (3, 0.0), (2, 0.75)])
def test_denoise_bit_exact(vrt, oracle, engine, mode, iterations, step):
    st, gb = _gbuffer(vrt, engine)
    st.denoiserSettings.iterations = iterations
    st.denoiserSettings.stepWidth = step
    st.denoiserSettings.mode = mode
    den = vrt.DenoiserStage(engine, st)
    out = den.record(gb.color, gb.normal, gb.position)
    engine.synchronize()
    g = gb.numpy()
    exp = oracle.denoise(g["color8"], g["normal8"], g["position"], iterations=iterations, step_width0=step, mode=mode)
    got = out.cpu().numpy()
    assert (got == exp).all(), int((got != exp).sum())
    if iterations:
        assert (got != g["color8"]).any()


def test_denoise_rejects_bad_parameters(vrt, engine):
    st, gb = _gbuffer(vrt, engine, (32, 32))
    den = vrt.DenoiserStage(engine, st)
    st.denoiserSettings.phiColor0 = 0.0
    with pytest.raises(vrt.VrtError, match="phi"):
        den.record(gb.color, gb.normal, gb.position)
    st.denoiserSettings.phiColor0 = 20.4
    st.denoiserSettings.iterations = 11
    with pytest.raises(vrt.VrtError, match="iterations"):
        den.record(gb.color, gb.normal, gb.position)


def test_full_renderer_default_settings(vrt, oracle, engine):
    """VoxelRenderer with the reference's default settings (FSR 'Balanced' render scale, AO 4, 2 denoiser passes)."""
    vol = vrt.synthetic.treehouse(64, seed=2)
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.blue_noise_standin(128)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    st = vrt.VoxelRenderSettings(targetResolution=(320, 180))
    assert st.renderResolution() == (188, 105)                          # 10/17 scale, truncated (voxel_render_settings.cpp:3-6)
    r = vrt.VoxelRenderer(engine, st, sc)
    r.camera.position = np.array([32.3, 32.2, -50.0], np.float32)
    img = r.render(); engine.synchronize()
    push = r.push_constants()
    exp = oracle.render(oracle.OracleScene(vol, pal, sky=sky, noise=noise), push, oracle.params_from(st.to_c()), nthreads=8)
    eimg = oracle.denoise(exp["color8"], exp["normal8"], exp["position"])
    assert (img.cpu().numpy() == eimg).all()


# ---- VRT_DENOISE_FAST: a stated tolerance instead of bit-exactness ------------------------------------------------------
# The weighted passes with hardware exp2 / reciprocal: at most ONE RGBA8 code per channel and pass away from the exact
# filter of the same input (the bound in include/vrt.h).  Checked per pass -- every pass's input is the exact previous pass
# -- and end to end against the oracle, where the differences of earlier passes are filtered along (still <= passes codes).

@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("step", [2.0, 1.0])
def test_fast_denoise_within_one_code_per_pass(vrt, oracle, engine, mode, step):
    import torch
    st, gb = _gbuffer(vrt, engine, (200, 120))
    g = gb.numpy()
    worst = 0
    for iterations in (2, 3, 4):
        # exact passes 0 .. iterations-2 by the oracle, then the last pass alone: exact (oracle) vs fast (GPU)
        exp = oracle.denoise(g["color8"], g["normal8"], g["position"], iterations=iterations, step_width0=step, mode=mode)
        st.denoiserSettings.iterations = iterations
        st.denoiserSettings.stepWidth = step
        st.denoiserSettings.mode = mode | vrt.DENOISE_FAST
        out = vrt.DenoiserStage(engine, st).record(gb.color, gb.normal, gb.position).cpu().numpy()
        d = np.abs(out.astype(np.int32) - exp.astype(np.int32))
        worst = max(worst, int(d.max()))
        assert d.max() <= iterations - 1, (iterations, int(d.max()))          # pass 0 is exact in either mode; one code per weighted pass
        assert (d != 0).mean() < 0.02                                           # and nearly all pixels agree exactly
    print("fast denoise: worst difference", worst, "codes")


def test_fast_denoise_single_weighted_pass_bound(vrt, oracle, engine):
    """One weighted pass on an exact input (the golden 16x16 edge case and a random G-buffer): <= 1 code per channel."""
    import os
    import torch
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_denoise16.npz"))
    dev = engine.torch_device
    rng = np.random.default_rng(99)
    H, W = 96, 160
    cases = [(gold["color"], gold["normal"], gold["position"])]
    pos = np.zeros((H, W, 4), np.float32); pos[..., :3] = rng.uniform(0, 64, (H, W, 3)).astype(np.float32) * (rng.random((H, W, 1)) < 0.7)
    pos[..., :3] = np.round(pos[..., :3] * 4) / 4                                # clustered positions: non-trivial weights
    nrm = np.zeros((H, W, 4), np.int8); nrm[..., :3] = rng.choice(np.array([-127, 0, 90, 127], np.int8), (H, W, 3))
    cases.append((rng.integers(0, 256, (H, W, 4), dtype=np.uint8), nrm, pos))
    for color, normal, position in cases:
        c, n, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (color, normal, position))
        for mode in (0, 1):
            st = vrt.VoxelRenderSettings(targetResolution=(color.shape[1], color.shape[0]))
            st.denoiserSettings.iterations = 2
            st.denoiserSettings.mode = mode | vrt.DENOISE_FAST
            fast = vrt.DenoiserStage(engine, st).record(c, n, p).cpu().numpy()
            exact = oracle.denoise(color, normal, position, iterations=2, mode=mode)
            assert np.abs(fast.astype(np.int32) - exact.astype(np.int32)).max() <= 1, mode


# ---- the verified pass: exact output from a cheap evaluation + a literal one where the cheap one cannot vouch ----------------
# (csrc/vrt_denoise_bound.h).  The default for whole-frame weighted passes with an integral tap offset; "denoise_verified" = 0 has
# the exact kernels compute every pixel.

def _hostile_gbuffer(rng, W, H):
    """Random codes in every channel (alpha and w included, SNORM -128 too), positions clustered so that the weights are
    neither 0 nor 1, a sprinkling of equal neighbours, huge, infinite and NaN positions."""
    color = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    flat = rng.random((H, W, 1)) < 0.3
    color = np.where(flat, color // 64 * 64, color).astype(np.uint8)             # flat patches: sums whose mean is a code exactly
    nrm = rng.choice(np.array([-128, -127, -90, -73, -1, 0, 1, 73, 90, 127], np.int8), (H, W, 4))
    nrm = np.where(rng.random((H, W, 1)) < 0.5, np.array([0, 127, 0, 0], np.int8), nrm).astype(np.int8)
    pos = rng.uniform(0, 64, (H, W, 4)).astype(np.float32)
    pos = (np.round(pos * 4) / 4 + rng.normal(0, 0.05, (H, W, 4))).astype(np.float32)
    pos[..., 3] = np.where(rng.random((H, W)) < 0.9, 0.0, pos[..., 3])
    odd = rng.random((H, W))
    pos[odd < 0.002] = np.float32(np.nan)
    pos[(odd >= 0.002) & (odd < 0.004)] = np.float32(np.inf)
    pos[(odd >= 0.004) & (odd < 0.006)] = np.float32(-3e19)
    pos[(odd >= 0.006) & (odd < 0.3)] = 0.0                                       # sky
    return color, nrm, pos


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("size", [(200, 120), (333, 61), (64, 8), (70, 13), (1, 1), (5, 3), (130, 2)])
def test_verified_pass_is_the_literal_pass(vrt, oracle, engine, mode, size):
    import torch
    W, H = size
    rng = np.random.default_rng(W * 1000 + H + mode)
    color, nrm, pos = _hostile_gbuffer(rng, W, H)
    dev = engine.torch_device
    c, n, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (color, nrm, pos))
    for iterations, step, phis in ((2, 2.0, None), (3, 2.0, None), (3, 1.0, None), (2, 2.0, (0.5, 0.2, 30.0)), (2, 4.0, (3.0, 1.0, 1.0))):
        st = vrt.VoxelRenderSettings(targetResolution=(W, H))
        d = st.denoiserSettings
        d.iterations, d.stepWidth, d.mode = iterations, step, mode
        if phis: d.phiColor0, d.phiNormal0, d.phiPos0 = phis
        stage = vrt.DenoiserStage(engine, st)
        assert stage.guard(1) < 0.02, stage.guard(1)
        engine.set_option("denoise_count", 1)
        try:
            got = stage.record(c, n, p).cpu().numpy().copy()
            redone = [stage.redone(i) for i in range(iterations)]
            engine.set_option("denoise_verified", 0)
            lit = vrt.DenoiserStage(engine, st).record(c, n, p).cpu().numpy().copy()
            assert stage.redone(1) == 0
        finally:
            engine.set_option("denoise_verified", 1); engine.set_option("denoise_count", 0)
        kw = dict(zip(("phi_color0", "phi_normal0", "phi_pos0"), phis)) if phis else {}
        exp = oracle.denoise(color, nrm, pos, iterations=iterations, step_width0=step, mode=mode, **kw)
        assert (lit == exp).all()
        assert (got == exp).all(), (iterations, step, int((got != exp).sum()))
        assert W * H < 10000 or (max(redone) < W * H // 4 and redone[1] > 0), redone                  # a few per cent are redone


def test_verified_pass_leaves_room(vrt, engine):
    """1080p of hostile input: with an EIGHTH of the guard the pair still reproduces the exact kernels -- the bound is not
    within a factor of eight of what the arithmetic does (evidence, not proof: the proof is the header's)."""
    import torch
    W, H = 1920, 1080
    rng = np.random.default_rng(5)
    color, nrm, pos = _hostile_gbuffer(rng, W, H)
    dev = engine.torch_device
    c, n, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (color, nrm, pos))
    st = vrt.VoxelRenderSettings(targetResolution=(W, H))
    st.denoiserSettings.iterations = 3
    engine.set_option("denoise_verified", 0)
    try:
        lit = vrt.DenoiserStage(engine, st).record(c, n, p).cpu().numpy().copy()
    finally:
        engine.set_option("denoise_verified", 1)
    stage = vrt.DenoiserStage(engine, st)
    engine.set_option("denoise_count", 1)
    try:
        got = stage.record(c, n, p).cpu().numpy().copy()
        full = [stage.redone(i) for i in (1, 2)]
        assert (got == lit).all(), int((got != lit).sum())
        engine.set_option("denoise_guard_div8", 1)
        got8 = stage.record(c, n, p).cpu().numpy().copy()
        eighth = [stage.redone(i) for i in (1, 2)]
    finally:
        engine.set_option("denoise_guard_div8", 0); engine.set_option("denoise_count", 0)
    print("guards", stage.guard(1), stage.guard(2), "redone", full, "with guard / 8", eighth, "of", W * H, "differing", int((got8 != lit).sum()))
    assert (got8 == lit).all(), int((got8 != lit).sum())
    assert all(0 < e < f for e, f in zip(eighth, full))


def test_verified_pass_fourth_components_appear_late(vrt, oracle, engine):
    """Colour alpha and position w are 0 in what K1 writes, and the rows of k_denoise_ver leave their sums out until a
    texel with one enters the ring: a frame that has a single such texel far down a column strip, in each of the planes."""
    import torch
    W, H = 150, 230
    rng = np.random.default_rng(77)
    dev = engine.torch_device
    for which in ("alpha", "w", "none"):
        color = rng.integers(0, 256, (H, W, 4), dtype=np.uint8); color[..., 3] = 0
        nrm = rng.choice(np.array([-127, 0, 90, 127], np.int8), (H, W, 4)); nrm[..., 3] = 0
        pos = (np.round(rng.uniform(0, 32, (H, W, 4)) * 4) / 4).astype(np.float32); pos[..., 3] = 0.0
        if which == "alpha": color[141, 70, 3] = 200
        if which == "w": pos[171, 20, 3] = 0.5; pos[3, 3, 3] = -0.0
        c, n, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (color, nrm, pos))
        st = vrt.VoxelRenderSettings(targetResolution=(W, H))
        st.denoiserSettings.iterations = 3
        got = vrt.DenoiserStage(engine, st).record(c, n, p).cpu().numpy()
        exp = oracle.denoise(color, nrm, pos, iterations=3)
        assert (got == exp).all(), (which, int((got != exp).sum()))


def test_verified_pass_fuzz_sweep(vrt, oracle, engine):
    """60 random cases of tests/fuzz_denoise.py (sizes, phis over the admitted range, step widths, modes, pass counts, G-buffer kinds)."""
    import fuzz_denoise
    verified, redone = fuzz_denoise.run(vrt, oracle, engine, 60, seed=2026)
    assert verified > 40                                    # most passes of the sweep do take the verified form


def test_verified_pass_when_every_pixel_is_redone(vrt, oracle, engine):
    """NaN positions everywhere: the cheap mean of every pixel is NaN, the list of a workgroup overflows (1 024 entries for 64 x 32
    pixels) and the workgroup evaluates its whole segment literally."""
    import torch
    W, H = 190, 90
    rng = np.random.default_rng(3)
    color = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    nrm = rng.choice(np.array([-127, 0, 127], np.int8), (H, W, 4))
    pos = np.full((H, W, 4), np.nan, np.float32)
    dev = engine.torch_device
    c, n, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (color, nrm, pos))
    for mode in (0, 1):
        st = vrt.VoxelRenderSettings(targetResolution=(W, H))
        st.denoiserSettings.iterations = 3; st.denoiserSettings.mode = mode
        stage = vrt.DenoiserStage(engine, st)
        engine.set_option("denoise_count", 1)
        try:
            got = stage.record(c, n, p).cpu().numpy()
            assert stage.redone(1) == W * H and stage.redone(2) == W * H
        finally:
            engine.set_option("denoise_count", 0)
        exp = oracle.denoise(color, nrm, pos, iterations=3, mode=mode)
        assert (got == exp).all(), int((got != exp).sum())


@pytest.mark.parametrize("size", [(200, 120), (333, 61), (59, 9), (1, 1), (117, 400)])
def test_pair_kernel_is_the_ver_kernel(vrt, oracle, engine, size):
    """Round 4: k_denoise_pair (every weight computed once, R waves per workgroup, 64 - 2 R output columns per strip) against
    k_denoise_ver (context option denoise_pair = 0) and the oracle, on hostile G-buffers: tap offsets 2 .. 5, the exact mode and
    VRT_DENOISE_FAST (the same cheap arithmetic in both kernels: bit-identical there too), the same pixels evaluated twice."""
    import torch
    W, H = size
    rng = np.random.default_rng(W * 77 + H)
    color, nrm, pos = _hostile_gbuffer(rng, W, H)
    dev = engine.torch_device
    c, n, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (color, nrm, pos))
    for iterations, step in ((2, 1.0), (3, 2.0), (2, 3.0), (2, 4.0), (3, 1.0)):              # weighted passes with offsets 2; 3, 5; 4; 5; 2, 3
        for mode in (0, vrt.DENOISE_FAST):
            st = vrt.VoxelRenderSettings(targetResolution=(W, H))
            d = st.denoiserSettings
            d.iterations, d.stepWidth, d.mode = iterations, step, mode
            out, redone = {}, {}
            for pair in (1, 0):
                with engine.options(denoise_pair=pair, denoise_count=1):
                    stage = vrt.DenoiserStage(engine, st)
                    out[pair] = stage.record(c, n, p).cpu().numpy().copy()
                    redone[pair] = [stage.redone(i) for i in range(iterations)]
            assert (out[1] == out[0]).all(), (iterations, step, mode, int((out[1] != out[0]).sum()))
            assert redone[1] == redone[0], (iterations, step, mode, redone)
            if mode == 0:
                exp = oracle.denoise(color, nrm, pos, iterations=iterations, step_width0=step, mode=0)
                assert (out[1] == exp).all(), (iterations, step, int((out[1] != exp).sum()))


@pytest.mark.parametrize("size", [(200, 120), (61, 333), (62, 8), (63, 9), (1, 1), (125, 7)])
def test_pass0_kernel_is_the_ver_kernel(vrt, oracle, engine, size):
    """Round 4: k_denoise_p0 (pass 0, a wave to itself: 62 output columns, neighbours through wave_shr / wave_shl, the kernel taken
    separably) against k_denoise_ver<.., PASS0> (context option denoise_p0 = 0) and the oracle, hostile colours with and without
    alpha; strips of exactly 62 and 63 columns."""
    import torch
    W, H = size
    rng = np.random.default_rng(W * 131 + H)
    color, nrm, pos = _hostile_gbuffer(rng, W, H)
    dev = engine.torch_device
    for alpha in (True, False):
        if not alpha: color = color.copy(); color[..., 3] = 0
        c, n, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (color, nrm, pos))
        for iterations in (1, 2):
            st = vrt.VoxelRenderSettings(targetResolution=(W, H))
            st.denoiserSettings.iterations = iterations
            out = {}
            for p0 in (1, 0):
                with engine.options(denoise_p0=p0):
                    out[p0] = vrt.DenoiserStage(engine, st).record(c, n, p).cpu().numpy().copy()
            exp = oracle.denoise(color, nrm, pos, iterations=iterations)
            assert (out[1] == exp).all(), (alpha, iterations, int((out[1] != exp).sum()))
            assert (out[0] == exp).all(), (alpha, iterations, int((out[0] != exp).sum()))
