"""Blit (blit.frag), accumulation and the jittered temporal frame graph through the C-ABI vs the oracle: bit-exact."""
import ctypes as C

import numpy as np
import pytest

from helpers import metallic_palette

pytestmark = pytest.mark.gpu


def _rand_rgba(w, h, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    img[h // 3: h // 2, w // 4: w // 2] = (255, 0, 255, 255)             # flat block + hard edges
    img[0, :] = 0
    img[:, -1] = 255
    return img


@pytest.mark.parametrize("sw,sh,tw,th", [
    (96, 54, 96, 54),        # identity
    (113, 63, 192, 108),     # BALANCED-style upscale, odd source
    (64, 36, 192, 108),      # 3x upscale
    (192, 108, 96, 54),      # downscale
    (160, 90, 120, 120),     # centre crop: wider source into square window
    (90, 160, 200, 100),     # centre crop: taller source into wide window
    (1, 1, 33, 17),          # single texel
    (37, 5, 3, 29),          # extreme aspect change
])
def test_blit_bit_exact(vrt, oracle, engine, sw, sh, tw, th):
    import torch
    src = _rand_rgba(sw, sh, seed=sw * 131 + th)
    d_src = torch.from_numpy(src).to(engine.torch_device)
    d_dst = torch.zeros((th, tw, 4), dtype=torch.uint8, device=engine.torch_device)
    vrt._capi.check(vrt.lib().vrt_blit(engine.ctx, d_src.data_ptr(), sw, sh, d_dst.data_ptr(), tw, th))
    engine.synchronize()
    got, exp = d_dst.cpu().numpy(), oracle.blit(src, tw, th)
    assert (got == exp).all(), (int((got != exp).sum()), np.argwhere(got != exp)[0].tolist())
    if (sw, sh) == (tw, th):
        assert (got == src).all()                                        # texel centres: identity copy


def test_blit_rejects_bad_arguments(vrt, engine):
    import torch
    t = torch.zeros((4, 4, 4), dtype=torch.uint8, device=engine.torch_device)
    L = vrt.lib()
    assert L.vrt_blit(engine.ctx, t.data_ptr(), 4, 4, t.data_ptr(), 4, 4) != 0
    assert L.vrt_blit(engine.ctx, None, 4, 4, t.data_ptr(), 4, 4) != 0
    assert L.vrt_blit(engine.ctx, t.data_ptr(), 0, 4, t.data_ptr() + 64, 4, 4) != 0
    assert L.vrt_resolve(engine.ctx, t.data_ptr(), t.data_ptr(), 4, 4, 0) != 0


@pytest.mark.parametrize("frames", [1, 2, 3, 7, 32])
def test_accumulate_resolve_exact_mean(vrt, engine, frames):
    import torch
    W, H = 150, 67
    rng = np.random.default_rng(frames)
    seq = rng.integers(0, 256, size=(frames, H, W, 4), dtype=np.uint8)
    seq[:, 0, 0] = 255
    seq[:, 0, 1] = 0
    acc = torch.full((H, W, 4), 0x7fffffff, dtype=torch.int32, device=engine.torch_device)   # garbage: reset must clear it
    out = torch.zeros((H, W, 4), dtype=torch.uint8, device=engine.torch_device)
    L = vrt.lib()
    for k in range(frames):
        d = torch.from_numpy(seq[k]).to(engine.torch_device)
        vrt._capi.check(L.vrt_accumulate(engine.ctx, d.data_ptr(), acc.data_ptr(), W, H, 1 if k == 0 else 0))
    vrt._capi.check(L.vrt_resolve(engine.ctx, acc.data_ptr(), out.data_ptr(), W, H, frames))
    engine.synchronize()
    s = seq.astype(np.int64).sum(axis=0)
    assert (acc.cpu().numpy().astype(np.int64) == s).all()
    exp = ((2 * s + frames) // (2 * frames)).astype(np.uint8)
    assert (out.cpu().numpy() == exp).all()
    assert (out.cpu().numpy()[0, 0] == 255).all() and (out.cpu().numpy()[0, 1] == 0).all()


def test_temporal_frame_graph_vs_oracle(vrt, oracle, engine):
    """geometry (jittered, frame-rotated noise) -> denoise -> accumulate -> resolve -> upscale -> window blit, four frames,
    against the oracle run frame by frame with the same push constants."""
    target = (160, 96)
    vol = vrt.synthetic.floating_cubes(40, seed=9, count=50)
    pal = metallic_palette(vrt)
    sky, noise = vrt.synthetic.sky_gradient(64, 32), vrt.synthetic.blue_noise_standin(64)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    st = vrt.VoxelRenderSettings(targetResolution=target)
    st.fsrSetttings.scaling = vrt.FsrScaling.QUALITY                     # render at 106 x 64
    st.occlusionSettings.numSamples = 2
    r = vrt.VoxelRenderer(engine, st, sc, temporal=True, windowSize=(120, 120))
    r.camera.position = np.array([20.3, 20.2, -30.0], np.float32)
    r.camera.updateDirectionVectors()
    RW, RH = st.renderResolution()
    assert (RW, RH) == (106, 64)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    pr = oracle.params_from(st.to_c())
    acc = np.zeros((RH, RW, 4), np.int64)
    jitters = set()
    for f in range(4):
        r.update(0.0)                                                    # advances jitter + frame, camera stays
        push = r.push_constants()
        assert push.frame == f + 1
        jitters.add((push.camera_jitter[0], push.camera_jitter[1]))
        got = r.render()
        engine.synchronize()
        fr = oracle.render(osn, push, pr, planes=["color8", "normal8", "position"])
        den = oracle.denoise(fr["color8"], fr["normal8"], fr["position"], iterations=st.denoiserSettings.iterations,
                             phi_color0=st.denoiserSettings.phiColor0, phi_normal0=st.denoiserSettings.phiNormal0,
                             phi_pos0=st.denoiserSettings.phiPos0, step_width0=st.denoiserSettings.stepWidth)
        acc += den
        mean = ((2 * acc + (f + 1)) // (2 * (f + 1))).astype(np.uint8)
        exp = oracle.blit(oracle.blit(mean, target[0], target[1]), 120, 120)
        g = got.cpu().numpy()
        assert g.shape == (120, 120, 4)
        assert (g == exp).all(), (f, int((g != exp).sum()))
    assert len(jitters) == 4
    # reset drops the history: the next frame equals a single-frame render
    r.upscaler.reset()
    got = r.render().cpu().numpy()
    engine.synchronize()
    fr = oracle.render(osn, r.push_constants(), pr, planes=["color8", "normal8", "position"])
    den = oracle.denoise(fr["color8"], fr["normal8"], fr["position"])
    assert (got == oracle.blit(oracle.blit(den, target[0], target[1]), 120, 120)).all()


def test_jitter_changes_the_image_subpixel(vrt, oracle, engine):
    """cameraJitter enters rayDir in world x / y (voxel_volume.frag:319): a jittered frame differs from the unjittered
    one, but only along silhouettes (sub-pixel shift), and matches the oracle bit for bit."""
    res = (128, 80)
    vol = vrt.synthetic.floating_cubes(40, seed=3, count=40)
    pal = vrt.synthetic.default_palette()
    sc = vrt.VoxelScene.from_dense(engine, vol, pal)
    st = vrt.VoxelRenderSettings.primary_only(res)
    r = vrt.VoxelRenderer(engine, st, sc)
    r.camera.position = np.array([20.3, 20.2, -30.0], np.float32)
    r.camera.updateDirectionVectors()
    base = r.render().cpu().numpy().copy()
    r.jitter = (0.25, -0.3888889)
    jit = r.render().cpu().numpy().copy()
    engine.synchronize()
    osn = oracle.OracleScene(vol, pal)
    exp = oracle.render(osn, r.push_constants(), oracle.params_from(st.to_c()), planes=["color8"])["color8"]
    assert (jit == exp).all()
    changed = (jit != base).any(axis=2).mean()
    assert 0.0 < changed < 0.25
