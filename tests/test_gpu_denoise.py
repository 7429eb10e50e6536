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
