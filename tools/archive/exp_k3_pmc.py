"""Ad-hoc: ten frames of config 3's two denoiser passes, exact and VRT_DENOISE_FAST, for a --pmc pass over the K3 kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), pal, sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
for mode in (0, vrt.DENOISE_FAST):
    st = vrt.VoxelRenderSettings(targetResolution=(1920, 1080))
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = 0
    st.denoiserSettings.enable = True; st.denoiserSettings.iterations = 2; st.denoiserSettings.mode = mode
    r = vrt.VoxelRenderer(eng, st, sc)
    r.camera.position = np.array((128.0, 128.0, -204.8), np.float32)
    for _ in range(6):
        r.render(); eng.synchronize()
    print(mode, eng.last_timings())
