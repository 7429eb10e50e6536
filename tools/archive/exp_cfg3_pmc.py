"""Ad-hoc: config 3 (shadow ray) frames for a --pmc SQ_INSTS_VALU run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), pal, sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
st = vrt.VoxelRenderSettings(targetResolution=(1920, 1080))
st.fsrSetttings.enable = False
st.occlusionSettings.numSamples = int(os.environ.get("AO", "0"))
st.traceSettings.maxReflections = int(os.environ.get("BOUNCES", "0"))
st.denoiserSettings.enable = False
r = vrt.VoxelRenderer(eng, st, sc)
r.camera.position = np.array((128.0, 128.0, -204.8), np.float32)
for _ in range(3):
    r.render(); eng.synchronize()
print(eng.last_timings())
