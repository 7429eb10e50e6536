"""K3 on a rank's strips: the reference's two denoiser passes at 1080p over rank 0's rows of an N-rank split (16-row strips dealt
round-robin, and one band per rank), k_denoise_pair against k_denoise_ver (denoise_pair = 0), by the context's events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import voxel_raytracing_amd as vrt
from helpers import camera_push
eng = vrt.Engine(0)
eng.set_timing(True)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), pal, sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
res = (1920, 1080)
st = vrt.VoxelRenderSettings(targetResolution=res); st.fsrSetttings.enable = False; st.occlusionSettings.numSamples = 4
gb = vrt.GeometryStage(eng, st, sc).record(camera_push(vrt, (256, 256, 256), res)); eng.synchronize()
st.denoiserSettings.iterations = 2
for N, rows in ((1, 16), (2, 16), (8, 16), (8, 144), (4, 272), (2, 544)):
    for pair in (0, 1):
        with eng.options(denoise_pair=pair):
            den = vrt.DenoiserStage(eng, st)
            sh = vrt._capi.Shard(0, N, rows) if N > 1 else None
            t = []
            for _ in range(20):
                den.record(gb.color, gb.normal, gb.position, sh); eng.synchronize()
                t.append(eng.last_timings()["denoise_ms"])
            print(f"K3SHARD ranks={N} strip_rows={rows} denoise_pair={pair}: {sorted(t)[10] * 1e3:.1f} us (min {min(t) * 1e3:.1f})", flush=True)
