"""Ad-hoc: stage timings at 1080p on the treehouse stand-in for BASELINE configs 2/3 and the reference defaults."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sc = vrt.VoxelScene.from_dense(eng, vol, pal, sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
res = (1920, 1080)
def run(name, ao, shadows, bounces, iters, trav="AUTO", reps=6, split=False):
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = ao
    st.traceSettings.shadows = shadows
    st.traceSettings.maxReflections = bounces
    st.traceSettings.traversal = getattr(vrt, "TRAVERSAL_" + trav)
    st.traceSettings.splitKernels = split
    st.denoiserSettings.enable = iters > 0
    st.denoiserSettings.iterations = max(iters, 1)
    r = vrt.VoxelRenderer(eng, st, sc)
    r.camera.position = np.array([128.0, 128.0, -204.8], np.float32)
    best = None
    for _ in range(reps):
        r.render(); eng.synchronize()
        t = eng.last_timings()
        if best is None or t["geometry_ms"] < best["geometry_ms"]:
            best = t
    rays = None
    print(f"{name:34s} {trav:8s} K1 {best['primary_ms']*1e3:8.1f} us  K1+K2 {best['geometry_ms']*1e3:9.1f} us  K3 {best['denoise_ms']*1e3:8.1f} us", flush=True)
for trav in sys.argv[1:] or ["AUTO"]:
    run("config2 primary only", 0, False, 0, 0, trav)
    run("config3 +shadow +1 denoise pass", 0, True, 0, 1, trav)
    run("config3 +shadow +2 denoise passes", 0, True, 0, 2, trav)
    run("AO4 + shadow, no bounce, 2 passes", 4, True, 0, 2, trav)
    run("reference defaults (AO4,5 bounces)", 4, True, 5, 2, trav)
    run("config3 split kernels", 0, True, 0, 2, trav, split=True)
    run("defaults split kernels", 4, True, 5, 2, trav, split=True)
