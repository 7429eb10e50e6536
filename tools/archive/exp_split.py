"""Ad-hoc: the megakernel against the split form (K1 + K2 over the compacted hit list: full waves of hit pixels) on the
configurations whose secondary rays are bound by the vector unit rather than by their chains of look-ups -- BASELINE configs[3]
(Mandelbulb 512^3 at 3840x2160, 2 bounces, AO 4) and the reference defaults at 1080p.  Geometry ms per frame, one frame per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sky, noise = vrt.synthetic.sky_gradient(512, 256), vrt.synthetic.blue_noise_standin(512)
def run(scene, res, ao, bounces, pos, split, n=8):
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = ao
    st.traceSettings.maxReflections = bounces
    st.traceSettings.splitKernels = split
    st.denoiserSettings.enable = False
    r = vrt.VoxelRenderer(eng, st, scene)
    r.camera.position = np.array(pos, np.float32)
    ts = []
    for _ in range(n):
        img = r.render(); eng.synchronize(); ts.append(eng.last_timings()["geometry_ms"])
    return float(np.median(ts[2:])), r.gBuffer.numpy()["color8"].copy()
sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), pal, sky=sky, noise=noise)
for name, scene, res, ao, b, pos in (("defaults 1080p", sc, (1920, 1080), 4, 5, (128.0, 128.0, -204.8)),):
    a, ia = run(scene, res, ao, b, pos, False); s, isp = run(scene, res, ao, b, pos, True)
    print(f"{name}: megakernel {a*1e3:.1f} us, split {s*1e3:.1f} us, equal {bool((ia == isp).all())}", flush=True)
scm = vrt.VoxelScene.from_dense(eng, vrt.synthetic.mandelbulb(512), pal, sky=sky, noise=noise)
a, ia = run(scm, (3840, 2160), 4, 2, (512 * 0.5 + 0.3, 512 * 0.5 + 0.2, -0.45 * 512), False, n=5)
s, isp = run(scm, (3840, 2160), 4, 2, (512 * 0.5 + 0.3, 512 * 0.5 + 0.2, -0.45 * 512), True, n=5)
print(f"mandelbulb 4K: megakernel {a*1e3:.1f} us, split {s*1e3:.1f} us, equal {bool((ia == isp).all())}", flush=True)
