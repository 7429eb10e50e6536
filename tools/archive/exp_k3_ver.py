"""Ad-hoc: config 3's denoiser passes at 1080p -- the exact kernels, the verified pair, VRT_DENOISE_FAST -- timed by the
context's events over single calls (ms per vrt_denoise call; pass 0 alone subtracted out with iterations = 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0); eng.set_option("denoise_count", int(os.environ.get("COUNT", "0")))
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), pal, sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
res = tuple(int(v) for v in os.environ.get("RES", "1920x1080").split("x"))
st = vrt.VoxelRenderSettings(targetResolution=res)
st.fsrSetttings.enable = False
st.occlusionSettings.numSamples = int(os.environ.get("AO", "0"))
stage = vrt.GeometryStage(eng, st, sc)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from helpers import camera_push
gb = stage.record(camera_push(vrt, (256, 256, 256), res)); eng.synchronize()
def run(iters, mode, verified):
    st.denoiserSettings.iterations = iters; st.denoiserSettings.mode = mode
    eng.set_option("denoise_verified", verified)
    den = vrt.DenoiserStage(eng, st)
    ts = []
    for _ in range(12):
        out = den.record(gb.color, gb.normal, gb.position); eng.synchronize()
        ts.append(eng.last_timings()["denoise_ms"])
    red = [den.redone(i) for i in range(iters)]
    return float(np.median(ts[2:])), out.cpu().numpy().copy(), red
for iters in (1, 2, 3):
    t_ex, a, _ = run(iters, 0, 0)
    t_ver, b, red = run(iters, 0, 1)
    t_fast, c, _ = run(iters, vrt.DENOISE_FAST, 1)
    t_fold, d, _ = run(iters, vrt.DENOISE_FAST, 0)
    print(f"iterations {iters}: exact {t_ex*1e3:.1f} us  verified {t_ver*1e3:.1f} us (redone {red}, equal {bool((a == b).all())})  "
          f"fast(new) {t_fast*1e3:.1f} us (max diff {int(np.abs(a.astype(int) - c.astype(int)).max())})  fast(old) {t_fold*1e3:.1f} us (max diff {int(np.abs(a.astype(int) - d.astype(int)).max())})", flush=True)
# back to back: 40 calls without a synchronisation in between (the kernel-trace shows whether a lone call pays for a cold start)
st.denoiserSettings.iterations = 2; st.denoiserSettings.mode = 0; eng.set_option("denoise_verified", 1)
den = vrt.DenoiserStage(eng, st)
import time
for rep in range(3):
    eng.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        den.record(gb.color, gb.normal, gb.position)
    eng.synchronize(); t1 = time.perf_counter()
    print(f"40 calls back to back: {(t1 - t0) / 40 * 1e6:.1f} us per call (two passes)", flush=True)
