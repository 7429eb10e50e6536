"""The workloads of bench.py's extra_configs as plain frame loops, one per launch, for rocprofv3 (tools/profile_r4.sh).  Every
workload prints `VARIANT <tag> geometry=<n launches> denoise_passes=<per frame>`; tools/profile_r4_collect.py attributes the
dispatches of a pass to the workloads BY ORDER, so that every PMC summary under profiles/ is one workload's (round 3's
megakernel summary averaged config 3 and the reference defaults).
   config3     treehouse 256^3, 1080p, shadow ray, 2 denoiser passes (verified pass; then VRT_DENOISE_FAST, then every pixel literally)
   defaults    the reference's defaults: AO 4, shadow, <= 5 bounces, 2 passes
   mandelbulb  BASELINE configs[3]: Mandelbulb 512^3, 4K, 2 bounces, AO 4, shadow
   brick       BASELINE configs[4]: the 2048^3 brick scene, 4K, max_steps 6144, 4 bounces, AO 4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
for a in sys.argv[1:]:
    k, v = a.split("="); eng.set_option(k, int(v))
eng.set_timing(True)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sky, noise = vrt.synthetic.sky_gradient(512, 256), vrt.synthetic.blue_noise_standin(512)


def loop(tag, scene, res, ao, bounces, iters, mode, max_steps, pos, n, frame=0):
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = ao
    st.traceSettings.maxReflections = bounces
    st.traceSettings.maxRaySteps = max_steps
    st.denoiserSettings.enable = iters > 0
    st.denoiserSettings.iterations = max(iters, 1)
    st.denoiserSettings.mode = mode
    r = vrt.VoxelRenderer(eng, st, scene)
    r.camera.position = np.array(pos, np.float32)
    r.frameCount = frame
    tg, td = [], []
    for _ in range(n):
        r.render(); eng.synchronize()
        t = eng.last_timings(); tg.append(t["geometry_ms"]); td.append(t["denoise_ms"])
    tg, td = sorted(tg), sorted(td)
    print(f"VARIANT {tag} geometry={n} denoise_passes={iters} geometry_us={tg[n // 2] * 1e3:.1f} denoise_us={td[n // 2] * 1e3:.1f}", flush=True)


sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), pal, sky=sky, noise=noise)
p0 = (128.0, 128.0, -204.8)
loop("config3", sc, (1920, 1080), 0, 0, 2, 0, 512, p0, 10)
loop("config3_fast", sc, (1920, 1080), 0, 0, 2, vrt.DENOISE_FAST, 512, p0, 10)
eng.set_option("denoise_verified", 0)
loop("config3_literal", sc, (1920, 1080), 0, 0, 2, 0, 512, p0, 10)
eng.set_option("denoise_verified", 1)
loop("defaults", sc, (1920, 1080), 4, 5, 2, 0, 512, p0, 10)
sc.destroy()
scm = vrt.VoxelScene.from_dense(eng, vrt.synthetic.mandelbulb(512), pal, sky=sky, noise=noise)
loop("mandelbulb", scm, (3840, 2160), 4, 2, 0, 0, 512, (512 * 0.5 + 0.3, 512 * 0.5 + 0.2, -0.45 * 512), 5, frame=5)
scm.destroy()
grid, pool = vrt.synthetic.sparse_brick_scene(2048, 0.015, seed=5)
sb = vrt.VoxelScene.from_bricks(eng, grid, pool, pal, sky=sky, noise=noise)
loop("brick", sb, (3840, 2160), 4, 4, 0, 0, 6144, (1024.3, 1024.2, -1638.4), 5, frame=17)
sb.destroy()
