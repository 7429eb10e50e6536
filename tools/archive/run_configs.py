"""The secondary-ray and brick configurations as a plain frame loop for rocprofv3 (tools/profile_r2.sh): config 3 (shadow ray, 2
denoiser passes: the verified pair, VRT_DENOISE_FAST, and the exact kernels computing every pixel), the reference defaults (AO 4, shadow, <= 5 bounces), and BASELINE configs[4]
(2048^3 brick scene at 3840x2160) -- 10 frames each, the device idle between frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sky, noise = vrt.synthetic.sky_gradient(512, 256), vrt.synthetic.blue_noise_standin(512)
sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), pal, sky=sky, noise=noise)
def loop(scene, res, dims, ao, bounces, iters, mode, max_steps, pos, n=10):
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
    for _ in range(n):
        r.render(); eng.synchronize()
    print(res, ao, bounces, iters, mode, eng.last_timings(), flush=True)
loop(sc, (1920, 1080), 256, 0, 0, 2, 0, 512, (128.0, 128.0, -204.8))
loop(sc, (1920, 1080), 256, 0, 0, 2, vrt.DENOISE_FAST, 512, (128.0, 128.0, -204.8))
eng.set_option("denoise_verified", 0)                    # the exact kernels computing every pixel (what the verified pass replaced)
loop(sc, (1920, 1080), 256, 0, 0, 2, 0, 512, (128.0, 128.0, -204.8))
eng.set_option("denoise_verified", 1)
loop(sc, (1920, 1080), 256, 4, 5, 2, 0, 512, (128.0, 128.0, -204.8))
grid, pool = vrt.synthetic.sparse_brick_scene(2048, 0.015, seed=5)
sb = vrt.VoxelScene.from_bricks(eng, grid, pool, pal, sky=sky, noise=noise)
loop(sb, (3840, 2160), 2048, 4, 4, 0, 0, 6144, (1024.3, 1024.2, -1638.4), n=5)
sb.destroy()
# BASELINE configs[3]: Mandelbulb 512^3, 3840x2160, 2 bounces, AO 4, shadow ray -- five frames, one per launch
scm = vrt.VoxelScene.from_dense(eng, vrt.synthetic.mandelbulb(512), pal, sky=sky, noise=noise)
loop(scm, (3840, 2160), 512, 4, 2, 0, 0, 512, (512 * 0.5 + 0.3, 512 * 0.5 + 0.2, -0.45 * 512), n=5)
