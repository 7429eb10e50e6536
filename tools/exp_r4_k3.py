"""K3 A/B: the reference's two denoiser passes (and three) at 1080p on a rendered G-buffer, through the context's events, with the
options given as name=value arguments (denoise_pair=0/1, denoise_pair_wgs=n ...).  Prints the median of 20 calls per variant and
checks the variants' outputs against each other (the verified pass is exact whichever kernel runs it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
eng.set_timing(True)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sky, noise = vrt.synthetic.sky_gradient(512, 256), vrt.synthetic.blue_noise_standin(512)
sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), pal, sky=sky, noise=noise)
variants = [a for a in sys.argv[1:]] or ["denoise_pair=0", "denoise_pair=1"]
ref = {}
for res in ((1920, 1080), (3840, 2160)):
    for iters, mode in ((2, 0), (3, 0), (2, vrt.DENOISE_FAST)):
        for ao in (0, 4):
            for v in variants:
                opts = dict((k, int(x)) for k, x in (kv.split("=") for kv in v.split(",")))
                with eng.options(**opts):
                    st = vrt.VoxelRenderSettings(targetResolution=res)
                    st.fsrSetttings.enable = False
                    st.occlusionSettings.numSamples = ao
                    st.traceSettings.maxReflections = 0
                    st.denoiserSettings.enable = True
                    st.denoiserSettings.iterations = iters
                    st.denoiserSettings.mode = mode
                    r = vrt.VoxelRenderer(eng, st, sc)
                    r.camera.position = np.array((128.0, 128.0, -204.8), np.float32)
                    td = []
                    for _ in range(20):
                        out = r.render(); eng.synchronize()
                        td.append(eng.last_timings()["denoise_ms"])
                    img = out.cpu().numpy().copy()
                    cnt = ""
                    if mode == 0:
                        with eng.options(denoise_count=1):
                            r.render(); eng.synchronize()
                            cnt = " redone=" + str([r._denoiserStage.redone(i) for i in range(iters)])
                    key = (res, iters, mode, ao)
                    same = "-"
                    if key in ref: same = "identical" if np.array_equal(ref[key], img) else f"DIFFERENT ({int((ref[key] != img).sum())} bytes)"
                    else: ref[key] = img
                    print(f"K3 {res[0]}x{res[1]} passes={iters} mode={mode} ao={ao} {v}: denoise_us={sorted(td)[10] * 1e3:.1f} min={min(td) * 1e3:.1f} {same}{cnt}", flush=True)
sc.destroy()
