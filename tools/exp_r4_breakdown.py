"""Where the megakernel's instructions go, by ray class: the same frame with the secondary rays switched on one class at a
time (primary only / + shadow / + AO / + AO + shadow / + bounces) on the three scenes of the secondary-ray configurations
(treehouse 256^3 at 1080p, Mandelbulb 512^3 at 4K, the 2048^3 brick scene at 4K); N frames per variant, one per launch.

  python3 tools/exp_r4_breakdown.py [scenes=tmb] [n=5] [name=value ...]      (context options)
  rocprofv3 --pmc SQ_INSTS_VALU ... -- python3 tools/exp_r4_breakdown.py     (tools/exp_r4_breakdown_pmc.py attributes the
                                                                             k_primary dispatches to the variants by order)
Prints one line per variant: VARIANT <scene> <name> launches=<n> geometry_us=<median>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import voxel_raytracing_amd as vrt

scenes, n = "tmb", 5
eng = vrt.Engine(0)
for a in sys.argv[1:]:
    if "=" not in a: continue
    k, v = a.split("=")
    if k == "scenes": scenes = v
    elif k == "n": n = int(v)
    elif k == "stats": pass
    else: eng.set_option(k, int(v))
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sky, noise = vrt.synthetic.sky_gradient(512, 256), vrt.synthetic.blue_noise_standin(512)
VARIANTS = (("primary", 0, False, 0), ("shadow", 0, True, 0), ("ao4", 4, False, 0), ("ao4+shadow", 4, True, 0), ("full", 4, True, None),
            ("bounces-only", 0, False, None))       # (the last: bounce rays without any AO / shadow ray -- what the chain's main rays cost)


def run(tag, scene, res, dims, pos, max_steps, bounces, frame):
    eng.set_timing(True)
    push = vrt.make_push(vrt.CameraController(position=pos), dims, res, frame=frame)
    for name, ao, sh, b in VARIANTS:
        st = vrt.VoxelRenderSettings(targetResolution=res)
        st.fsrSetttings.enable = False
        st.denoiserSettings.enable = False
        st.occlusionSettings.numSamples = ao
        st.traceSettings.shadows = sh
        st.traceSettings.maxReflections = bounces if b is None else b
        st.traceSettings.maxRaySteps = max_steps
        geo = vrt.GeometryStage(eng, st, scene)
        t = []
        for _ in range(n):
            geo.record(push); eng.synchronize()
            t.append(eng.last_timings()["geometry_ms"])
        t = sorted(t[1:]) if n > 1 else t
        print(f"VARIANT {tag} {name} launches={n} geometry_us={t[len(t) // 2] * 1e3:.1f}", flush=True)
        if name == "full" and "stats" in sys.argv[1:]:
            # how the chain's rays sit in the waves: pixels whose primary hit reflects, per 8 x 8 tile (a wave of K1)
            gb = vrt.GeometryBuffer(eng, res[0], res[1], ["rays_total", "hit_id"])
            stc, fr = st.to_c(), gb.to_c()
            import ctypes as C
            vrt._capi.check(vrt.lib().vrt_render_geometry(eng.ctx, scene.handle, C.byref(push), C.byref(stc), C.byref(fr), None)); eng.synchronize()
            g = gb.numpy()
            hit = g["hit_id"] != 0
            metal = g["hit_id"] >= 200
            H8, W8 = res[1] // 8 * 8, res[0] // 8 * 8
            tm = metal[:H8, :W8].reshape(H8 // 8, 8, W8 // 8, 8).sum((1, 3))
            th = hit[:H8, :W8].reshape(H8 // 8, 8, W8 // 8, 8).sum((1, 3))
            print(f"STATS {tag}: hit {hit.mean():.3f} of the pixels, reflecting {metal.mean():.3f}; waves with a hit {np.mean(th > 0):.3f}, with a reflecting hit {np.mean(tm > 0):.3f}, "
                  f"reflecting lanes in those {tm[tm > 0].mean():.1f} of 64; rays per pixel {g['rays_total'].mean():.2f}", flush=True)


if "t" in scenes:
    sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), pal, sky=sky, noise=noise)
    pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
    run("treehouse1080p", sc, (1920, 1080), (256, 256, 256), pos0, 512, 5, 0)
    sc.destroy()
if "m" in scenes:
    scm = vrt.VoxelScene.from_dense(eng, vrt.synthetic.mandelbulb(512), pal, sky=sky, noise=noise)
    run("mandelbulb4k", scm, (3840, 2160), (512, 512, 512), (512 * 0.5 + 0.3, 512 * 0.5 + 0.2, -0.45 * 512), 512, 2, 5)
    scm.destroy()
if "b" in scenes:
    grid, pool = vrt.synthetic.sparse_brick_scene(2048, 0.015, seed=5)
    sb = vrt.VoxelScene.from_bricks(eng, grid, pool, pal, sky=sky, noise=noise)
    pos5, yaw5, pitch5 = vrt.synthetic.default_camera_for(2048, 2048, 2048)
    run("bricks4k", sb, (3840, 2160), (2048, 2048, 2048), (pos5[0] + 0.3, pos5[1] + 0.2, pos5[2]), 6144, 4, 17)
    sb.destroy()
