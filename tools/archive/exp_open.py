"""Ad-hoc: open-cell termination (fields with 0 where a ray's octant is empty to the volume's corner) against fields without:
planes identical?  stage timings at 1080p on the treehouse stand-in."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
def scene(open_cells):
    if open_cells: os.environ.pop("VRT_OPEN_CELLS", None)
    else: os.environ["VRT_OPEN_CELLS"] = "0"
    return vrt.VoxelScene.from_dense(eng, vol, pal, sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
scenes = {"closed": scene(False), "open": scene(True)}
res = (1920, 1080)
PL = vrt.host.GBUFFER_PLANES + ("color_f", "hit_id", "hit_mask", "rays_total")
def settings(ao, shadows, bounces):
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = ao
    st.traceSettings.shadows = shadows
    st.traceSettings.maxReflections = bounces
    st.denoiserSettings.enable = False
    return st
cams = [(128.0, 128.0, -204.8, 90.0, 0.0), (40.0, 200.0, -60.0, 70.0, -25.0), (128.0, 60.0, 128.0, 30.0, 10.0), (300.0, 300.0, 300.0, 225.0, -35.0),
        (128.0, 300.0, 128.0, 90.0, -89.0), (-6.0, 128.0, 100.0, 10.0, 5.0), (600.0, 140.0, -500.0, 128.0, -3.0)]
for name, ao, sh, bo in [("primary only", 0, False, 0), ("config 3 (shadow)", 0, True, 0), ("defaults (AO4, shadow, 5 bounces)", 4, True, 5)]:
    st = settings(ao, sh, bo)
    for ci, (x, y, z, yaw, pitch) in enumerate(cams):
        cam = vrt.CameraController(position=(x, y, z), yaw=yaw, pitch=pitch) if ci else vrt.CameraController(position=(x, y, z))
        push = vrt.make_push(cam, (256, 256, 256), res)
        out, tm = {}, {}
        for k, sc in list(scenes.items()) + [("tags", scenes["open"])]:
            os.environ["VRT_TILE_TAGS"] = "1" if k == "tags" else "0"
            gb = vrt.GeometryBuffer(eng, res[0], res[1], PL)
            stc = st.to_c(); fr = gb.to_c()
            ts = []
            for _ in range(6):
                vrt._capi.check(vrt.lib().vrt_render_geometry(eng.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
                eng.synchronize(); ts.append(eng.last_timings()["geometry_ms"] * 1e3)
            tm[k] = min(ts); out[k] = gb.numpy()
        bad = [p for p in PL if not np.array_equal(out["closed"][p], out["open"][p]) or not np.array_equal(out["closed"][p], out["tags"][p])]
        print(f"{name:36s} cam {ci}: closed {tm['closed']:7.1f} us  open {tm['open']:7.1f} us  + tile tags {tm['tags']:7.1f} us  hit {float((out['open']['hit_id'] != 0).mean()):.3f}  mismatching planes: {bad}", flush=True)
