import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(), sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
res = (1920, 1080)
for ao, sh in ((0, True), (4, False), (4, True)):
    st = vrt.VoxelRenderSettings(targetResolution=res); st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = ao; st.traceSettings.shadows = sh; st.traceSettings.maxReflections = 0
    stage = vrt.GeometryStage(eng, st, sc, debug_planes=True)
    cam = vrt.CameraController(position=(128.0, 128.0, -204.8))
    gb = stage.record(vrt.make_push(cam, (256, 256, 256), res)); eng.synchronize()
    t = eng.last_timings()
    o = gb.numpy()
    sec = o["steps_total"].astype(np.int64) - o["steps_primary"].astype(np.int64)
    rays = o["rays_total"].astype(np.int64) - 1
    hit = o["hit_id"] != 0
    print(f"ao={ao} shadows={sh}: K2 {(t['geometry_ms']-t['primary_ms'])*1e3:.1f} us; hit px {hit.sum()}; secondary rays {rays.sum()}; secondary steps {sec.sum()} "
          f"({sec.sum()/max(rays.sum(),1):.1f}/ray); primary steps {o['steps_primary'].sum()}")
