"""Ad-hoc experiment: K1 time vs step budget / view axis / resolution (diagnosing the latency floor)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import voxel_raytracing_amd as vrt

eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
pal = vrt.synthetic.default_palette()
sc = vrt.VoxelScene.from_dense(eng, vol, pal, sky=vrt.synthetic.sky_gradient(512, 256))

def run(res, trav, max_steps=512, yaw=90.0, pos=(128.0, 128.0, -204.8), reps=8):
    st = vrt.VoxelRenderSettings.primary_only(res, getattr(vrt, "TRAVERSAL_" + trav))
    st.traceSettings.maxRaySteps = max_steps
    stage = vrt.GeometryStage(eng, st, sc)
    cam = vrt.CameraController(position=pos, yaw=yaw, pitch=0.0)
    push = vrt.make_push(cam, (256, 256, 256), res)
    ts = []
    for _ in range(reps):
        stage.record(push); eng.synchronize()
        ts.append(eng.last_timings()["primary_ms"] * 1e3)
    return min(ts), float(np.median(ts))

for trav in sys.argv[1:] or ["DENSE"]:
    for res in [(480, 270), (1920, 1080)]:
        for ms in (16, 64, 128, 256, 512):
            print(trav, res, "max_steps", ms, "z-view us(min,med)", run(res, trav, ms), flush=True)
        print(trav, res, "x-view (yaw 0, from -x)", run(res, trav, 512, yaw=0.0, pos=(-204.8, 128.0, 128.0)), flush=True)
