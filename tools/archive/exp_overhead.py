"""Ad-hoc: K1 time with the traversal budget forced to 1 (ray gen + setup + epilogue + stores only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(), sky=vrt.synthetic.sky_gradient(512, 256))
res = (1920, 1080)
cam = vrt.CameraController(position=(128.0, 128.0, -204.8))
push = vrt.make_push(cam, (256, 256, 256), res)
def run(planes, ms, trav="DF", reps=10):
    st = vrt.VoxelRenderSettings.primary_only(res, getattr(vrt, "TRAVERSAL_" + trav))
    st.traceSettings.maxRaySteps = ms
    gb = vrt.GeometryBuffer(eng, res[0], res[1], planes)
    stc = st.to_c(); fr = gb.to_c()
    ts = []
    for _ in range(reps):
        vrt._capi.check(vrt.lib().vrt_render_geometry(eng.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
        eng.synchronize(); ts.append(eng.last_timings()["primary_ms"] * 1e3)
    return round(min(ts), 1)
allp = vrt.host.GBUFFER_PLANES
print("max_steps=1  all 6 planes:", run(allp, 1), " color only:", run(("color8",), 1), " no planes:", run((), 1))
print("max_steps=512 all 6 planes:", run(allp, 512), " color only:", run(("color8",), 512), " no planes:", run((), 512))
