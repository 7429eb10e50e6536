import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(), sky=vrt.synthetic.sky_gradient(512, 256))
res = (1920, 1080)
st = vrt.VoxelRenderSettings.primary_only(res, getattr(vrt, "TRAVERSAL_" + (sys.argv[1] if len(sys.argv) > 1 else "DF")))
gb = vrt.GeometryBuffer(eng, res[0], res[1], vrt.host.GBUFFER_PLANES + vrt.host.DEBUG_PLANES)
cam = vrt.CameraController(position=(128.0, 128.0, -204.8))
push = vrt.make_push(cam, (256, 256, 256), res)
stc = st.to_c(); stc.flags = 1
fr = gb.to_c()
vrt._capi.check(vrt.lib().vrt_render_geometry(eng.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
eng.synchronize()
o = gb.numpy()
outer, near, steps = o["steps_total"].astype(np.int64), o["rays_total"].astype(np.int64), o["steps_primary"].astype(np.int64)
H, W = outer.shape
Hp = (H + 7) // 8 * 8
def blk(a):
    p = np.zeros((Hp, W), np.int64); p[:H] = a
    return p.reshape(Hp // 8, 8, W // 8, 8)
wo = blk(outer).max(axis=(1, 3)); wn = blk(near).max(axis=(1, 3)); ws = blk(steps).max(axis=(1, 3))
live = ws > 0
print("waves", live.sum(), "per-wave mean: max-steps", ws[live].mean(), "outer iters", wo[live].mean(), "near iters", wn[live].mean())
print("totals: wave-steps", ws.sum(), "outer", wo.sum(), "near", wn.sum(), " max outer in a wave", wo.max(), "max steps", ws.max())
hist = np.bincount(np.minimum(wo[live], 399) // 20)
print("outer-iteration histogram per wave (bins of 20):", hist.tolist())
ratio = ws[live] / np.maximum(wo[live], 1)
print("mean run length (steps/outer) per wave: mean", ratio.mean(), "median", np.median(ratio))
