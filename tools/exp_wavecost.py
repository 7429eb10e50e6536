"""Per-wave cost model of K1/DF: duration (wave timestamps) against iterations and services (debug planes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(), sky=vrt.synthetic.sky_gradient(512, 256))
res = (1920, 1080)
st = vrt.VoxelRenderSettings.primary_only(res, vrt.TRAVERSAL_DF)
gb = vrt.GeometryBuffer(eng, res[0], res[1], vrt.host.GBUFFER_PLANES + vrt.host.DEBUG_PLANES)
cam = vrt.CameraController(position=(128.0, 128.0, -204.8))
push = vrt.make_push(cam, (256, 256, 256), res)
fr = gb.to_c()
out = {}
for flags in (2, 1):
    stc = st.to_c(); stc.flags = flags
    for _ in range(3):
        vrt._capi.check(vrt.lib().vrt_render_geometry(eng.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
        eng.synchronize()
    o = gb.numpy()
    out[flags] = (o["steps_total"].astype(np.int64).copy(), o["rays_total"].astype(np.int64).copy(), o["steps_primary"].astype(np.int64).copy())
H, W = out[2][0].shape
def wave(a, f): return f(a[:H // 8 * 8].reshape(H // 8, 8, W // 8, 8), axis=(1, 3))
t0 = wave(out[2][0], np.min); t1 = wave(out[2][1], np.max)
dur = (t1 - t0) * 0.01
services = wave(out[1][0], np.max); steps = wave(out[1][2], np.max)
A = np.stack([np.ones(dur.size), steps.ravel(), services.ravel()], 1)
coef, *_ = np.linalg.lstsq(A, dur.ravel(), rcond=None)
print("dur_us ~ %.2f + %.4f*steps + %.4f*services" % tuple(coef), "resid std", (A @ coef - dur.ravel()).std())
for lo, hi in [(0, 10), (10, 20), (20, 30), (30, 50), (50, 100)]:
    m = (dur >= lo) & (dur < hi)
    if m.any(): print(f"dur [{lo},{hi}) us: waves {int(m.sum())}, mean steps {steps[m].mean():.0f}, mean services {services[m].mean():.1f}")
print("services: mean", services.mean(), "max", services.max(), "steps mean", steps.mean(), "max", steps.max())
np.savez_compressed("gpurun_out/wavecost.npz", dur=dur.astype(np.float32), steps=steps.astype(np.int32), services=services.astype(np.int32), start=((t0 - t0.min()) * 0.01).astype(np.float32))
