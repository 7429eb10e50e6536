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
stc = st.to_c(); stc.flags = 2
fr = gb.to_c()
for _ in range(3):
    vrt._capi.check(vrt.lib().vrt_render_geometry(eng.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
    eng.synchronize()
print("kernel us", eng.last_timings()["primary_ms"] * 1e3)
o = gb.numpy()
t0 = o["steps_total"].astype(np.int64); t1 = o["rays_total"].astype(np.int64)
H, W = t0.shape
# one value per wave (8x8 block)
b0 = t0[:H // 8 * 8].reshape(H // 8, 8, W // 8, 8)[:, 0, :, 0]; b1 = t1[:H // 8 * 8].reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
base = b0.min()
start = (b0 - base) * 0.01; end = (b1 - base) * 0.01       # us
dur = end - start
print("waves", start.size, "kernel span us", end.max(), "mean wave dur", dur.mean(), "max dur", dur.max())
edges = np.arange(0, end.max() + 10, 10)
active = [(int(((start <= t) & (end > t)).sum())) for t in edges]
print("active waves every 10us:", active)
print("waves started every 10us:", np.histogram(start, bins=edges)[0].tolist())
late = dur > np.percentile(dur, 99)
print("99th pct dur", np.percentile(dur, 99), "start of slowest 1%: mean", start[late].mean(), "end", end[late].mean())
rows = dur.mean(axis=1)
print("mean wave duration by 8-px row group (every 8th):", [round(float(x), 1) for x in rows[::8]])
print("mean start time by row group (every 8th):", [round(float(x), 1) for x in start.mean(axis=1)[::8]])
if os.path.isdir("gpurun_out"):
    np.savez_compressed("gpurun_out/timeline.npz", start=start.astype(np.float32), end=end.astype(np.float32),
                        steps=o["steps_primary"].astype(np.int32)[:H // 8 * 8].reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3)))
