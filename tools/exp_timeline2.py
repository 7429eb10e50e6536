"""Wave timeline of the PRODUCT K1 (a development build with -DVRT_EXP_STAMPS: every wave's start and end stamp, 10 ns units, in the
motion plane): VRT_LIB=.../libvrt_hip_stamps.so python tools/exp_timeline2.py [name=value ...] [--frames 1] [--mode primary|cfg3|defaults]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import voxel_raytracing_amd as vrt

args = sys.argv[1:]
def flag(name, default):
    if name in args:
        i = args.index(name); v = args[i + 1]; del args[i:i + 2]; return v
    return default
mode = flag("--mode", "primary")
W, H = 1920, 1080
eng = vrt.Engine(0)
for a in args:
    k, v = a.split("="); eng.set_option(k, int(v))
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(metallic_ids=range(200, 256)), sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
push = vrt.make_push(vrt.CameraController(position=pos0, yaw=yaw, pitch=pitch), (256, 256, 256), (W, H))
if mode == "primary": st = vrt.VoxelRenderSettings.primary_only((W, H))
elif mode == "cfg3":
    st = vrt.VoxelRenderSettings.primary_only((W, H)); st.traceSettings.shadows = True
else:
    st = vrt.VoxelRenderSettings(targetResolution=(W, H)); st.fsrSetttings.enable = False
stage = vrt.GeometryStage(eng, st, sc)
launch = stage.prepare()
for _ in range(5): gb = launch(push)
torch.cuda.synchronize()
o = gb.numpy()
m = o["motion"].view(np.uint32).astype(np.int64).reshape(H, W, 2)
hit = o["mask8"].reshape(H, W) != 0
t0 = m[..., 0]; t1 = m[..., 1]
hb, wb = H // 8, W // 8
b0 = t0[:hb * 8].reshape(hb, 8, wb, 8)[:, 0, :, 0]; b1 = t1[:hb * 8].reshape(hb, 8, wb, 8).max(axis=(1, 3))
traced = hit[:hb * 8].reshape(hb, 8, wb, 8).any(axis=(1, 3))
base = b0.min()
start = (b0 - base) * 0.01; end = (b1 - base) * 0.01
dur = end - start
print(f"waves {start.size}  span {end.max():.1f} us  mean wave {dur.mean():.2f} us  max {dur.max():.1f}  waves with a hit {int(traced.sum())}: mean {dur[traced].mean():.2f} us, without: {dur[~traced].mean():.2f} us")
step = 2.0
edges = np.arange(0, end.max() + step, step)
print("t(us)   active  act.hit  started  finished")
for a, b in zip(edges[:-1], edges[1:]):
    act = (start <= a) & (end > a)
    print(f"{a:5.0f}  {int(act.sum()):7d} {int((act & traced).sum()):7d} {int(((start >= a) & (start < b)).sum()):8d} {int(((end >= a) & (end < b)).sum()):8d}")
rows = slice(None, None, 8)
print("mean start by row group of 64 px:", [round(float(x), 1) for x in start.reshape(hb, wb)[rows].mean(axis=1)])
rg = lambda a, f: [round(float(f(a.reshape(hb, wb)[r:r + 8])), 1) for r in range(0, hb, 8)]
print("max wave duration by row group:", rg(dur, np.max))
print("latest end by row group:", rg(end, np.max))
print("waves longer than 12 us by row group:", [int((dur.reshape(hb, wb)[r:r + 8] > 12).sum()) for r in range(0, hb, 8)])
print("duration percentiles 50/90/99/99.9:", [round(float(np.percentile(dur, p)), 1) for p in (50, 90, 99, 99.9)], " waves > 12 us:", int((dur > 12).sum()), " > 20 us:", int((dur > 20).sum()))
late = end > 24
print("waves ending after 24 us:", int(late.sum()), " their mean start", round(float(start[late].mean()), 1), " mean duration", round(float(dur[late].mean()), 1))
