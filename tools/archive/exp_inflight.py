"""Throughput of K1 with 1, 2 and 3 frames in flight (one context + stream + G-buffer per frame slot, one shared scene)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import voxel_raytracing_amd as vrt
res = (1920, 1080)
engines = [vrt.Engine(0, use_torch_stream=False) for _ in range(3)]
eng = engines[0]
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(metallic_ids=range(200, 256)), sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
eng.synchronize()
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
push = vrt.make_push(vrt.CameraController(position=pos0, yaw=yaw, pitch=pitch), (256, 256, 256), res)
gbs = [vrt.GeometryBuffer(e, res[0], res[1]) for e in engines]
frs = [g.to_c() for g in gbs]
for e in engines: e.set_timing(False)
def run(settings, nslots, n=300):
    stc = settings.to_c()
    for k in range(20):
        e = engines[k % nslots]
        vrt._capi.check(vrt.lib().vrt_render_geometry(e.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(frs[k % nslots]), None))
    for e in engines: e.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        e = engines[k % nslots]
        vrt._capi.check(vrt.lib().vrt_render_geometry(e.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(frs[k % nslots]), None))
    for e in engines: e.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
prim = vrt.VoxelRenderSettings.primary_only(res)
full = vrt.VoxelRenderSettings(targetResolution=res); full.fsrSetttings.enable = False
cfg3 = vrt.VoxelRenderSettings.primary_only(res); cfg3.traceSettings.shadows = True
for name, st in (("primary", prim), ("config3", cfg3), ("defaults", full)):
    print(name, " | ".join(f"{k} in flight {run(st, k):.1f} us/frame" for k in (1, 2, 3)), flush=True)
same = all((gbs[0].color == g.color).all().item() and (gbs[0].position == g.position).all().item() for g in gbs[1:])
print("all slots hold the same frame:", same)
