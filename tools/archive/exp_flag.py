"""K1 / geometry time with and without one development flag of the library: python tools/exp_flag.py FLAG"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import voxel_raytracing_amd as vrt
FLAG = int(sys.argv[1])
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(metallic_ids=range(200, 256)), sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
res = (1920, 1080)
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
gb = vrt.GeometryBuffer(eng, res[0], res[1])
fr = gb.to_c()
def run(flags, settings, push, n=100):
    stc = settings.to_c(); stc.flags |= flags
    eng.set_timing(False)
    for _ in range(10):
        vrt._capi.check(vrt.lib().vrt_render_geometry(eng.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        vrt._capi.check(vrt.lib().vrt_render_geometry(eng.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, gb.color.clone(), gb.position.clone()
prim = vrt.VoxelRenderSettings.primary_only(res)
full = vrt.VoxelRenderSettings(targetResolution=res); full.fsrSetttings.enable = False
cfg3 = vrt.VoxelRenderSettings.primary_only(res); cfg3.traceSettings.shadows = True
cams = {"default": vrt.CameraController(position=pos0, yaw=yaw, pitch=pitch),
        "low, looking up": vrt.CameraController(position=(128.0, 20.0, -150.0), yaw=90.0, pitch=-25.0),
        "inside": vrt.CameraController(position=(100.0, 90.0, 60.0), yaw=60.0, pitch=10.0)}
for cname, cam in cams.items():
    push = vrt.make_push(cam, (256, 256, 256), res)
    for name, st in (("primary", prim), ("config3", cfg3), ("defaults", full)):
        a, ca, pa = run(0, st, push); b, cb, pb = run(FLAG, st, push)
        print(f"{cname:16s} {name:9s} off {a:7.1f} us | on {b:7.1f} us | same image: {bool((ca == cb).all() and (pa == pb).all())}", flush=True)
