"""K1 time per frame as a function of the frames per launch (vrt_render_geometry_batch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import voxel_raytracing_amd as vrt
res = (1920, 1080)
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(metallic_ids=range(200, 256)), sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
eng.set_timing(False)
def pushes_for(n):
    return [vrt.make_push(vrt.CameraController(position=(pos0[0] + 1.5 * f, pos0[1] + 0.5 * f, pos0[2] + 2.0 * f), yaw=yaw, pitch=pitch), (256, 256, 256), res) for f in range(n)]
def run(st, n, reps=40):
    stage = vrt.GeometryStage(eng, st, sc)
    pushes = pushes_for(n)
    launch = stage.prepare_batch(n) if n > 1 else None
    single = stage.prepare() if n == 1 else None
    def go():
        if n == 1: single(pushes[0])
        else: launch(pushes)
    for _ in range(10): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / n * 1e3
prim = vrt.VoxelRenderSettings.primary_only(res)
full = vrt.VoxelRenderSettings(targetResolution=res); full.fsrSetttings.enable = False
cfg3 = vrt.VoxelRenderSettings.primary_only(res); cfg3.traceSettings.shadows = True
for name, st in (("primary", prim), ("config3", cfg3), ("defaults", full)):
    print(name, " | ".join(f"{n}/launch {run(st, n):.1f} us/frame" for n in (1, 8, 32)), flush=True)
