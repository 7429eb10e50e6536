"""K1 time per frame for launches of 8 (slots in the kernel arguments) and 16..64 frames (slot table in device memory)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import voxel_raytracing_amd as vrt
res = (1920, 1080)
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(metallic_ids=range(200, 256)), sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
eng.set_timing(False)
st = vrt.VoxelRenderSettings.primary_only(res)
def pushes_for(n):
    return [vrt.make_push(vrt.CameraController(position=(pos0[0] + 1.5 * t, pos0[1] + 0.5 * t, pos0[2] + 2.0 * t), yaw=yaw, pitch=pitch), (256, 256, 256), res)
            for t in (f * 8.0 / n for f in range(n))]
for n in (8, 16, 32, 64, 8):
    stage = vrt.GeometryStage(eng, st, sc)
    launch = stage.prepare_batch(n)
    pushes = pushes_for(n)
    reps = 512 // n
    for _ in range(3): launch(pushes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch(pushes)
    e1.record(); torch.cuda.synchronize()
    print(f"{n} frames/launch: {e0.elapsed_time(e1) / reps / n * 1e3:.2f} us/frame", flush=True)
    del launch, stage
    torch.cuda.empty_cache()
