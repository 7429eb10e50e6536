"""K1 time per step of EVERY rank of an N-GPU bench step (one GPU plays them one after the other), for contiguous bands and
for 16-row strips: the step time of the job is the slowest rank's."""
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
def timed(fn, reps=40):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for N in (2, 4, 8):
    pushes = pushes_for(N * 8)
    for name, sr in (("bands", None), ("strips16", 16)):
        ts = []
        for r in range(N):
            sb = vrt.distributed.ShardedBatch(vrt.GeometryStage(eng, st, sc), N * 8, r, N, assemble_on="owners", direct="only", strip_rows=sr)
            ts.append(timed(lambda: sb.render(pushes)))
            del sb
            torch.cuda.empty_cache()
        print(f"N={N} {name:8s}: max {max(ts):.1f}  min {min(ts):.1f}  " + " ".join(f"{t:.0f}" for t in ts), flush=True)
