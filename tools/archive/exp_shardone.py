"""One configuration for counter collection: python tools/exp_shardone.py N  -- 64 frames per launch, rank 1 of N (N = 1: unsharded)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import voxel_raytracing_amd as vrt
N = int(sys.argv[1]); F = 64
res = (1920, 1080)
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(metallic_ids=range(200, 256)), sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
eng.set_timing(False)
st = vrt.VoxelRenderSettings.primary_only(res)
pushes = [vrt.make_push(vrt.CameraController(position=(pos0[0] + 1.5 * t, pos0[1] + 0.5 * t, pos0[2] + 2.0 * t), yaw=yaw, pitch=pitch), (256, 256, 256), res)
          for t in (f * 8.0 / F for f in range(F))]
sb = vrt.distributed.ShardedBatch(vrt.GeometryStage(eng, st, sc), F, min(1, N - 1), N, assemble_on="owners", direct="only")
for _ in range(3): sb.render(pushes)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): sb.render(pushes)
e1.record(); torch.cuda.synchronize()
print(f"N={N}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch of {F} frame slots", flush=True)
