"""One rank of the N = 8 bench step at its real sizes (256 frame slots of 1080p, 32 per block) on one GPU: allocation, the
single 256-slot launch, the unpack of 256 (source, frame) pairs, and the rank's own block against unsharded renders (the
strips of the other sources are taken from this rank's own send buffer, which is what they would send for vrank = source
rotation... only the self-sent block is checked)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import voxel_raytracing_amd as vrt
res = (1920, 1080)
N, FB = 8, 32
F = N * FB
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(metallic_ids=range(200, 256)), sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
eng.set_timing(False)
st = vrt.VoxelRenderSettings.primary_only(res)
pushes = [vrt.make_push(vrt.CameraController(position=(pos0[0] + 1.5 * t, pos0[1] + 0.5 * t, pos0[2] + 2.0 * t), yaw=yaw, pitch=pitch), (256, 256, 256), res)
          for t in (8.0 * f / F for f in range(F))]
rank = 3
t0 = time.time()
sb = vrt.distributed.ShardedBatch(vrt.GeometryStage(eng, st, sc), F, rank, N, assemble_on="owners", direct="only")
torch.cuda.synchronize()
print(f"allocated in {time.time() - t0:.1f} s, {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
for _ in range(3): sb.render(pushes)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): sb.render(pushes)
e1.record(); torch.cuda.synchronize()
print(f"K1 over {F} slots: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us per step ({e0.elapsed_time(e1) / 10 / FB * 1e3:.1f} us per frame-equivalent)", flush=True)
recv = sb.recv_buffers()
sent = sb.pack()
recv[rank * FB:(rank + 1) * FB].copy_(sent[rank * FB:(rank + 1) * FB])      # what this rank sends to itself
e0.record(); sb.assemble(); e1.record(); torch.cuda.synchronize()
print(f"unpack of {F} pairs: {e0.elapsed_time(e1) * 1e3:.0f} us", flush=True)
# the self-sent strips of the rank's own block are the rows of virtual rank (rank + rank) % N
chk = vrt.GeometryStage(eng, st, sc)
vr = sb.virtual_rank(rank, rank)
rows = vrt.distributed.owned_rows(res[1], vr, N, sb.strip_rows)
ok = True
for j in (0, FB - 1):
    alone = chk.record(pushes[rank * FB + j]).color
    torch.cuda.synchronize()
    ok = ok and bool((sb.finals[j][rows] == alone[rows]).all().item())
print("own rows of the own block match unsharded renders:", ok, flush=True)

# the same step with the unpack in line and on the side stream (fake "arrived" work objects: no collective here)
class Arrived:
    def wait(self): pass
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
def inline_step():
    sb.render(pushes); sb.assemble()
print(f"step with the unpack in line: {timed(inline_step):.0f} us", flush=True)
sb.side_unpack = True
def side_step():
    sb.wait_finals(); sb._unpack_pending = False     # the guard of start_gather
    sb.render(pushes); sb._work = Arrived(); sb.finish()
print(f"step with the unpack on the side stream: {timed(side_step):.0f} us", flush=True)
