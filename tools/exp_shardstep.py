"""What one rank of an N-GPU bench step costs on its GPU, without the collective: K1 over its strips of the N x 8 frames
(ShardedBatch.render), the strip pack and the unpack of one step's worth of received strips -- against the N = 1 step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import voxel_raytracing_amd as vrt
res = (1920, 1080)
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(metallic_ids=range(200, 256)), sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
eng.set_timing(False)
st = vrt.VoxelRenderSettings.primary_only(res)
def pushes_for(n):                      # the same 8-pose dolly move, sampled n / 8 times as finely
    return [vrt.make_push(vrt.CameraController(position=(pos0[0] + 1.5 * t, pos0[1] + 0.5 * t, pos0[2] + 2.0 * t), yaw=yaw, pitch=pitch), (256, 256, 256), res)
            for t in (f * 8.0 / n for f in range(n))]
def timed(fn, reps=60):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
modes = [a for a in sys.argv[1:] if a in ("owners", "root")] or ["owners"]
direct = False if "copy" in sys.argv[1:] else ("only" if "only" in sys.argv[1:] else True)
for N in (1, 2, 4, 8):
    for mode in modes:
        F = N * 8
        sb = vrt.distributed.ShardedBatch(vrt.GeometryStage(eng, st, sc), F, min(1, N - 1), N, assemble_on=mode, direct=direct, side_unpack=("side" in sys.argv[1:]), rotate=(False if "norotate" in sys.argv[1:] else None), strip_rows=(16 if "strips16" in sys.argv[1:] else None), in_place=("inplace" in sys.argv[1:]))
        pushes = pushes_for(F)
        k1 = timed(lambda: sb.render(pushes))
        if N == 1:
            print(f"N=1: K1 {k1:.1f} us/step", flush=True)
            continue
        pk = timed(sb.pack)
        line = f"N={N} {mode}: K1 {k1:.1f} us/step  pack {pk:.1f}"
        if sb._receives():
            sb.recv_buffers()
            up = timed(sb.assemble)
            line += f"  unpack {up:.1f}  sum {k1 + pk + up:.1f}"
            if sb.side_unpack:
                class Arrived:
                    def wait(self): pass
                def one():
                    sb.render(pushes); sb._work = Arrived(); sb.finish(); sb.wait_finals(); sb._unpack_pending = False
                # (wait_finals here stands for the guard of the next start_gather: it comes after the next render in the real
                #  order, so this is the pessimistic placement)
                def two():
                    sb.wait_finals(); sb._unpack_pending = False; sb.render(pushes); sb._work = Arrived(); sb.finish()
                al = timed(two)
                line += f"  with side-stream unpack {al:.1f}"
            else:
                al = timed(lambda: (sb.render(pushes), sb.assemble(), sb.pack()))
                line += f"  in sequence {al:.1f}"
        print(line, flush=True)
        if True:
            import time
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): sb.render(pushes)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            print(f"      host time to enqueue one render(): {(t1 - t0) / 3 * 1e6:.0f} us", flush=True)
        del sb
        torch.cuda.empty_cache()
