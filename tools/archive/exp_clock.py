"""Ad-hoc: does K1 get faster when the GPU is kept busy (DVFS)?  Launch N frames back-to-back, time each with events."""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(), sky=vrt.synthetic.sky_gradient(512, 256))
res = (1920, 1080)
for trav in sys.argv[1:] or ["DENSE"]:
    st = vrt.VoxelRenderSettings.primary_only(res, getattr(vrt, "TRAVERSAL_" + trav))
    stage = vrt.GeometryStage(eng, st, sc)
    cam = vrt.CameraController(position=(128.0, 128.0, -204.8))
    push = vrt.make_push(cam, (256, 256, 256), res)
    N = 3000
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    evs[0].record()
    t0 = time.perf_counter()
    for i in range(N):
        stage.record(push); evs[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ts = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(N)]) * 1e3
    print(trav, "wall ms/frame", wall / N * 1e3, "event us: first10", ts[:10].mean(), "mid", ts[N//2-50:N//2+50].mean(), "last100", ts[-100:].mean(), "min", ts.min(), flush=True)
print(subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout[-400:])
