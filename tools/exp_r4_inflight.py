"""One frame per vrt_render_geometry call with k = 1, 2, 3 contexts taking the calls in turn (the reference's frames in flight,
engine.hpp:19): wall-clock us per frame, with the tile tags on the context's stream (tags_async=0) and on a stream of their own.
Under `rocprofv3 --kernel-trace` tools/exp_r4_inflight_trace.py turns the dispatch stamps into how the kernels of the three
queues overlap.   python3 tools/exp_r4_inflight.py [n=240] [configs=pcd]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
import voxel_raytracing_amd as vrt
n, configs = 240, "pcd"
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "n": n = int(v)
    if k == "configs": configs = v
res = (1920, 1080)
engines = [vrt.Engine(0, use_torch_stream=False) for _ in range(3)]
eng = engines[0]
sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), vrt.synthetic.default_palette(metallic_ids=range(200, 256)),
                               sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
eng.synchronize()
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
pushes = [vrt.make_push(vrt.CameraController(position=(pos0[0] + 1.5 * t, pos0[1] + 0.5 * t, pos0[2] + 2.0 * t), yaw=yaw, pitch=pitch), (256, 256, 256), res)
          for t in (8.0 * f / 16 for f in range(16))]
gbs = [vrt.GeometryBuffer(e, res[0], res[1]) for e in engines]
frs = [g.to_c() for g in gbs]
for e in engines: e.set_timing(False)


def run(stc, k, m):
    for j in range(m):
        e = engines[j % k]
        vrt._capi.check(vrt.lib().vrt_render_geometry(e.ctx, sc.handle, C.byref(pushes[j % 16]), C.byref(stc), C.byref(frs[j % k]), None))
    for e in engines: e.synchronize()


prim = vrt.VoxelRenderSettings.primary_only(res)
full = vrt.VoxelRenderSettings(targetResolution=res); full.fsrSetttings.enable = False
cfg3 = vrt.VoxelRenderSettings.primary_only(res); cfg3.traceSettings.shadows = True
for name, st in (("primary", prim), ("config3", cfg3), ("defaults", full)):
    if name[0] not in configs: continue
    stc = st.to_c()
    for ta in (0, 1):
        for e in engines: e.set_option("tags_async", ta)
        row = []
        for k in (1, 2, 3):
            run(stc, k, 12)
            t0 = time.perf_counter()
            run(stc, k, n)
            row.append((time.perf_counter() - t0) / n * 1e6)
        print(f"INFLIGHT {name} tags_async={ta}: " + " | ".join(f"{k + 1} in flight {v:.1f} us" for k, v in enumerate(row)), flush=True)
