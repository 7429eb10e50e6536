"""One frame per launch at 1080p on the bench scene: config 3 (shadow ray) and the reference defaults (AO 4, shadow, <= 5 bounces),
geometry ms; python tools/exp_secondary.py [name=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
for a in sys.argv[1:]:
    k, v = a.split("="); eng.set_option(k, int(v))
res = (1920, 1080)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(metallic_ids=range(200, 256)), sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
push = vrt.make_push(vrt.CameraController(position=pos0, yaw=yaw, pitch=pitch), (256, 256, 256), res)
full = vrt.VoxelRenderSettings(targetResolution=res); full.fsrSetttings.enable = False
cfg3 = vrt.VoxelRenderSettings.primary_only(res); cfg3.traceSettings.shadows = True
eng.set_timing(True)
for name, st in (("config3", cfg3), ("defaults", full)):
    geo = vrt.GeometryStage(eng, st, sc)
    t = []
    for _ in range(12):
        geo.record(push); eng.synchronize()
        t.append(eng.last_timings()["geometry_ms"])
    t = sorted(t[2:])
    print(f"{' '.join(sys.argv[1:]) or 'defaults'}: {name} {t[len(t) // 2] * 1e3:.1f} us (min {t[0] * 1e3:.1f})", flush=True)
