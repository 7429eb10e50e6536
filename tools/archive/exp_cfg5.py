"""BASELINE configs[4] (2048^3 sparse-brick scene, 3840x2160, max_steps 6144, 4 bounces, AO 4) on one GPU: geometry ms per frame.
python tools/exp_cfg5.py [name=value ...] [--reps 7]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import voxel_raytracing_amd as vrt
args = sys.argv[1:]
reps = 7
if "--reps" in args:
    i = args.index("--reps"); reps = int(args[i + 1]); del args[i:i + 2]
def flag(name, default):
    if name in args:
        i = args.index(name); v = args[i + 1]; del args[i:i + 2]; return v
    return default
ao, sh, bo = int(flag("--ao", "4")), int(flag("--shadows", "1")), int(flag("--bounces", "4"))
split = int(flag("--split", "0"))
xflags = int(flag("--flags", "0"))
eng = vrt.Engine(0)
for a in args:
    k, v = a.split("="); eng.set_option(k, int(v))
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sky, noise = vrt.synthetic.sky_gradient(512, 256), vrt.synthetic.blue_noise_standin(512)
grid, pool = vrt.synthetic.sparse_brick_scene(2048, 0.015, seed=5)
sc = vrt.VoxelScene.from_bricks(eng, grid, pool, pal, sky=sky, noise=noise)
pos5, yaw5, pitch5 = vrt.synthetic.default_camera_for(2048, 2048, 2048)
cam5 = vrt.CameraController(position=(pos5[0] + 0.3, pos5[1] + 0.2, pos5[2]), yaw=yaw5, pitch=pitch5)
push = vrt.make_push(cam5, (2048, 2048, 2048), (3840, 2160), frame=17)
st = vrt.VoxelRenderSettings(targetResolution=(3840, 2160))
st.fsrSetttings.enable = False
st.occlusionSettings.numSamples = ao; st.traceSettings.shadows = bool(sh); st.traceSettings.maxReflections = bo; st.traceSettings.maxRaySteps = 6144
st.denoiserSettings.enable = False
st.traceSettings.splitKernels = bool(split)
if xflags:
    _to_c = st.to_c
    def to_c():
        c = _to_c(); c.flags |= xflags; return c
    st.to_c = to_c
geo = vrt.GeometryStage(eng, st, sc)
eng.set_timing(True)
t = []
for _ in range(reps + 2):
    geo.record(push); eng.synchronize()
    t.append(eng.last_timings()["geometry_ms"])
t = sorted(t[2:])
print(f"{' '.join(args) or 'defaults'} ao={ao} shadows={sh} bounces={bo} split={split}: config 5 geometry {t[len(t) // 2]:.3f} ms (min {t[0]:.3f}), scene {sc.memory_bytes() / 1e6:.0f} MB", flush=True)

import ctypes as C
cnt = (C.c_uint64 * 4)()
vrt._capi.check(vrt.lib().vrt_debug_brick_counts(eng.ctx, cnt))
geo.record(push); eng.synchronize()
vrt._capi.check(vrt.lib().vrt_debug_brick_counts(eng.ctx, cnt))
if cnt[0]:
    print(f"   one frame: lane look-ups {cnt[0] / 1e6:.1f} M, in occupied bricks {cnt[1] / 1e6:.1f} M ({cnt[1] / cnt[0]:.3f}), solid found {cnt[2] / 1e6:.2f} M, border / open {cnt[3] / 1e6:.2f} M", flush=True)
