"""K1 on the bench line's workload (1080p treehouse stand-in, primary rays only) with context options set from the command
line: python tools/exp_k1.py [name=0|1 ...] [--frames 64] [--reps 40] [--res 1920x1080] [--ab name]  (--ab: alternate the option
on / off in one process, four rounds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import voxel_raytracing_amd as vrt

args = sys.argv[1:]
def flag(name, default):
    if name in args:
        i = args.index(name); v = args[i + 1]; del args[i:i + 2]; return v
    return default
frames, reps = int(flag("--frames", "64")), int(flag("--reps", "40"))
W, H = (int(x) for x in flag("--res", "1920x1080").split("x"))
ab = flag("--ab", None)
xflags = int(flag("--flags", "0"))         # extra vrt_settings.flags (8: 16x16-pixel workgroups of four waves)
back = float(flag("--back", "0"))          # camera moved back along -z by this many voxels (far away: nearly every wave is a sky wave)
turn = float(flag("--turn", "0"))          # degrees added to the yaw: 180 = every pixel is sky
eng = vrt.Engine(0)
for a in args:
    k, v = a.split("="); eng.set_option(k, int(v))
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(metallic_ids=range(200, 256)), sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
eng.set_timing(False)
pushes = [vrt.make_push(vrt.CameraController(position=(pos0[0] + 1.5 * t, pos0[1] + 0.5 * t, pos0[2] + 2.0 * t - back), yaw=yaw + turn, pitch=pitch), (256, 256, 256), (W, H))
          for t in (8.0 * f / frames for f in range(frames))]
st = vrt.VoxelRenderSettings.primary_only((W, H))
if xflags:
    _to_c = st.to_c
    def to_c():
        c = _to_c(); c.flags |= xflags; return c
    st.to_c = to_c
stage = vrt.GeometryStage(eng, st, sc)
launch = stage.prepare_batch(frames) if frames > 1 else stage.prepare()
def go():
    launch(pushes) if frames > 1 else launch(pushes[0])
def timed():
    for _ in range(5): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / frames * 1e3
if ab:
    for r in range(4):
        for v in (1, 0):
            eng.set_option(ab, v)
            print(f"{ab}={v}: {timed():.2f} us/frame", flush=True)
else:
    print(f"{' '.join(args) or 'defaults'}: {timed():.2f} us/frame ({frames} frames per launch, {W}x{H})", flush=True)
