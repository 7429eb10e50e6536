"""Look-up statistics of config 5's primary rays (library built with -DVRT_TRACE_COUNTERS: make -C csrc variant NAME=cnt
EXTRA=-DVRT_TRACE_COUNTERS; VRT_LIB=.../libvrt_hip_cnt.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
grid, pool = vrt.synthetic.sparse_brick_scene(2048, 0.015, seed=5)
sc = vrt.VoxelScene.from_bricks(eng, grid, pool, pal, sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
pos5, yaw5, pitch5 = vrt.synthetic.default_camera_for(2048, 2048, 2048)
W, H = 3840, 2160
push = vrt.make_push(vrt.CameraController(position=(pos5[0] + 0.3, pos5[1] + 0.2, pos5[2]), yaw=yaw5, pitch=pitch5), (2048, 2048, 2048), (W, H), frame=17)
st = vrt.VoxelRenderSettings.primary_only((W, H)); st.traceSettings.maxRaySteps = 6144
gb = vrt.GeometryBuffer(eng, W, H, ("steps_primary", "steps_total", "rays_total", "hit_id"))
stc = st.to_c(); stc.flags = 1
frm = gb.to_c()
vrt._capi.check(vrt.lib().vrt_render_geometry(eng.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(frm), None))
eng.synchronize()
steps = gb.steps_primary.to(torch.int64); look = gb.steps_total.to(torch.int64); occ = gb.rays_total.to(torch.int64)
hit = (gb.hit_id != 0)
# per wave (8x8 block): look-ups = max over lanes (every live lane looks at every look-up of its wave), iterations = max over lanes
def blocks(t): return t.reshape(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)
bl, bs, bo = blocks(look), blocks(steps), blocks(occ)
wl, ws = bl.max(1).values, bs.max(1).values
traced = wl > 0
print(f"pixels hit {hit.float().mean().item():.3f}; waves traced {int(traced.sum())} of {traced.numel()}")
print(f"per traced wave: look-ups {wl[traced].float().mean().item():.1f}, iterations {ws[traced].float().mean().item():.1f}, iterations per look-up {(ws[traced].sum() / wl[traced].sum()).item():.2f}")
print(f"lane-level: look-ups in occupied bricks {occ.sum().item() / max(1, look.sum().item()):.3f} of all; lane utilisation (sum lane steps / (wave steps x 64)) {(bs.sum() / (ws.sum() * 64)).item():.3f}")
print(f"sum lane steps {steps.sum().item() / 1e6:.1f} M, wave iterations {ws.sum().item() / 1e6:.2f} M, wave look-ups {wl.sum().item() / 1e6:.2f} M")
