"""Exact per-frame work counts of K1/DF on the bench frame (needs the counters variant of the library:
make -C voxel-raytracing_amd/csrc variant NAME=cnt EXTRA=-DVRT_TRACE_COUNTERS; VRT_LIB=.../libvrt_hip_cnt.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, ctypes as C
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
vol = vrt.synthetic.treehouse(256, seed=2)
sc = vrt.VoxelScene.from_dense(eng, vol, vrt.synthetic.default_palette(), sky=vrt.synthetic.sky_gradient(512, 256))
res = (1920, 1080)
st = vrt.VoxelRenderSettings.primary_only(res, vrt.TRAVERSAL_DF)
gb = vrt.GeometryBuffer(eng, res[0], res[1], vrt.host.GBUFFER_PLANES + vrt.host.DEBUG_PLANES)
pos0 = (128.0, 128.0, -204.8)
for t in (0.0, 4.0, 7.75):
    cam = vrt.CameraController(position=(pos0[0] + 1.5 * t, pos0[1] + 0.5 * t, pos0[2] + 2.0 * t))
    push = vrt.make_push(cam, (256, 256, 256), res)
    stc = st.to_c(); stc.flags = 1
    fr = gb.to_c()
    vrt._capi.check(vrt.lib().vrt_render_geometry(eng.ctx, sc.handle, C.byref(push), C.byref(stc), C.byref(fr), None))
    eng.synchronize()
    o = gb.numpy()
    outer, steps, hit = o["steps_total"].astype(np.int64), o["steps_primary"].astype(np.int64), o["hit_id"] != 0
    H, W = outer.shape
    def blk(a, f):
        return f(a.reshape(H // 8, 8, W // 8, 8), axis=(1, 3))
    wo, ws, wh = blk(outer, np.max), blk(steps, np.max), blk(hit, np.sum)
    geo = ws > 0
    print(f"pose t={t}: waves {wo.size}, sky-only {int((~geo).sum())}, geometry {int(geo.sum())} "
          f"(all 64 lanes hit: {int((wh == 64).sum())}, no lane hits: {int((geo & (wh == 0)).sum())}, mixed: {int((geo & (wh > 0) & (wh < 64)).sum())})")
    print(f"   look-ups (loop trips) total {int(wo[geo].sum())} = {wo[geo].mean():.1f} per geometry wave; "
          f"wave-iterations total {int(ws.sum())} = {ws[geo].mean():.1f} per geometry wave; lane-steps {int(steps.sum())}")
    nh = geo & (wh == 0)
    print(f"   of which in waves without a hit: look-ups {int(wo[nh].sum())}, wave-iterations {int(ws[nh].sum())}; all-hit waves: {int(wo[wh == 64].sum())}, {int(ws[wh == 64].sum())}")
