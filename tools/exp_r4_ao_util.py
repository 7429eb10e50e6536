"""How full are the AO rays' look-ups?  Lane look-ups (VRT_FLAG_MARCHED_COUNTS | VRT_FLAG_LOOKUP_COUNTS: one per live lane and look) of a
frame with and without AO rays; divide the difference by 64 x the wave-level loads the same two frames issue (SQ_INSTS_VMEM_RD,
tools/exp_r4_breakdown.sh) for the fraction of lanes that are live in an AO look-up."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import voxel_raytracing_amd as vrt
eng = vrt.Engine(0)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sky, noise = vrt.synthetic.sky_gradient(512, 256), vrt.synthetic.blue_noise_standin(512)


def counts(scene, st, push, W, H, flags):
    gb = vrt.GeometryBuffer(eng, W, H, ("steps_primary", "steps_total", "rays_total", "hit_id"))
    stc = st.to_c(); stc.flags |= flags
    frm = gb.to_c()
    vrt._capi.check(vrt.lib().vrt_render_geometry(eng.ctx, scene.handle, C.byref(push), C.byref(stc), C.byref(frm), None))
    eng.synchronize()
    return (int(gb.steps_primary.to(torch.int64).sum()), int(gb.steps_total.to(torch.int64).sum()), int(gb.rays_total.to(torch.int64).sum()), int((gb.hit_id != 0).sum()))


def run(tag, scene, res, dims, pos, frame):
    push = vrt.make_push(vrt.CameraController(position=pos), dims, res, frame=frame)
    for name, ao, sh in (("primary", 0, False), ("ao4", 4, False), ("shadow", 0, True)):
        st = vrt.VoxelRenderSettings(targetResolution=res)
        st.fsrSetttings.enable = False; st.denoiserSettings.enable = False
        st.occlusionSettings.numSamples = ao; st.traceSettings.shadows = sh; st.traceSettings.maxReflections = 0
        it = counts(scene, st, push, res[0], res[1], 16)
        lk = counts(scene, st, push, res[0], res[1], 16 | 32)
        print(f"UTIL {tag} {name}: iterations primary/total {it[0]}/{it[1]} rays {it[2]} hits {it[3]} | lane look-ups primary/total {lk[0]}/{lk[1]}", flush=True)


sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), pal, sky=sky, noise=noise)
pos0, yaw, pitch = vrt.synthetic.default_camera_for(256, 256, 256)
run("treehouse1080p", sc, (1920, 1080), (256, 256, 256), pos0, 0)
sc.destroy()
scm = vrt.VoxelScene.from_dense(eng, vrt.synthetic.mandelbulb(512), pal, sky=sky, noise=noise)
run("mandelbulb4k", scm, (3840, 2160), (512, 512, 512), (512 * 0.5 + 0.3, 512 * 0.5 + 0.2, -0.45 * 512), 5)
