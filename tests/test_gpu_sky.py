"""The sky-texel fast path (csrc/vrt_sky.h; skyColor, voxel_volume.frag:98-105) on the GPU itself: waves that cannot hit
anything decide their sky texel with the hardware's 1-ulp rcp / rsq / sqrt and store the miss pixel without normalising the
ray.  (1) vrt_debug_sky_texels: for tens of millions of directions -- random, and aimed at texel edges from both sides at
distances from 1e-7 to 1e-3 of a texel -- every lane that says "sure" has the texel the numeric spec computes, for several
texture sizes; (2) frames over skies whose texels all differ (any wrong texel is a wrong pixel) are identical with the path
on and off, and equal the oracle's: all six reference targets, ragged sizes, jitter, 16x16-tile kernels, batches in kernel
arguments and in the table, sharded launches that also write the packed strips."""
import ctypes as C

import numpy as np
import pytest

from helpers import camera_push, metallic_palette

pytestmark = pytest.mark.gpu

GB = ["color8", "depth", "motion", "mask8", "position", "normal8", "hit_id"]


def _noise_sky(w, h, seed):
    rng = np.random.default_rng(seed)
    sky = np.zeros((h, w, 4), np.float32)
    sky[..., :3] = rng.random((h, w, 3), dtype=np.float32) * 1.2 - 0.1          # some texels clamp at either end
    sky[..., 3] = 1.0
    return sky


def _directions(torch, dev, w, h, n, seed):
    """n unnormalised directions: a third uniformly random (with scales from 1e-3 to 1e3), a third aimed next to a vertical
    texel edge, a third next to a horizontal one -- on either side, at distances spread over 1e-7 .. 1e-3 texels."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    m = n // 3
    rnd = torch.randn((n - 2 * m, 3), generator=g, device=dev, dtype=torch.float32)
    def aimed(k, on_u):
        u = torch.rand(k, generator=g, device=dev, dtype=torch.float64)
        v = torch.rand(k, generator=g, device=dev, dtype=torch.float64) * 0.9 + 0.05
        size = w if on_u else h
        edge = torch.randint(0, size + 1, (k,), generator=g, device=dev).double()
        off = 10.0 ** (torch.rand(k, generator=g, device=dev, dtype=torch.float64) * 4.0 - 7.0)
        off = off * (torch.randint(0, 2, (k,), generator=g, device=dev).double() * 2.0 - 1.0)
        t = (edge + off) / size
        if on_u: u = t
        else: v = t.clamp(0.04, 0.96)
        phi = (u - 0.5) / 0.1591
        el = (v - 0.5) / 0.3183
        dy = -torch.sin(el)
        c = torch.cos(el)
        d = torch.stack([c * torch.cos(phi), dy, c * torch.sin(phi)], dim=1)
        return d.float()
    d = torch.cat([rnd, aimed(m, True), aimed(m, False)], dim=0)
    scale = 10.0 ** (torch.rand((n, 1), generator=g, device=dev) * 6.0 - 3.0)
    return (d * scale).contiguous()


@pytest.mark.parametrize("w,h", [(512, 256), (2048, 1024), (4096, 2048), (37, 11), (1, 1), (8192, 16)])
def test_fast_texel_equals_spec_texel_on_the_device(vrt, engine, w, h):
    import torch
    vol = np.zeros((8, 8, 8), np.uint8); vol[4, 4, 4] = 1
    sc = vrt.VoxelScene.from_dense(engine, vol, metallic_palette(vrt), sky=_noise_sky(w, h, 1), noise=vrt.synthetic.blue_noise_standin(8))
    n_total, sure_total = 0, 0
    for chunk in range(4):
        n = 6_000_000
        d = _directions(torch, engine.torch_device, w, h, n, 100 + chunk)
        out = torch.zeros((n, 4), dtype=torch.int32, device=engine.torch_device)
        vrt._capi.check(vrt.lib().vrt_debug_sky_texels(engine.ctx, sc.handle, C.c_void_p(d.data_ptr()), n, C.c_void_p(out.data_ptr())))
        engine.synchronize()
        sure = out[:, 2] != 0
        wrong = sure & (out[:, 0] != out[:, 1])
        nw = int(wrong.sum().item())
        if nw:
            i = int(torch.nonzero(wrong)[0].item())
            raise AssertionError(f"{nw} sure lanes with another texel than the spec's; first: v = {d[i].tolist()}, spec {int(out[i, 0]):#x}, fast {int(out[i, 1]):#x}")
        n_total += n; sure_total += int(sure.sum().item())
    # random directions are sure but for the polar caps (|d.y| > 0.96: 4 % of the sphere) and the guard bands; two thirds of
    # the sample are aimed AT an edge, most of those inside a band
    assert sure_total > 0.25 * n_total, (sure_total, n_total)
    sc.destroy()


def _frames(vrt, engine, sc, st, pushes, fast, shard=None, strips=False):
    W, H = st.renderResolution()
    planes = GB + (["color8_strips"] if strips else [])
    gbs = [vrt.GeometryBuffer(engine, W, H, planes) for _ in pushes]
    stc = st.to_c()
    n = len(pushes)
    parr = (vrt._capi.Push * n)(*pushes)
    farr = (vrt._capi.Frame * n)(*[g.to_c() for g in gbs])
    with engine.options(sky_fast=fast):
        vrt._capi.check(vrt.lib().vrt_render_geometry_batch(engine.ctx, sc.handle, n, parr, C.byref(stc), farr,
                                                            C.byref(shard) if shard is not None else None))
        engine.synchronize()
    return [g.numpy() for g in gbs]


CAMERAS = [((0.5, 0.5, -0.8), 90.0, 0.0), ((0.15, 0.8, -0.25), 70.0, -25.0), ((1.2, 1.2, 1.2), 225.0, -35.0), ((0.5, 0.5, -6.0), 90.0, 0.0),
           ((2.4, 0.55, -2.0), 128.0, -3.0), ((-0.5, 0.3, 0.5), 200.0, 10.0), ((0.5, 3.0, 0.5), 90.0, -60.0), ((0.5, -2.0, 0.4), 45.0, 70.0)]


@pytest.mark.parametrize("trav", ["AUTO", "BITMASK", "DENSE"])
def test_frames_equal_with_and_without_the_fast_path_and_the_oracle(vrt, oracle, engine, trav):
    vol = vrt.synthetic.treehouse(40, seed=3)
    D, H, W = vol.shape
    pal = metallic_palette(vrt)
    for si, (sw, sh) in enumerate([(512, 256), (97, 41), (2048, 1024)]):
        sky, noise = _noise_sky(sw, sh, 7 + si), vrt.synthetic.blue_noise_standin(32)
        sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
        osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
        for ci, (p, yaw, pitch) in enumerate(CAMERAS):
            res = [(160, 96), (131, 77), (64, 40), (200, 120)][(ci + si) % 4]
            st = vrt.VoxelRenderSettings.primary_only(res, getattr(vrt, "TRAVERSAL_" + trav))
            if ci % 3 == 1:
                st.occlusionSettings.numSamples = 2; st.traceSettings.shadows = True
            if ci % 3 == 2 and trav == "AUTO":
                st.traceSettings.splitKernels = True; st.traceSettings.shadows = True
            push = camera_push(vrt, (W, H, D), res, (p[0] * W, p[1] * H, p[2] * D), yaw, pitch, frame=ci,
                               jitter=(0.3, -0.2) if ci % 2 else (0.0, 0.0))
            on = _frames(vrt, engine, sc, st, [push], 1)[0]
            off = _frames(vrt, engine, sc, st, [push], 0)[0]
            for k in GB:
                assert (on[k] == off[k]).all(), (trav, si, ci, k, int((on[k] != off[k]).sum()))
            exp = oracle.render(osn, push, oracle.params_from(st.to_c()), planes=["color8", "hit_id", "normal8", "position", "depth", "mask8"], nthreads=8)
            for k in ("color8", "hit_id", "normal8", "position", "depth", "mask8"):
                assert (on[k] == exp[k]).all(), (trav, si, ci, k, int((on[k] != exp[k]).sum()))
        sc.destroy()


def test_batches_and_sharded_strips(vrt, oracle, engine):
    """12 frames in one launch (slots in the table) and 3 (kernel arguments), unsharded and as rank 1 of 3 with the packed strips
    written by the kernel: with and without the fast path, and the full frames against the oracle's colour."""
    vol = vrt.synthetic.floating_cubes(32, seed=5, count=12)
    D, H, W = vol.shape
    pal = metallic_palette(vrt)
    sky, noise = _noise_sky(512, 256, 21), vrt.synthetic.blue_noise_standin(32)
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
    res = (176, 112)
    st = vrt.VoxelRenderSettings.primary_only(res)
    for nf in (12, 3):
        pushes = [camera_push(vrt, (W, H, D), res, (16.0 + 3.0 * f, 18.0, -60.0 - f), 90.0 + 2.0 * f, -3.0, frame=f) for f in range(nf)]
        on = _frames(vrt, engine, sc, st, pushes, 1)
        off = _frames(vrt, engine, sc, st, pushes, 0)
        for f in range(nf):
            for k in GB:
                assert (on[f][k] == off[f][k]).all(), (nf, f, k)
        exp = oracle.render(osn, pushes[nf - 1], oracle.params_from(st.to_c()), planes=["color8"], nthreads=8)
        assert (on[nf - 1]["color8"] == exp["color8"]).all()
        shard = vrt._capi.Shard(1, 3, 16)
        son = _frames(vrt, engine, sc, st, pushes, 1, shard=shard, strips=True)
        soff = _frames(vrt, engine, sc, st, pushes, 0, shard=shard, strips=True)
        for f in range(nf):
            assert (son[f]["color8_strips"] == soff[f]["color8_strips"]).all(), (nf, f)
            rows = [y for y in range(res[1]) if (y // 16) % 3 == 1]
            assert (son[f]["color8"][rows] == on[f]["color8"][rows]).all(), (nf, f)
    sc.destroy()


def test_option_names(vrt, engine):
    for name in ("tile_tags", "box_rect", "xcd_regions", "fast_loop", "no_bounce_kernel", "sky_fast", "open_cells", "df_prefetch", "df_own"):
        old = engine.option(name)
        with engine.options(**{name: 0}):
            assert engine.option(name) == 0
        assert engine.option(name) == old
    with pytest.raises(vrt.VrtError):
        engine.set_option("no_such_switch", 1)
