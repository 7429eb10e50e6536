"""Screen-tile sharding on ONE GPU: every simulated rank renders its strips; assembling them (pack ->
unpack, as the RCCL gather would deliver them) must reproduce the unsharded frame bit-for-bit, with
and without the sharded denoiser + ring halo exchange (emulated by local copies)."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import camera_push, metallic_palette

pytestmark = pytest.mark.gpu


def _setup(vrt, engine, res, denoise):
    vol = vrt.synthetic.floating_cubes(48, seed=9, count=70)
    sc = vrt.VoxelScene.from_dense(engine, vol, metallic_palette(vrt), sky=vrt.synthetic.sky_gradient(64, 32),
                                   noise=vrt.synthetic.blue_noise_standin(64))
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = 1
    st.denoiserSettings.enable = denoise
    return sc, st


@pytest.mark.parametrize("nranks,strip_rows", [(2, 16), (3, 16), (8, 16), (4, 32)])
def test_sharded_trace_assembles_to_full_frame(vrt, engine, nranks, strip_rows):
    res = (100, 150)
    sc, st = _setup(vrt, engine, res, False)
    r = vrt.VoxelRenderer(engine, st, sc)
    r.camera.position = np.array([24.3, 24.2, -40.0], np.float32)
    full = r.render().clone(); engine.synchronize()
    D = vrt.distributed
    W, H = res
    final = torch.zeros_like(full)
    lib, ctx = vrt.lib(), engine.ctx
    for rank in range(nranks):
        sf = D.ShardedFrame(r, rank, nranks, strip_rows)
        color = sf.render_local()
        packed = sf.pack(color).clone()
        # host mirror of the packing
        rm = D.packed_row_map(H, rank, nranks, strip_rows)
        assert (packed.cpu().numpy() == D.pack_np(color.cpu().numpy(), rm)).all()
        sh = vrt._capi.Shard(rank, nranks, strip_rows)
        vrt._capi.check(lib.vrt_unpack_rows(ctx, packed.data_ptr(), final.data_ptr(), W, H, 4, C.byref(sh)))
    engine.synchronize()
    assert (final == full).all()


@pytest.mark.parametrize("nranks,strip_rows,iters", [(2, 16, 2), (3, 16, 2), (4, 32, 3)])
def test_sharded_denoise_with_halo(vrt, engine, nranks, strip_rows, iters):
    res = (96, 130)
    sc, st = _setup(vrt, engine, res, True)
    st.denoiserSettings.iterations = iters
    r = vrt.VoxelRenderer(engine, st, sc)
    r.camera.position = np.array([24.3, 24.2, -40.0], np.float32)
    full = r.render().clone(); engine.synchronize()
    D = vrt.distributed
    W, H = res
    lib, ctx = vrt.lib(), engine.ctx
    ds = st.denoiser_to_c()
    halo = lib.vrt_denoise_halo_rows(C.byref(ds))
    assert halo == sum(2 * i + 1 for i in range(iters))
    # every rank traces its strips into its own G-buffer
    gbs = []
    for rank in range(nranks):
        stage = vrt.GeometryStage(engine, st, sc)
        gbs.append(stage.record(r.push_constants(), vrt._capi.Shard(rank, nranks, strip_rows)))
    # ring halo exchange emulated with device copies: rank r receives rank r+1's first rows and rank r-1's last rows
    for name, bpp in (("color8", 4), ("normal8", 4), ("position", 16)):
        for rank in range(nranks):
            for src, direction in (((rank + 1) % nranks, -1), ((rank - 1) % nranks, 1)):
                sh = vrt._capi.Shard(src, nranks, strip_rows)
                nbytes = lib.vrt_halo_bytes(W, H, bpp, C.byref(sh), halo)
                buf = torch.zeros(nbytes, dtype=torch.uint8, device=full.device)
                vrt._capi.check(lib.vrt_pack_halo(ctx, gbs[src].planes[name].data_ptr(), buf.data_ptr(), W, H, bpp, C.byref(sh), halo, direction))
                rm = D.halo_row_map(H, src, nranks, strip_rows, halo, direction)
                assert (buf.cpu().numpy().reshape(len(rm), -1) ==
                        D.pack_np(gbs[src].planes[name].cpu().numpy().reshape(H, -1).view(np.uint8), rm)).all()
                vrt._capi.check(lib.vrt_unpack_halo(ctx, buf.data_ptr(), gbs[rank].planes[name].data_ptr(), W, H, bpp, C.byref(sh), halo, direction))
    final = torch.zeros_like(full)
    for rank in range(nranks):
        den = vrt.DenoiserStage(engine, st)
        sh = vrt._capi.Shard(rank, nranks, strip_rows)
        out = den.record(gbs[rank].color, gbs[rank].normal, gbs[rank].position, sh)
        packed = torch.zeros((D.packed_rows(H, nranks, strip_rows), W, 4), dtype=torch.uint8, device=full.device)
        vrt._capi.check(lib.vrt_pack_rows(ctx, out.data_ptr(), packed.data_ptr(), W, H, 4, C.byref(sh)))
        vrt._capi.check(lib.vrt_unpack_rows(ctx, packed.data_ptr(), final.data_ptr(), W, H, 4, C.byref(sh)))
    engine.synchronize()
    assert (final == full).all(), int((final != full).sum())
