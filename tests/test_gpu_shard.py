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


# (stepWidth, iterations) beyond the default: extents of 2 and 6 rows put rows -2, -1 of rank 0's top strip in one 4-row
# block with frame rows 0 and 1 (the block's origin must come from its first EXISTING row); stepWidth 0 keeps every
# pass at a reach of 1; 2.5 takes the untiled bilinear kernel
@pytest.mark.parametrize("nranks,strip_rows,iters,step_width", [(2, 16, 2, 2.0), (3, 16, 2, 2.0), (4, 32, 3, 2.0),
                                                                (2, 16, 2, 1.0), (3, 16, 2, 1.0), (2, 16, 2, 5.0), (3, 16, 2, 5.0),
                                                                (2, 16, 3, 0.0), (3, 16, 3, 1.0), (2, 16, 2, 2.5),
                                                                (3, 48, 3, 2.0), (2, 80, 2, 2.0), (5, 32, 2, 2.0)])       # one band per rank
def test_sharded_denoise_with_halo(vrt, engine, nranks, strip_rows, iters, step_width):
    res = (96, 130)
    sc, st = _setup(vrt, engine, res, True)
    st.denoiserSettings.iterations = iters
    st.denoiserSettings.stepWidth = step_width
    r = vrt.VoxelRenderer(engine, st, sc)
    r.camera.position = np.array([24.3, 24.2, -40.0], np.float32)
    full = r.render().clone(); engine.synchronize()
    D = vrt.distributed
    W, H = res
    lib, ctx = vrt.lib(), engine.ctx
    ds = st.denoiser_to_c()
    halo = lib.vrt_denoise_halo_rows(C.byref(ds))
    assert halo == sum(int(np.ceil(i * step_width + 1)) for i in range(iters))
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


def test_batched_strip_copies_equal_single_ones(vrt, engine):
    """vrt_pack_rows_batch / vrt_unpack_rows_batch (frames x source ranks in one launch) against the per-image calls."""
    import ctypes as C
    import torch
    W, H, F, N = 200, 150, 5, 3
    dev = engine.torch_device
    L, ctx = vrt.lib(), engine.ctx
    rng = np.random.default_rng(3)
    fulls = [torch.from_numpy(rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)).to(dev) for _ in range(F)]
    prow = vrt.distributed.packed_rows(H, N, 16)
    P = C.c_void_p
    packed_by_rank = []
    for rank in range(N):
        sh = vrt._capi.Shard(rank, N, 16)
        batch = torch.full((F, prow, W, 4), 77, dtype=torch.uint8, device=dev)
        vrt._capi.check(L.vrt_pack_rows_batch(ctx, F, (P * F)(*[t.data_ptr() for t in fulls]), (P * F)(*[batch[f].data_ptr() for f in range(F)]),
                                              W, H, 4, C.byref(sh)))
        for f in range(F):
            one = torch.full((prow, W, 4), 77, dtype=torch.uint8, device=dev)
            vrt._capi.check(L.vrt_pack_rows(ctx, fulls[f].data_ptr(), one.data_ptr(), W, H, 4, C.byref(sh)))
            engine.synchronize()
            assert (one == batch[f]).all(), (rank, f)
        packed_by_rank.append(batch)
    outs = torch.zeros((F, H, W, 4), dtype=torch.uint8, device=dev)
    n = F * N
    src = (P * n)(*[packed_by_rank[s][f].data_ptr() for s in range(N) for f in range(F)])
    dst = (P * n)(*[outs[f].data_ptr() for s in range(N) for f in range(F)])
    shards = (vrt._capi.Shard * n)(*[vrt._capi.Shard(s, N, 16) for s in range(N) for f in range(F)])
    vrt._capi.check(L.vrt_unpack_rows_batch(ctx, n, src, dst, W, H, 4, shards))
    engine.synchronize()
    for f in range(F):
        assert (outs[f] == fulls[f]).all(), f                  # pack on every rank + unpack at the root = identity
    assert L.vrt_unpack_rows_batch(ctx, n, src, dst, W, H, 4, None) != 0


def test_sharded_batch_assembles_to_the_unsharded_frames(vrt, engine):
    """ShardedBatch (bench.py's N > 1 path) with the collective replaced by device copies: three simulated ranks render,
    pack and hand over their strips of a 5-frame batch; the root's assembled frames must equal the unsharded renders."""
    import torch
    vol = vrt.synthetic.treehouse(48, seed=6)
    sc = vrt.VoxelScene.from_dense(engine, vol, vrt.synthetic.default_palette(), sky=vrt.synthetic.sky_gradient(64, 32))
    res, F, N = (208, 136), 5, 3
    st = vrt.VoxelRenderSettings.primary_only(res)
    pushes = [vrt.make_push(vrt.CameraController(position=(24.0 + f, 25.0, -40.0 + 2.0 * f)), (48, 48, 48), res) for f in range(F)]
    ref = [g.color.clone() for g in vrt.distributed.ShardedBatch(vrt.GeometryStage(engine, st, sc), F, 0, 1).step(pushes)]
    ranks = [vrt.distributed.ShardedBatch(vrt.GeometryStage(engine, st, sc), F, r, N) for r in range(N)]
    bufs = ranks[0].recv_buffers()
    for r, sb in enumerate(ranks):
        sb.render(pushes)
        bufs[r].copy_(sb.pack())                                # stands in for the gather
    finals = ranks[0].assemble()
    engine.synchronize()
    for f in range(F):
        assert (finals[f] == ref[f]).all(), f
    assert any((ref[f] != ref[0]).any().item() for f in range(1, F))


@pytest.mark.parametrize("rotate,in_place", [(True, False), (False, False), (True, True)])
def test_sharded_batch_owner_blocks_assemble_to_the_unsharded_frames(vrt, engine, rotate, in_place):
    """ShardedBatch(assemble_on="owners"): frame block b ends up on rank b.  The all-to-all is replaced by device copies
    (chunk d of rank s's send buffer -> chunk s of rank d's receive buffer); 136 rows = 8.5 strips over 3 ranks, so the
    ranks own different numbers of rows and the rotation of the strip assignment per block matters."""
    import torch
    vol = vrt.synthetic.treehouse(48, seed=6)
    sc = vrt.VoxelScene.from_dense(engine, vol, vrt.synthetic.default_palette(), sky=vrt.synthetic.sky_gradient(64, 32))
    res, N, FB = (208, 136), 3, 2
    F = N * FB
    st = vrt.VoxelRenderSettings.primary_only(res)
    pushes = [vrt.make_push(vrt.CameraController(position=(24.0 + f, 25.0, -40.0 + 2.0 * f)), (48, 48, 48), res) for f in range(F)]
    ref = [g.color.clone() for g in vrt.distributed.ShardedBatch(vrt.GeometryStage(engine, st, sc), F, 0, 1).step(pushes)]
    ranks = [vrt.distributed.ShardedBatch(vrt.GeometryStage(engine, st, sc), F, r, N, assemble_on="owners", rotate=rotate, in_place=in_place)
             for r in range(N)]
    assert all(sb.in_place == in_place for sb in ranks)       # (with the rotation: one band of 48 rows per rank)
    recv = [sb.recv_buffers() for sb in ranks]
    for s, sb in enumerate(ranks):
        sb.render(pushes)
        sent = sb.pack()
        for d in range(N):
            recv[d][s * FB:(s + 1) * FB].copy_(sent[d * FB:(d + 1) * FB])
    rows = []
    for d, sb in enumerate(ranks):
        finals = sb.assemble()
        engine.synchronize()
        assert list(sb.owned_frames()) == list(range(d * FB, (d + 1) * FB))
        for j, f in enumerate(sb.owned_frames()):
            assert (finals[j] == ref[f]).all(), (d, f)
            if in_place:                                       # the frame as a consumer walks it: N bands of the receive buffer, in row order
                bands = finals.bands(j)
                assert [r0 for r0, _ in bands] == sorted(r0 for r0, _ in bands) and sum(t.shape[0] for _, t in bands) == res[1]
                for r0, t in bands:
                    assert t.is_contiguous() and (t == ref[f][r0:r0 + t.shape[0]]).all()
        # rows this rank traces per step: equal over the ranks only with the rotation
        rows.append(sum(len(vrt.distributed.owned_rows(res[1], sb.virtual_rank(d, b), N, sb.strip_rows)) for b in range(N)))
    assert sum(rows) == N * res[1]
    assert (max(rows) == min(rows)) == rotate


def test_sharded_batch_owner_blocks_need_an_even_batch(vrt, engine):
    vol = vrt.synthetic.treehouse(32, seed=1)
    sc = vrt.VoxelScene.from_dense(engine, vol, vrt.synthetic.default_palette())
    st = vrt.VoxelRenderSettings.primary_only((64, 48))
    with pytest.raises(ValueError):
        vrt.distributed.ShardedBatch(vrt.GeometryStage(engine, st, sc), 5, 0, 3, assemble_on="owners")


def test_collectives_of_sharded_batch_run_on_rccl(vrt, engine):
    """The two collectives ShardedBatch issues, on a real RCCL group (world size 1: what one GPU allows): RGBA8 strips as a
    4-D uint8 device tensor, async_op, completion by work.wait() on the launch stream."""
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        D = vrt.distributed
        sb = D.ShardedBatch.__new__(D.ShardedBatch)
        sb.rank, sb.nranks, sb.group = 0, 1, None
        send = torch.randint(0, 256, (8, 144, 96, 4), dtype=torch.uint8, device="cuda")
        for owners in (True, False):
            sb.owners = owners
            recv = torch.zeros_like(send)
            w = sb._collective(send, recv if owners else [recv], True)
            w.wait()
            torch.cuda.synchronize()
            assert (recv == send).all(), owners
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["primary_only", "megakernel", "split"])
def test_color8_strips_is_the_packed_colour(vrt, engine, mode):
    """vrt_frame.color8_strips: the tracing kernel's own copy of the colour in packed-strip order must be what
    vrt_pack_rows makes of the full colour plane -- for every rank of an uneven split (136 rows, 3 ranks) and unsharded."""
    vol = vrt.synthetic.treehouse(48, seed=6)
    sc = vrt.VoxelScene.from_dense(engine, vol, metallic_palette(vrt), sky=vrt.synthetic.sky_gradient(64, 32), noise=vrt.synthetic.blue_noise_standin(64))
    res = (208, 136)
    W, H = res
    st = vrt.VoxelRenderSettings.primary_only(res) if mode == "primary_only" else vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.traceSettings.splitKernels = mode == "split"
    stage = vrt.GeometryStage(engine, st, sc)
    L = vrt._capi.lib()
    n = 3
    pushes = [vrt.make_push(vrt.CameraController(position=(24.0 + f, 25.0, -40.0 + 2.0 * f)), (48, 48, 48), res) for f in range(n)]
    for shard in (None, vrt.make_shard(0, 3, 16), vrt.make_shard(2, 3, 16)):
        prow = vrt.distributed.packed_rows(H, 3, 16) if shard is not None else H
        launch = stage.prepare_batch(n, shard)
        strips = torch.full((n, prow, W, 4), 77, dtype=torch.uint8, device="cuda")
        tab = (vrt._capi.Frame * n)()
        C.memmove(tab, launch._keepalive[1], C.sizeof(tab))
        for f in range(n):
            tab[f].color8_strips = strips[f].data_ptr()
        gbs = launch(pushes, tab)
        engine.synchronize()
        for f in range(n):
            want = torch.full((prow, W, 4), 77, dtype=torch.uint8, device="cuda")
            if shard is None:
                want.copy_(gbs[f].color)
            else:
                assert L.vrt_pack_rows(engine.ctx, gbs[f].color.data_ptr(), want.data_ptr(), W, H, 4, C.byref(shard)) == 0
                engine.synchronize()
                rm = vrt.distributed.packed_row_map(H, shard.rank, 3, 16)
                want[torch.from_numpy(rm < 0).cuda()] = 77                       # padding rows: untouched by the kernel
            assert (strips[f] == want).all(), (mode, shard is not None and shard.rank, f)
        assert strips.ne(77).any()


def test_side_stream_unpack_assembles_the_same_frames(vrt, engine):
    """ShardedBatch(side_unpack=True): finish() hands the unpack to a second stream (with a context of its own) and the
    launch stream only waits for the arrival event.  Three steps with the all-to-all replaced by device copies and a work
    object whose wait() has nothing to wait for; the receive buffers are reused every step, so the copies play the role of
    the next collective and take the same guard (wait for the previous unpack)."""
    vol = vrt.synthetic.treehouse(48, seed=6)
    sc = vrt.VoxelScene.from_dense(engine, vol, vrt.synthetic.default_palette(), sky=vrt.synthetic.sky_gradient(64, 32))
    res, N, FB = (208, 136), 3, 2
    F = N * FB
    st = vrt.VoxelRenderSettings.primary_only(res)

    class Arrived:
        def wait(self):
            pass

    ranks = [vrt.distributed.ShardedBatch(vrt.GeometryStage(engine, st, sc), F, r, N, assemble_on="owners", side_unpack=True) for r in range(N)]
    alone = vrt.distributed.ShardedBatch(vrt.GeometryStage(engine, st, sc), F, 0, 1)
    recv = [sb.recv_buffers() for sb in ranks]
    for step in range(3):
        pushes = [vrt.make_push(vrt.CameraController(position=(24.0 + f + 3.0 * step, 25.0, -40.0 + 2.0 * f)), (48, 48, 48), res) for f in range(F)]
        ref = [g.color.clone() for g in alone.step(pushes)]
        for s, sb in enumerate(ranks):
            sb.render(pushes)
            sent = sb.pack()
            for d in range(N):
                ranks[d].wait_finals()                           # what start_gather() does before the buffers are overwritten
                recv[d][s * FB:(s + 1) * FB].copy_(sent[d * FB:(d + 1) * FB])
        for d, sb in enumerate(ranks):
            sb._work = Arrived()
            finals = sb.finish()
            assert sb._unpack_pending
        torch.cuda.synchronize()
        for d, sb in enumerate(ranks):
            for j, f in enumerate(sb.owned_frames()):
                assert (sb.finals[j] == ref[f]).all(), (step, d, f)


@pytest.mark.parametrize("mode,rotate", [("root", False), ("owners", True), ("owners", False)])
@pytest.mark.parametrize("iters,step_width", [(2, 2.0), (2, 1.0), (3, 1.0)])
def test_sharded_batch_with_denoiser(vrt, engine, mode, rotate, iters, step_width):
    """ShardedBatch(denoise=True): config 3 sharded -- the halo rows of colour, normal and position of ALL frames travel in
    one packed ring exchange per step (emulated here by handing rank r the buffers of ranks r + 1 and r - 1), every frame is
    filtered on its owner's rows, the filtered strips are assembled.  Must equal the unsharded render + denoise."""
    import torch
    vol = vrt.synthetic.floating_cubes(48, seed=9, count=70)
    sc = vrt.VoxelScene.from_dense(engine, vol, metallic_palette(vrt), sky=vrt.synthetic.sky_gradient(64, 32), noise=vrt.synthetic.blue_noise_standin(64))
    res, N, FB = (96, 136), 3, 2
    F = N * FB
    st = vrt.VoxelRenderSettings(targetResolution=res)
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = 0
    st.traceSettings.maxReflections = 0
    st.denoiserSettings.iterations = iters
    st.denoiserSettings.stepWidth = step_width
    pushes = [vrt.make_push(vrt.CameraController(position=(24.3 + f, 24.2, -40.0 + 2.0 * f)), (48, 48, 48), res, frame=f) for f in range(F)]
    ref = []
    for p in pushes:
        gb = vrt.GeometryStage(engine, st, sc).record(p)
        ref.append(vrt.DenoiserStage(engine, st).record(gb.color, gb.normal, gb.position).clone())
    D = vrt.distributed
    ranks = [D.ShardedBatch(vrt.GeometryStage(engine, st, sc), F, r, N, strip_rows=16, assemble_on=mode, rotate=rotate, denoise=True) for r in range(N)]
    assert all(sb.denoise and not sb.direct for sb in ranks)
    for sb in ranks:
        sb.render(pushes)
    sends = [tuple(t.clone() for t in sb.pack_halos()) for sb in ranks]            # (send_up, send_down) of every rank
    for r, sb in enumerate(ranks):
        sb.unpack_halos(sends[(r + 1) % N][0], sends[(r - 1) % N][1])               # from_below = what rank r + 1 sent up, ...
        sb.run_denoiser()
    if mode == "root":
        bufs = ranks[0].recv_buffers()
        for r, sb in enumerate(ranks):
            bufs[r].copy_(sb.pack())
        finals = ranks[0].assemble()
        engine.synchronize()
        for f in range(F):
            assert (finals[f] == ref[f]).all(), (f, int((finals[f] != ref[f]).sum()))
    else:
        recv = [sb.recv_buffers() for sb in ranks]
        for s, sb in enumerate(ranks):
            sent = sb.pack()
            for d in range(N):
                recv[d][s * FB:(s + 1) * FB].copy_(sent[d * FB:(d + 1) * FB])
        for d, sb in enumerate(ranks):
            finals = sb.assemble()
            engine.synchronize()
            for j, f in enumerate(sb.owned_frames()):
                assert (finals[j] == ref[f]).all(), (d, f, int((finals[j] != ref[f]).sum()))
    assert (ref[0] != ref[1]).any()


def test_batch_of_one_rank_honours_denoise_and_refuses_what_it_cannot_do(vrt, oracle, engine):
    """ShardedBatch(denoise=True) at N = 1 returns the filtered colour of every frame, as it does at N > 1 (it used to hand back
    the raw G-buffers); with the denoiser switched off in the settings, or with a diagnostic-plane stage in direct mode, it says
    so instead of dropping the request."""
    D = vrt.distributed
    vol = vrt.synthetic.floating_cubes(32, seed=3, count=20)
    pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
    sc = vrt.VoxelScene.from_dense(engine, vol, pal, sky=vrt.synthetic.sky_gradient(32, 16), noise=vrt.synthetic.blue_noise_standin(32))
    res = (96, 64)
    st = vrt.VoxelRenderSettings.primary_only(res)
    st.traceSettings.shadows = True
    st.denoiserSettings.enable = True
    pushes = [vrt.make_push(vrt.CameraController(position=(16.0 + f, 17.0, -30.0)), (32, 32, 32), res, frame=f) for f in range(3)]
    got = D.ShardedBatch(vrt.GeometryStage(engine, st, sc), 3, 0, 1, denoise=True).step(pushes)
    engine.synchronize()
    osn = oracle.OracleScene(vol, pal, sky=vrt.synthetic.sky_gradient(32, 16), noise=vrt.synthetic.blue_noise_standin(32))
    for f in range(3):
        exp = oracle.render(osn, pushes[f], oracle.params_from(st.to_c()), planes=["color8", "normal8", "position"], nthreads=4)
        assert (got[f].cpu().numpy() == oracle.denoise(exp["color8"], exp["normal8"], exp["position"])).all(), f
    st.denoiserSettings.enable = False
    with pytest.raises(ValueError):
        D.ShardedBatch(vrt.GeometryStage(engine, st, sc), 3, 0, 1, denoise=True)
    with pytest.raises(ValueError):
        D.ShardedBatch(vrt.GeometryStage(engine, st, sc, debug_planes=True), 4, 0, 2, direct=True)
    sc.destroy()
