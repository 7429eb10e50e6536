"""The N>1 path on CPU: world_size-2/3 gloo process groups run the host side of the sharded frame --
strip ownership, packing, the ONE gather per step to rank 0, de-interleave, and the ring halo exchange --
with the oracle standing in for the kernels (tests may use the oracle; the product path never does)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, strip_rows, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import voxel_raytracing_amd as vrt
    from oracle import oracle
    from helpers import camera_push, metallic_palette
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = vrt.distributed
    W, H = 48, 70
    vol = vrt.synthetic.floating_cubes(32, seed=3, count=40)
    pal = metallic_palette(vrt)
    osn = oracle.OracleScene(vol, pal)
    st = vrt.VoxelRenderSettings.primary_only((W, H))
    push = camera_push(vrt, (32, 32, 32), (W, H))
    params = oracle.params_from(st.to_c())
    # "render" only the rows this rank owns
    full = np.zeros((H, W, 4), np.uint8)
    for y in D.owned_rows(H, rank, world, strip_rows):
        full[y] = oracle.render(osn, push, params, planes=["color8"], rows=(int(y), int(y) + 1))["color8"][y]
    packed = torch.from_numpy(D.pack_np(full, D.packed_row_map(H, rank, world, strip_rows)))
    bufs = D.gather_packed(packed, dst=0)
    ok = True
    if rank == 0:
        final = np.zeros_like(full)
        for src, b in enumerate(bufs):
            D.unpack_np(b.numpy(), final, D.packed_row_map(H, src, world, strip_rows))
        ref = oracle.render(osn, push, params, planes=["color8"])["color8"]
        ok = bool((final == ref).all())
    else:
        assert bufs is None
    # ring halo exchange: after it, every rank holds `halo` valid rows either side of each owned strip
    halo = 4
    plane = np.zeros((H, W, 4), np.uint8)
    truth = (np.arange(H)[:, None, None] * 3 + np.arange(W)[None, :, None] + np.arange(4)[None, None, :]).astype(np.uint8)
    own = D.owned_rows(H, rank, world, strip_rows)
    plane[own] = truth[own]
    up = torch.from_numpy(D.pack_np(plane, D.halo_row_map(H, rank, world, strip_rows, halo, -1)))
    down = torch.from_numpy(D.pack_np(plane, D.halo_row_map(H, rank, world, strip_rows, halo, +1)))
    from_below, from_above = D.exchange_halo(up, down)
    D.unpack_np(from_below.numpy(), plane, D.halo_row_map(H, (rank + 1) % world, world, strip_rows, halo, -1))
    D.unpack_np(from_above.numpy(), plane, D.halo_row_map(H, (rank - 1) % world, world, strip_rows, halo, +1))
    need = set()
    for y in own:
        for d in range(-halo, halo + 1):
            if 0 <= y + d < H:
                need.add(y + d)
    need = np.array(sorted(need))
    ok = ok and bool((plane[need] == truth[need]).all())
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put(int(t.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,strip_rows", [(2, 16), (3, 16)])
def test_gather_and_halo_gloo(world, strip_rows):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, strip_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 1


def _batch_worker(rank, world, port, q, mode="root"):
    """ShardedBatch's step protocol (bench.py's N > 1 path) with numpy stand-ins for the three device calls: the collective
    of step k must deliver step k's frames even though step k + 1 is "rendered" before it is completed.  mode "root": gather
    to rank 0; "owners": frame block b to rank b by one all-to-all, the strip assignment rotated per block."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import voxel_raytracing_amd as vrt
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = vrt.distributed
    owners = mode == "owners"
    W, H, SR = 40, 70, 16
    FB = 2
    F = world * FB if owners else 3
    prow = D.packed_rows(H, world, SR)
    rmap = [D.packed_row_map(H, r, world, SR) for r in range(world)]

    def truth(step):                                            # frame f of step `step`
        return np.stack([((np.arange(H)[:, None, None] * 7 + np.arange(W)[None, :, None] * 3 + np.arange(4)[None, None, :] + 11 * f + 29 * step) % 251).astype(np.uint8)
                         for f in range(F)])

    sb = D.ShardedBatch.__new__(D.ShardedBatch)                 # the protocol only: no engine, no kernels
    sb.rank, sb.nranks, sb.group, sb.F, sb.W, sb.H, sb.strip_rows = rank, world, None, F, W, H, SR
    sb.owners, sb.rotate, sb.FB, sb.host_staged = owners, owners, FB, False
    sb.packed = torch.zeros((F, prow, W, 4), dtype=torch.uint8)
    sb._root = None
    state = {"step": -1}
    mine_frames = list(sb.owned_frames())
    assert mine_frames == (list(range(rank * FB, (rank + 1) * FB)) if owners else (list(range(F)) if rank == 0 else []))
    if mine_frames:
        sb.finals = torch.zeros((len(mine_frames), H, W, 4), dtype=torch.uint8)
    sb.render = lambda pushes: state.__setitem__("step", state["step"] + 1)
    def pack():
        t = truth(state["step"])
        for f in range(F):
            vr = sb.virtual_rank(rank, f // FB) if owners else rank
            own = D.owned_rows(H, vr, world, SR)
            mine = np.zeros_like(t[f]); mine[own] = t[f][own]   # a rank holds its own rows only
            sb.packed[f] = torch.from_numpy(D.pack_np(mine, rmap[vr]))
        return sb.packed
    sb.pack = pack
    def recv_buffers():
        if sb._root is None:
            sb._root = (torch.empty_like(sb.packed) if owners else [torch.empty_like(sb.packed) for _ in range(world)],)
        return sb._root[0]
    sb.recv_buffers = recv_buffers
    def assemble():
        out = np.zeros((len(mine_frames), H, W, 4), np.uint8)
        if owners:
            for src in range(world):
                for j in range(FB):
                    D.unpack_np(sb._root[0][src * FB + j].numpy(), out[j], rmap[sb.virtual_rank(src, rank)])
        else:
            for src, b in enumerate(sb._root[0]):
                for f in range(F):
                    D.unpack_np(b[f].numpy(), out[f], rmap[src])
        sb.finals.copy_(torch.from_numpy(out))
        return sb.finals
    sb.assemble = assemble
    ok = True
    got = []
    for k in range(4):                                          # overlapped: call k completes step k - 1
        out = sb.step(None, overlap=True)
        got.append(None if out is None else out.numpy().copy())
    last = sb.finish()
    if mine_frames:
        ok = got[0] is None and all((got[k] == truth(k - 1)[mine_frames]).all() for k in (1, 2, 3)) and bool((last.numpy() == truth(3)[mine_frames]).all())
    else:
        ok = all(g is None for g in got) and last is None
    ok = ok and sb.finish() is None                              # nothing left in flight
    out = sb.step(None, overlap=False)                           # synchronous form: this step's frames at once
    if mine_frames:
        ok = ok and bool((out.numpy() == truth(4)[mine_frames]).all())
    else:
        ok = ok and out is None
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put(int(t.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["root", "owners"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_batch_step_protocol_gloo(world, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 1


def _denoise_worker(rank, world, port, q, mode):
    """ShardedBatch.step with denoise=True on CPU: the real step / denoise_step / exchange_halo / collective code over gloo,
    the device work (tracing, halo packing, the filter) replaced by numpy row maps and the oracle's denoiser."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import voxel_raytracing_amd as vrt
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = vrt.distributed
    owners = mode == "owners"
    W, H, SR, FB = 24, 70, 16, 2
    F = world * FB if owners else 3
    iters, sw0 = 2, 1.0                                        # reach 1 + 2: halo 3, extents 2 and 0
    halo = 3
    rng = np.random.default_rng(7)
    color = rng.integers(0, 256, (F, H, W, 4), dtype=np.uint8)
    normal = np.zeros((F, H, W, 4), np.int8); normal[..., 0] = 127; normal[:, :, W // 2:, 0] = 0; normal[:, :, W // 2:, 1] = 127
    pos = np.zeros((F, H, W, 4), np.float32)
    pos[..., 0] = np.arange(W)[None, None, :] * 0.25; pos[..., 1] = np.arange(H)[None, :, None] * 0.25; pos[:, H // 3:, :, 2] = 2.0
    want = np.stack([oracle.denoise(color[f], normal[f], pos[f], iterations=iters, step_width0=sw0) for f in range(F)])
    prow = D.packed_rows(H, world, SR)
    mls = D.max_local_strips(H, world, SR)

    sb = D.ShardedBatch.__new__(D.ShardedBatch)
    sb.rank, sb.nranks, sb.group, sb.F, sb.W, sb.H, sb.strip_rows = rank, world, None, F, W, H, SR
    sb.owners, sb.rotate, sb.FB, sb.host_staged, sb.denoise, sb.direct = owners, owners, FB, False, True, False
    sb.packed = torch.zeros((F, prow, W, 4), dtype=torch.uint8)
    sb._root = None
    mine_frames = list(sb.owned_frames())
    if mine_frames:
        sb.finals = torch.zeros((len(mine_frames), H, W, 4), dtype=torch.uint8)
    vr = lambda f, r=None: (sb.virtual_rank(rank if r is None else r, f // FB) if owners else (rank if r is None else r))
    planes = {}

    def render(pushes):                                        # a rank holds its own rows of every plane, zeros elsewhere
        for name, full in (("c", color), ("n", normal), ("p", pos)):
            planes[name] = np.zeros_like(full)
            for f in range(F):
                own = D.owned_rows(H, vr(f), world, SR)
                planes[name][f][own] = full[f][own]
    sb.render = render

    def pack_halos():
        ups, downs = [], []
        for name in ("c", "n", "p"):
            for f in range(F):
                a = planes[name][f].reshape(H, -1).view(np.uint8)
                ups.append(D.pack_np(a, D.halo_row_map(H, vr(f), world, SR, halo, -1)).reshape(-1))
                downs.append(D.pack_np(a, D.halo_row_map(H, vr(f), world, SR, halo, +1)).reshape(-1))
        return torch.from_numpy(np.concatenate(ups)), torch.from_numpy(np.concatenate(downs))
    sb.pack_halos = pack_halos

    def unpack_halos(from_below, from_above):
        o = 0
        for name, bpp in (("c", 4), ("n", 4), ("p", 16)):
            for f in range(F):
                n = mls * halo * W * bpp
                for buf, src, d in ((from_below, (rank + 1) % world, -1), (from_above, (rank - 1) % world, +1)):
                    a = planes[name][f].reshape(H, -1).view(np.uint8)
                    D.unpack_np(buf.numpy()[o:o + n].reshape(mls * halo, W * bpp), a, D.halo_row_map(H, vr(f, src), world, SR, halo, d))
                o += n
    sb.unpack_halos = unpack_halos

    den = {}
    def run_denoiser():                                        # the filter over what this rank holds: right on its own rows
        for f in range(F):
            den[f] = oracle.denoise(planes["c"][f], planes["n"][f], planes["p"][f], iterations=iters, step_width0=sw0)
    sb.run_denoiser = run_denoiser

    def pack():
        for f in range(F):
            sb.packed[f] = torch.from_numpy(D.pack_np(den[f], D.packed_row_map(H, vr(f), world, SR)))
        return sb.packed
    sb.pack = pack

    def recv_buffers():
        if sb._root is None:
            sb._root = (torch.empty_like(sb.packed) if owners else [torch.empty_like(sb.packed) for _ in range(world)],)
        return sb._root[0]
    sb.recv_buffers = recv_buffers

    def assemble():
        out = np.zeros((len(mine_frames), H, W, 4), np.uint8)
        if owners:
            for src in range(world):
                for j in range(FB):
                    D.unpack_np(sb._root[0][src * FB + j].numpy(), out[j], D.packed_row_map(H, sb.virtual_rank(src, rank), world, SR))
        else:
            for src, b in enumerate(sb._root[0]):
                for f in range(F):
                    D.unpack_np(b[f].numpy(), out[f], D.packed_row_map(H, src, world, SR))
        sb.finals.copy_(torch.from_numpy(out))
        return sb.finals
    sb.assemble = assemble

    out = sb.step(None, overlap=False)
    ok = True
    if mine_frames:
        ok = bool((out.numpy() == want[mine_frames]).all())
    else:
        ok = out is None
    sb.step(None, overlap=True)                                  # and the overlapped form: completed by finish()
    last = sb.finish()
    if mine_frames:
        ok = ok and bool((last.numpy() == want[mine_frames]).all())
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put(int(t.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["root", "owners"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_batch_denoise_halo_exchange_gloo(world, mode):
    """The batched, sharded denoiser path (one packed ring exchange of halo rows per step) equals the unsharded filter."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_denoise_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 1
