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
