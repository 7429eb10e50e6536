"""Screen-tile sharding of a frame over the GPUs of one node (no reference analogue; BASELINE north_star).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI; "gloo" on CPU for the tests).
The frame is cut into horizontal strips of `strip_rows` rows; strip s belongs to rank s % nranks
(interleaving balances sky against geometry).  The volume, palette, sky and noise are replicated.  The only
data-path collective is ONE per step over the packed RGBA8 strips -- a gather to rank 0, or one gather per frame
block to the rank that owns the block, issued together as a single all-to-all (ShardedBatch); the sharded denoiser adds
a ring-neighbour exchange of `halo` guide rows (strip s needs rows of strips s-1 and s+1, which live on
ranks r-1 and r+1).

The row maps here are the host mirror of vrt_pack_rows / vrt_pack_halo (csrc/vrt_device.hip: k_rows); the
CPU tests check both against each other.
"""
import ctypes as C

import numpy as np

from . import _capi


def default_strip_rows(H: int, nranks: int) -> int:
    """Rows per interleaved strip: a multiple of 16 (the C-ABI's granule), as large as still leaves every rank at least
    four strips to balance sky against geometry with, at most 64 (1080 rows: 64 at N = 2 and 4, 32 at N = 8; a frame too
    short for that gets 16-row strips, and then a rank may own fewer than four -- or none)."""
    if nranks <= 1:
        return 16
    return int(min(64, max(16, H // (4 * nranks) // 16 * 16)))


def band_strip_rows(H: int, nranks: int) -> int:
    """One strip per rank: contiguous bands of ceil(H / nranks) rows, rounded up to 16.  Only balanced when the
    assignment rotates (ShardedBatch(rotate=True)): a band of sky costs a fraction of a band of geometry."""
    return max(16, (((H + nranks - 1) // nranks) + 15) // 16 * 16)


def n_strips(H: int, strip_rows: int) -> int:
    return (H + strip_rows - 1) // strip_rows


def max_local_strips(H: int, nranks: int, strip_rows: int) -> int:
    return (n_strips(H, strip_rows) + nranks - 1) // nranks


def packed_rows(H: int, nranks: int, strip_rows: int) -> int:
    """Rows in a rank's packed buffer (identical on every rank; short ranks zero-pad)."""
    if nranks <= 1:
        return H
    return max_local_strips(H, nranks, strip_rows) * strip_rows


def packed_row_map(H: int, rank: int, nranks: int, strip_rows: int) -> np.ndarray:
    """packed row index -> frame row (or -1 for padding)."""
    if nranks <= 1:
        return np.arange(H, dtype=np.int64)
    out = np.full(packed_rows(H, nranks, strip_rows), -1, dtype=np.int64)
    for k in range(max_local_strips(H, nranks, strip_rows)):
        g = k * nranks + rank
        beg, end = g * strip_rows, min(H, (g + 1) * strip_rows)
        if beg < H:
            out[k * strip_rows:k * strip_rows + (end - beg)] = np.arange(beg, end)
    return out


def owned_rows(H: int, rank: int, nranks: int, strip_rows: int) -> np.ndarray:
    m = packed_row_map(H, rank, nranks, strip_rows)
    return m[m >= 0]


def halo_row_map(H: int, rank: int, nranks: int, strip_rows: int, halo: int, direction: int) -> np.ndarray:
    """Rows of vrt_pack_halo: first (direction -1) / last (+1) `halo` rows of every strip `rank` owns."""
    out = np.full(max_local_strips(H, nranks, strip_rows) * halo, -1, dtype=np.int64)
    for k in range(max_local_strips(H, nranks, strip_rows)):
        g = k * nranks + rank
        beg, end = g * strip_rows, min(H, (g + 1) * strip_rows)
        for j in range(halo):
            y = beg + j if direction < 0 else end - halo + j
            if beg <= y < end:
                out[k * halo + j] = y
    return out


def pack_np(full: np.ndarray, row_map: np.ndarray) -> np.ndarray:
    out = np.zeros((len(row_map),) + full.shape[1:], dtype=full.dtype)
    ok = row_map >= 0
    out[ok] = full[row_map[ok]]
    return out


def unpack_np(packed: np.ndarray, full: np.ndarray, row_map: np.ndarray) -> None:
    ok = row_map >= 0
    full[row_map[ok]] = packed[ok]


# ---- collectives -----------------------------------------------------------------------------------

def gather_packed(packed, dst: int = 0, group=None):
    """One collective per step: gather every rank's packed strips on `dst`.  Returns the list of per-rank
    tensors on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return [packed]
    if rank == dst:
        bufs = [torch.empty_like(packed) for _ in range(world)]
        dist.gather(packed, gather_list=bufs, dst=dst, group=group)
        return bufs
    dist.gather(packed, gather_list=None, dst=dst, group=group)
    return None


def exchange_halo(send_up, send_down, group=None):
    """Ring-neighbour exchange.  send_up (first rows of my strips) goes to rank-1, send_down (last rows) to
    rank+1.  Returns (from_below, from_above): what rank+1 / rank-1 sent me."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    up, down = (rank - 1) % world, (rank + 1) % world
    from_below = torch.empty_like(send_up)
    from_above = torch.empty_like(send_down)
    if world == 1:
        from_below.copy_(send_up); from_above.copy_(send_down)
        return from_below, from_above
    ops = [dist.P2POp(dist.isend, send_up, up, group), dist.P2POp(dist.irecv, from_below, down, group),
           dist.P2POp(dist.isend, send_down, down, group), dist.P2POp(dist.irecv, from_above, up, group)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    return from_below, from_above


# ---- device-side sharded frame -----------------------------------------------------------------------

class ShardedFrame:
    """Per-rank driver: render owned strips, (optionally) exchange halos + denoise, gather RGBA8 on rank 0."""

    def __init__(self, renderer, rank: int, nranks: int, strip_rows: int = None, group=None):
        import torch
        self.r = renderer
        self.rank, self.nranks, self.group = int(rank), int(nranks), group
        W, H = renderer.settings.renderResolution()
        self.W, self.H = W, H
        self.strip_rows = strip_rows or default_strip_rows(H, nranks)
        self.shard = _capi.Shard(self.rank, self.nranks, self.strip_rows) if nranks > 1 else None
        dev = renderer.engine.torch_device
        self.prow = packed_rows(H, nranks, self.strip_rows)
        self.packed = torch.zeros((self.prow, W, 4), dtype=torch.uint8, device=dev)
        self.final = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
        self._halo_bufs = {}

    def _lib(self):
        return _capi.lib()

    def _halo_exchange(self, gb):
        """Fill the guide rows just outside my strips from the ring neighbours (color, normal, position)."""
        import torch
        l, ctx = self._lib(), self.r.engine.ctx
        ds = self.r.settings.denoiser_to_c()
        halo = l.vrt_denoise_halo_rows(C.byref(ds))
        if halo <= 0 or self.nranks <= 1:
            return
        if halo > self.strip_rows:
            raise ValueError(f"denoiser halo ({halo} rows) exceeds strip_rows ({self.strip_rows}); use larger strips")
        mls = max_local_strips(self.H, self.nranks, self.strip_rows)
        up_rank, down_rank = (self.rank - 1) % self.nranks, (self.rank + 1) % self.nranks
        for name, bpp in (("color8", 4), ("normal8", 4), ("position", 16)):
            full = gb.planes[name]
            key = (name, halo)
            if key not in self._halo_bufs:
                mk = lambda: torch.zeros((mls * halo, self.W, bpp), dtype=torch.uint8, device=full.device)
                self._halo_bufs[key] = (mk(), mk())
            s_up, s_down = self._halo_bufs[key]
            _capi.check(l.vrt_pack_halo(ctx, full.data_ptr(), s_up.data_ptr(), self.W, self.H, bpp, C.byref(self.shard), halo, -1))
            _capi.check(l.vrt_pack_halo(ctx, full.data_ptr(), s_down.data_ptr(), self.W, self.H, bpp, C.byref(self.shard), halo, 1))
            from_below, from_above = exchange_halo(s_up, s_down, self.group)
            sh_below = _capi.Shard(down_rank, self.nranks, self.strip_rows)   # sender of from_below
            sh_above = _capi.Shard(up_rank, self.nranks, self.strip_rows)
            _capi.check(l.vrt_unpack_halo(ctx, from_below.data_ptr(), full.data_ptr(), self.W, self.H, bpp, C.byref(sh_below), halo, -1))
            _capi.check(l.vrt_unpack_halo(ctx, from_above.data_ptr(), full.data_ptr(), self.W, self.H, bpp, C.byref(sh_above), halo, 1))

    def render_local(self):
        """Trace (and denoise) the rows this rank owns; returns the full-frame RGBA8 plane (own rows valid)."""
        r = self.r
        push = r.push_constants()
        gb = r._geometryStage.record(push, self.shard)
        r.gBuffer = gb
        color = gb.color
        if r.settings.denoiserSettings.enable:
            self._halo_exchange(gb)
            color = r._denoiserStage.record(gb.color, gb.normal, gb.position, self.shard)
        return color

    def pack(self, color_full):
        l, ctx = self._lib(), self.r.engine.ctx
        if self.nranks <= 1:
            return color_full
        _capi.check(l.vrt_pack_rows(ctx, color_full.data_ptr(), self.packed.data_ptr(), self.W, self.H, 4, C.byref(self.shard)))
        return self.packed

    def gather(self, packed):
        """RCCL gather of the packed strips to rank 0 and de-interleave there.  Returns the final image on rank 0."""
        l, ctx = self._lib(), self.r.engine.ctx
        if self.nranks <= 1:
            return packed
        bufs = gather_packed(packed, 0, self.group)
        if self.rank != 0:
            return None
        for src, b in enumerate(bufs):
            sh = _capi.Shard(src, self.nranks, self.strip_rows)
            _capi.check(l.vrt_unpack_rows(ctx, b.data_ptr(), self.final.data_ptr(), self.W, self.H, 4, C.byref(sh)))
        return self.final

    def step(self):
        return self.gather(self.pack(self.render_local()))


class _BandFrames:
    """ShardedBatch(in_place=True).finals: the finished frames of this rank's block as they lie in the receive buffer.
    bands(j): the row bands of frame j in frame order, [(first row, tensor [rows, W, 4] -- a view), ...];
    self[j]: a contiguous [H, W, 4] copy, assembled when asked for (tests, the bench's comparison with single-GPU renders)."""

    def __init__(self, sb):
        self.sb = sb

    def __len__(self):
        return self.sb.FB

    def bands(self, j: int):
        sb = self.sb
        buf = sb._recv[sb._recv_cur]
        out = []
        for s in range(sb.nranks):                             # source s traced this block as virtual rank vr: band vr of the frame
            vr = sb.virtual_rank(s, sb.rank)
            r0 = vr * sb.strip_rows
            rows = max(0, min(sb.strip_rows, sb.H - r0))
            if rows:
                out.append((r0, buf[s * sb.FB + j][:rows]))
        return sorted(out, key=lambda t: t[0])

    def __getitem__(self, j: int):
        import torch
        sb = self.sb
        img = torch.empty((sb.H, sb.W, 4), dtype=torch.uint8, device=sb.packed.device)
        for r0, t in self.bands(j):
            img[r0:r0 + t.shape[0]].copy_(t)
        return img


class ShardedBatch:
    """Per-rank driver for a batch of F frames (consecutive camera poses): ONE K1 launch over this rank's strips of all of
    them (vrt_render_geometry_batch / _slots), strip packing, ONE collective per step, strip unpacking at the receivers.
    denoise=True runs the sharded denoiser on every frame of the batch between the tracing and the collective: ONE ring
    exchange per step carries the halo rows (first / last `halo` rows of every owned strip) of the colour, normal and position
    planes of ALL frames to the two ring neighbours -- two packed buffers out, two in (exchange_halo) -- then each frame is
    filtered on its owner's rows (vrt_denoise with the frame's strip assignment) and the filtered colour is what is packed.
    With the rotating assignment block b is traced as virtual rank (rank + b) % N by every rank, so a strip's neighbours are
    the same two real ranks for every block.

    assemble_on = "root":   every frame is gathered to rank 0 (dist.gather): the single-display case.  Rank 0 receives
                            (N-1)/N of every frame of the batch over its 7 inbound xGMI links, so the batch rate is bound by
                            those links once N x frame rate x frame bytes exceeds them.
    assemble_on = "owners": frame block b (frames b*F/N .. (b+1)*F/N - 1) is gathered to rank b -- N gathers with N different
                            roots issued as one all-to-all (dist.all_to_all_single), so every GPU receives over all of its
                            links at once and finished frames end up spread over the ranks (each process encodes / writes
                            its own).  xGMI is point-to-point and fully connected: this is the layout that scales on it.
                            With rotate=True (default in this mode) block b is traced with the strip assignment rotated by
                            b (this rank plays rank (rank + b) % N): when the strips do not divide evenly (1080 rows =
                            67.5 strips of 16 over 8 ranks: 9 or 8 each) every rank still traces the same number of rows
                            per step.
    direct = True:          K1 itself writes the colour a second time in packed-strip order (vrt_frame.color8_strips) into
                            the send buffer, so pack() has nothing left to do; the send buffer is double-buffered because
                            the collective of step k is still reading it while step k + 1 is traced.  direct = "only":
                            the packed copy is the only colour K1 stores (the GeometryBuffers' colour planes stay
                            unwritten).  direct = False keeps the copy kernel (vrt_pack_rows_batch) between K1 and the
                            collective."""

    def __init__(self, stage, n_frames: int, rank: int, nranks: int, strip_rows: int = None, group=None, host_staged: bool = False,
                 assemble_on: str = "root", rotate: bool = None, direct: bool = True, side_unpack: bool = False, denoise: bool = False,
                 in_place: bool = False):
        import torch
        if denoise and not stage._settings.denoiserSettings.enable:
            raise ValueError("ShardedBatch(denoise=True) with the denoiser switched off in the stage's settings (denoiserSettings.enable)")
        # N = 1: the same call returns filtered frames as at N > 1 -- the unsharded denoiser on every frame of the batch (step())
        self._denoise_alone = bool(denoise) and int(nranks) <= 1
        self.denoise = bool(denoise) and int(nranks) > 1
        if self.denoise:
            direct = False                                     # the colour that travels is the denoiser's output, not K1's
        # (the EFFECTIVE direct: a sharded denoising batch never uses the direct frame table, whatever the argument says)
        if direct and int(nranks) > 1 and getattr(stage, "_debug", False):
            raise ValueError("ShardedBatch(direct=...) launches write the frames' planes through its own frame table, which carries no "
                             "diagnostic planes (hit_voxel, steps_*): use direct=False with a debug_planes stage")
        self.stage, self.F = stage, int(n_frames)
        self.host_staged = bool(host_staged)       # collective through host memory (gloo rehearsal of the N > 1 path on one GPU)
        self.side_unpack = bool(side_unpack) and not self.host_staged      # assemble received frames on a second stream
        self.rank, self.nranks, self.group = int(rank), int(nranks), group
        if assemble_on not in ("root", "owners"):
            raise ValueError("assemble_on must be 'root' or 'owners'")
        self.owners = assemble_on == "owners" and self.nranks > 1
        self.rotate = bool(self.owners if rotate is None else rotate) and self.owners
        W, H = stage._settings.renderResolution()
        self.W, self.H = W, H
        # with the rotation every rank traces every part of the screen once per step, so the strips need not interleave to
        # balance sky against geometry: one contiguous band per rank keeps a rank's rays next to each other
        self.strip_rows = strip_rows or (band_strip_rows(H, nranks) if self.rotate else default_strip_rows(H, nranks))
        self.shard = _capi.Shard(self.rank, self.nranks, self.strip_rows) if nranks > 1 else None
        dev = stage.engine.torch_device
        self.prow = packed_rows(H, nranks, self.strip_rows)
        P = C.c_void_p
        if not self.owners:
            self.launch = stage.prepare_batch(self.F, self.shard)
            self._blocks = None
        else:
            if self.F % self.nranks:
                raise ValueError(f"assemble_on='owners' needs the batch ({self.F} frames) to divide over the {self.nranks} ranks")
            self.FB = self.F // self.nranks
            # the strip assignment of every block; ONE launch traces all blocks (vrt_render_geometry_slots)
            self._blocks = [_capi.Shard(self.virtual_rank(self.rank, b), self.nranks, self.strip_rows) for b in range(self.nranks)]
            self.launch = stage.prepare_batch(self.F, shards=[self._blocks[f // self.FB] for f in range(self.F)])
        self.gbs = self.launch._keepalive[4]
        self.direct = bool(direct) and nranks > 1
        if nranks > 1:
            self.packed = torch.zeros((self.F, self.prow, W, 4), dtype=torch.uint8, device=dev)
            self._full_ptrs = (P * self.F)(*[g.color.data_ptr() for g in self.gbs])
            self._packed_ptrs = (P * self.F)(*[self.packed[f].data_ptr() for f in range(self.F)])
            self._root = None
            if self.direct:
                self._send = [self.packed, torch.zeros_like(self.packed)]
                self._frames = []
                frs = self.launch._keepalive[1]
                for buf in self._send:
                    tab = (_capi.Frame * self.F)()
                    C.memmove(tab, frs, C.sizeof(tab))
                    for f in range(self.F):
                        tab[f].color8_strips = buf[f].data_ptr()
                        if direct == "only":
                            tab[f].color8 = None
                    self._frames.append(tab)
                self._cur = 1
            # in_place ("owners" with one band per rank): a finished frame STAYS where the collective put it -- N row bands, each a
            # contiguous block of the receive buffer, in a known order -- instead of being copied into one [H, W, 4] image: the
            # copy is a pass over every frame at HBM speed (23 us of a 337 us step at N = 8) for the benefit of a consumer that can
            # just as well walk eight chunks (bands(j)); finals[j] assembles a contiguous copy when somebody asks for one.
            self.in_place = bool(in_place) and self.owners and max_local_strips(H, nranks, self.strip_rows) == 1
            if self.in_place:
                self.finals = _BandFrames(self)
                self._recv = None
                self._recv_cur = 0
            elif self.owners:
                self.finals = torch.zeros((self.FB, H, W, 4), dtype=torch.uint8, device=dev)
            elif rank == 0:
                self.finals = torch.zeros((self.F, H, W, 4), dtype=torch.uint8, device=dev)

    # -- sharded denoiser inside the batch ------------------------------------------------------------------------------
    def frame_shard(self, f: int, rank: int = None) -> "_capi.Shard":
        """The strip assignment frame f of the batch is traced with by `rank` (default: this rank)."""
        rank = self.rank if rank is None else rank
        vr = self.virtual_rank(rank, f // self.FB) if self.owners else rank
        return _capi.Shard(vr, self.nranks, self.strip_rows)

    def _halo_setup(self):
        import torch
        lib = _capi.lib()
        ds = self.stage._settings.denoiser_to_c()
        self._ds = ds
        self.halo = int(lib.vrt_denoise_halo_rows(C.byref(ds)))
        if self.halo > self.strip_rows:
            raise ValueError(f"denoiser halo ({self.halo} rows) exceeds strip_rows ({self.strip_rows}); use larger strips")
        mls = max_local_strips(self.H, self.nranks, self.strip_rows)
        dev = self.stage.engine.torch_device
        P = C.c_void_p
        self._planes = (("color8", 4), ("normal8", 4), ("position", 16))
        rows = mls * self.halo
        # [plane][frame][rows][W][bpp] in ONE buffer per direction: colour, normal, position of every frame
        sizes = [self.F * rows * self.W * bpp for _, bpp in self._planes]
        self._halo_off = [0, sizes[0], sizes[0] + sizes[1]]
        total = sum(sizes)
        self._send_up, self._send_down = torch.zeros(total, dtype=torch.uint8, device=dev), torch.zeros(total, dtype=torch.uint8, device=dev)
        up, down = (self.rank - 1) % self.nranks, (self.rank + 1) % self.nranks
        self._halo_tabs = []
        for k, (name, bpp) in enumerate(self._planes):
            full = (P * self.F)(*[g.planes[name].data_ptr() for g in self.gbs])
            def sub(buf, k=k, bpp=bpp):
                return (P * self.F)(*[buf.data_ptr() + self._halo_off[k] + f * rows * self.W * bpp for f in range(self.F)])
            mine = (_capi.Shard * self.F)(*[self.frame_shard(f) for f in range(self.F)])
            below = (_capi.Shard * self.F)(*[self.frame_shard(f, down) for f in range(self.F)])     # sender of from_below
            above = (_capi.Shard * self.F)(*[self.frame_shard(f, up) for f in range(self.F)])
            self._halo_tabs.append((bpp, full, sub, mine, below, above))
        self._den_targets = [[torch.zeros_like(g.color), torch.zeros_like(g.color)] for g in self.gbs]
        self._den_out = [None] * self.F

    def pack_halos(self):
        """First / last `halo` rows of every owned strip of every frame, three planes, into the two send buffers."""
        if getattr(self, "_halo_tabs", None) is None:
            self._halo_setup()
        lib, ctx = _capi.lib(), self.stage.engine.ctx
        for bpp, full, sub, mine, _, _ in self._halo_tabs:
            _capi.check(lib.vrt_pack_halo_batch(ctx, self.F, full, sub(self._send_up), self.W, self.H, bpp, mine, self.halo, -1))
            _capi.check(lib.vrt_pack_halo_batch(ctx, self.F, full, sub(self._send_down), self.W, self.H, bpp, mine, self.halo, 1))
        return self._send_up, self._send_down

    def unpack_halos(self, from_below, from_above):
        """The neighbours' rows into the guide planes, just outside this rank's strips."""
        lib, ctx = _capi.lib(), self.stage.engine.ctx
        P = C.c_void_p
        rows = max_local_strips(self.H, self.nranks, self.strip_rows) * self.halo
        for k, (bpp, full, _, _, below, above) in enumerate(self._halo_tabs):
            def sub(buf, k=k, bpp=bpp):
                return (P * self.F)(*[buf.data_ptr() + self._halo_off[k] + f * rows * self.W * bpp for f in range(self.F)])
            _capi.check(lib.vrt_unpack_halo_batch(ctx, self.F, sub(from_below), full, self.W, self.H, bpp, below, self.halo, -1))
            _capi.check(lib.vrt_unpack_halo_batch(ctx, self.F, sub(from_above), full, self.W, self.H, bpp, above, self.halo, 1))

    def run_denoiser(self):
        """vrt_denoise on every frame's own rows; afterwards pack() takes the filtered colour."""
        lib, ctx = _capi.lib(), self.stage.engine.ctx
        for f, g in enumerate(self.gbs):
            res = C.c_void_p()
            sh = self.frame_shard(f)
            t0, t1 = self._den_targets[f]
            _capi.check(lib.vrt_denoise(ctx, self.W, self.H, C.byref(self._ds), g.color.data_ptr(), g.normal.data_ptr(), g.position.data_ptr(),
                                        t0.data_ptr(), t1.data_ptr(), C.byref(sh), C.byref(res)))
            self._den_out[f] = t0 if res.value == t0.data_ptr() else (t1 if res.value == t1.data_ptr() else g.color)
        P = C.c_void_p
        self._full_ptrs = (P * self.F)(*[t.data_ptr() for t in self._den_out])
        self._pack_tabs = None

    def denoise_step(self):
        """Between render() and pack(): one packed ring exchange of the halo rows, then the sharded filter."""
        up, down = self.pack_halos()
        from_below, from_above = exchange_halo(up, down, self.group)
        self.unpack_halos(from_below, from_above)
        self.run_denoiser()

    def virtual_rank(self, rank: int, block: int) -> int:
        """The strip assignment `rank` traces frame block `block` with."""
        return (rank + block) % self.nranks if self.rotate else rank

    def owned_frames(self) -> range:
        """Batch indices of the frames that end up in self.finals on this rank."""
        if self.nranks <= 1 or not self.owners:
            return range(self.F) if self.rank == 0 else range(0)
        return range(self.rank * self.FB, (self.rank + 1) * self.FB)

    def render(self, pushes):
        """This rank's strips of every frame; returns the GeometryBuffers (own rows valid).  direct: self.packed is the send
        buffer this call filled."""
        if not getattr(self, "direct", False):
            return self.launch(pushes)
        self._cur ^= 1
        self.packed = self._send[self._cur]
        return self.launch(pushes, self._frames[self._cur])

    def pack(self):
        if getattr(self, "direct", False):           # render() wrote the strips in place
            return self.packed
        lib, ctx = _capi.lib(), self.stage.engine.ctx
        if self._blocks is None:
            _capi.check(lib.vrt_pack_rows_batch(ctx, self.F, self._full_ptrs, self._packed_ptrs, self.W, self.H, 4, C.byref(self.shard)))
            return self.packed
        P = C.c_void_p
        if getattr(self, "_pack_tabs", None) is None:      # per block: its slice of the pointer tables
            self._pack_tabs = [((P * self.FB)(*self._full_ptrs[b * self.FB:(b + 1) * self.FB]),
                                (P * self.FB)(*self._packed_ptrs[b * self.FB:(b + 1) * self.FB])) for b in range(self.nranks)]
        for (full, packed), sh in zip(self._pack_tabs, self._blocks):
            _capi.check(lib.vrt_pack_rows_batch(ctx, self.FB, full, packed, self.W, self.H, 4, C.byref(sh)))
        return self.packed

    def recv_buffers(self):
        """The receive side, built once.  "root" (rank 0): one [F, packed rows, W, 4] buffer per source and the pointer
        tables of the (source, frame) unpack.  "owners" (every rank): one buffer of the same shape as self.packed, laid out
        [source][frame of my block], and the tables of its unpack into self.finals."""
        import torch
        if getattr(self, "in_place", False):
            if self._recv is None:                             # two receive buffers: step k's frames are read while step k + 1 arrives
                self._recv = [torch.empty_like(self.packed), torch.empty_like(self.packed)]
            return self._recv[self._recv_cur ^ 1]              # the one the NEXT collective fills
        if self._root is None:
            P = C.c_void_p
            if self.owners:
                buf = torch.empty_like(self.packed)
                n = self.F
                src = (P * n)(*[buf[s * self.FB + j].data_ptr() for s in range(self.nranks) for j in range(self.FB)])
                dst = (P * n)(*[self.finals[j].data_ptr() for s in range(self.nranks) for j in range(self.FB)])
                shards = (_capi.Shard * n)(*[_capi.Shard(self.virtual_rank(s, self.rank), self.nranks, self.strip_rows)
                                             for s in range(self.nranks) for j in range(self.FB)])
                self._root = (buf, n, src, dst, shards)
            else:
                bufs = [torch.empty_like(self.packed) for _ in range(self.nranks)]
                n = self.F * self.nranks
                src = (P * n)(*[bufs[s][f].data_ptr() for s in range(self.nranks) for f in range(self.F)])
                dst = (P * n)(*[self.finals[f].data_ptr() for s in range(self.nranks) for f in range(self.F)])
                shards = (_capi.Shard * n)(*[_capi.Shard(s, self.nranks, self.strip_rows) for s in range(self.nranks) for f in range(self.F)])
                self._root = (bufs, n, src, dst, shards)
        return self._root[0]

    def assemble(self, ctx=None):
        """After the collective filled recv_buffers(): the received strips into self.finals (one launch per 64 (source, frame)
        pairs).  Rank 0 in "root" mode, every rank in "owners" mode.  ctx: another context (stream) to run it on."""
        if getattr(self, "in_place", False):                   # the collective that just completed filled the other buffer: it is current now
            self._recv_cur ^= 1
            return self.finals
        _, n, src, dst, shards = self._root
        _capi.check(_capi.lib().vrt_unpack_rows_batch(ctx or self.stage.engine.ctx, n, src, dst, self.W, self.H, 4, shards))
        return self.finals

    def _side_stream(self):
        """A second stream with a context of its own for the unpack (side_unpack): the strided copy of a step's frames is
        HBM work that fits under the next step's tracing, which is instruction-bound."""
        if getattr(self, "_side", None) is None:
            import torch
            from .host import Engine
            dev = self.stage.engine.torch_device
            self._side = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(self._side):
                self._side_engine = Engine(self.stage.engine.device)      # binds its context to the current (= side) stream
            self._side_engine.set_timing(False)
            self._ev_arrived = torch.cuda.Event()
            self._ev_unpacked = torch.cuda.Event()
            self._unpack_pending = False
        return self._side

    def _receives(self) -> bool:
        return self.owners or self.rank == 0

    def _collective(self, send, recv, async_op):
        import torch.distributed as dist
        if self.owners:
            return dist.all_to_all_single(recv, send, group=self.group, async_op=async_op)
        return dist.gather(send, gather_list=recv, dst=0, group=self.group, async_op=async_op)

    def gather(self):
        """One RCCL collective over all frames' packed strips (gather to rank 0, or the all-to-all of "owners"); the
        receivers assemble their frames into self.finals."""
        bufs = self.recv_buffers() if self._receives() else None
        self._collective(self.packed, bufs, False)
        return self.assemble() if self._receives() else None

    # -- the collective of step k overlapped with the tracing of step k + 1 -------------------------------------------
    def start_gather(self):
        """Enqueue the collective without making the launch stream wait for it (async_op: it runs on RCCL's own stream,
        ordered after the pack by an event)."""
        bufs = self.recv_buffers() if self._receives() else None
        if getattr(self, "host_staged", False):                  # rehearsal: device -> host, host collective, host -> device
            src = self.packed.cpu()
            if self.owners:
                hb = src.new_empty(src.shape)
            else:
                hb = [src.new_empty(src.shape) for _ in range(self.nranks)] if self.rank == 0 else None
            self._work = self._collective(src, hb, True)
            self._host = (src, hb, bufs)
            return
        if getattr(self, "_unpack_pending", False):              # the receive buffers are still being read by the side stream
            import torch
            torch.cuda.current_stream().wait_event(self._ev_unpacked)
            self._unpack_pending = False
        self._work = self._collective(self.packed, bufs, True)

    def finish(self):
        """Make the launch stream wait for the collective in flight (if any) and assemble its frames on the receivers.
        Returns self.finals there when a collective was completed, else None.  With side_unpack the assembly runs on a second
        stream (self.finals is complete once that stream is: torch.cuda.synchronize(), or self.wait_finals())."""
        w = getattr(self, "_work", None)
        if w is None:
            return None
        if getattr(self, "side_unpack", False) and self._receives() and not getattr(self, "host_staged", False):
            import torch
            side = self._side_stream()
            with torch.cuda.stream(side):
                w.wait()                                         # the SIDE stream waits for the collective ...
                self._ev_arrived.record()
                self.assemble(self._side_engine.ctx)             # ... and unpacks while the launch stream traces on
                self._ev_unpacked.record()
            self._unpack_pending = True
            torch.cuda.current_stream().wait_event(self._ev_arrived)     # the launch stream: that collective's send buffer is free
            self._work = None
            return self.finals
        w.wait()
        self._work = None
        if getattr(self, "host_staged", False) and self._receives():
            _, hb, bufs = self._host
            if self.owners:
                bufs.copy_(hb)
            else:
                for b, h in zip(bufs, hb):
                    b.copy_(h)
        return self.assemble() if self._receives() else None

    def wait_finals(self):
        """Make the current stream wait for a side-stream assembly in flight (side_unpack)."""
        if getattr(self, "_unpack_pending", False):
            import torch
            torch.cuda.current_stream().wait_event(self._ev_unpacked)

    def step(self, pushes, overlap: bool = True):
        """One step.  overlap=True: this step's K1 is launched first and runs while the previous step's collective is still
        in flight; then that one is completed (the receivers assemble ITS frames), this step's strips are packed and their
        collective is started.  The caller ends a sequence with finish().  Returns the frames completed by this call on a
        receiver (the previous step's with overlap, this step's without; None elsewhere / on the first overlapped step);
        N = 1: this step's GeometryBuffers."""
        gbs = self.render(pushes)
        if self.nranks <= 1:
            if self._denoise_alone:                            # what N > 1 returns with denoise=True: the filtered colour of every frame
                from .host import DenoiserStage
                if not hasattr(self, "_alone"):
                    self._alone = [DenoiserStage(self.stage.engine, self.stage._settings) for _ in gbs]
                return [d.record(g.color, g.normal, g.position) for d, g in zip(self._alone, gbs)]
            return gbs
        if getattr(self, "denoise", False):
            self.denoise_step()
        if not overlap:
            self.pack()
            return self.gather()
        out = self.finish()
        self.pack()
        self.start_gather()
        return out
