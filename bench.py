#!/usr/bin/env python3
"""bench.py -- BASELINE metric: Mrays/s at 1080p on the treehouse scene (primary rays), plus the achieved
algorithmic GB/s of the primary-ray DDA kernel against the MI355X HBM peak.

  python bench.py --gpus N --steps K --warmup W                  (N > 1: starts the N ranks itself, one per GPU)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: a batch of N x F frames (N = number of GPUs,
F = --frames-per-gpu, default 128 on one GPU and 32 on several -- a rank's launch then has 128 ... 256 frame slots at every N; consecutive camera poses of the same 8-unit dolly move at every N and F), every frame cut
into screen strips that are dealt to the N ranks.  A rank traces its strips of ALL frames of the batch with ONE K1 launch
(vrt_render_geometry_batch / _slots: the next frame's tiles are dispatched while the previous frame drains), packs
them, and ONE RCCL collective per step moves the strips to where the frames are assembled: frame block b (F frames) is
gathered to rank b, the N gathers issued as a single all-to-all so that every GPU receives over all of its xGMI links
(VRT_ASSEMBLE=root: everything to rank 0 with one dist.gather instead -- bound by rank 0's inbound links).  A finished frame
stays where the collective put it: N row bands, each a contiguous block of the receive buffer, in row order
(ShardedBatch(in_place=True); finals[j] makes a contiguous copy when asked, as the end-of-run comparison does).  The strip
assignment is rotated per block, so every rank traces the same number of rows per step although 1080 rows are 67.5
strips.  The collective of a step runs while the next step is traced.  Per-GPU work per step is therefore F frames'
worth of rays at every N ("weak" scaling); value = total primary rays of all ranks / wall time.  At N = 1 there is no
collective and no copy.

Workload = BASELINE.json configs[1]: treehouse stand-in (synthetic:treehouse(seed=2), 256^3 -- the real
treehouse.vox is a git-LFS pointer in the reference checkout), 1920x1080, primary rays only.

Three byte counts per launch, all over the kernel's measured duration and the 8 TB/s peak (DESIGN.md 7 has the table):
  roofline.frac            SURVEY 8(d)'s accounting over the work the kernel PERFORMS: 1 B per DDA iteration the product march takes (every
                           frame of the launch, counted by the counting twins of the loops that are timed) + 37 B per pixel.  An iteration of
                           a clearance run is an ADDITION IN A REGISTER, not a fetch: this figure measures arithmetic in byte units.
  roofline.frac_requested  the bytes lanes really ask memory for: one per clearance look-up of a live lane, one per voxel id read, + the
                           37 B per pixel stored.  This is the memory statement; it agrees with the counters (hbm_frac) to within the
                           G-buffer's write-combining.
  roofline.hbm_frac        what the counters saw: FETCH_SIZE / WRITE_SIZE of the same command (profiles/r04_k_primary_pmc.json, quoted only
                           while its csrc digest matches).
roofline.frac_reference_steps keeps 8(d)'s formula over the iterations of the REFERENCE's loop (the product proves most of them
unnecessary and does not take them: the figure passes 1 and is not a bound); roofline.bound_in_practice says what the kernel is
bound by -- vector-instruction issue -- with the counters behind it; roofline.write_floor_ms is the stored bytes at the 6.1 TB/s
plain stores reach on this part.

Besides the headline the JSON line carries (rank 0, N = 1, outside the timed region):
  roofline.single_frame_launch   the same kernel with ONE frame per launch (the reference's call pattern, engine.cpp:81-92)
  extra_configs                  BASELINE configs[2] (shadow ray + 1 and 2 denoiser passes) and the reference's default
                                 workload (AO 4 x 64, shadow, <= 5 bounces, 2 passes; voxel_render_settings.hpp:21-35):
                                 per-kernel ms, rays, DDA steps and the same algorithmic-byte accounting
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured-achievable
N_SIMD, CLOCK_HZ, VALU_CYCLES = 1024, 2.4e9, 4    # 256 CUs x 4 SIMDs; a full-width wave64 vector instruction occupies its SIMD for 4 cycles
STORE_GBS = 6100.0             # what plain stores reach on this part (same guide): the floor of a kernel that only writes its G-buffer
B_OUT = 37                     # bytes stored per pixel: the reference's 6-target G-buffer (geometry_stage.cpp:22-33)
K3_BYTES_PASS0 = 8             # denoiser pass 0 (phi = +inf): colour in + colour out
K3_BYTES_PASS = 28             # weighted pass: colour + normal + position in (4 + 4 + 16), colour out
KERNEL_SOURCES = ("vrt_device.hip", "vrt_traverse.h", "vrt_spec.h", "vrt_sky.h", "vrt_tags.h", "vrt_denoise_bound.h", "vrt_internal.h", "vrt_api.hip", "Makefile")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def csrc_sha16():
    """Identity of the kernel sources a PMC summary was collected on (tools/pmc_summary.py stores the same digest)."""
    h = hashlib.sha256()
    for n in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "voxel-raytracing_amd", "csrc", n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--volume", type=int, default=256)
    ap.add_argument("--traversal", default="AUTO", choices=["AUTO", "DENSE", "BITMASK", "JUMP", "DF", "DFJ"])
    ap.add_argument("--frames-per-gpu", type=int, default=None,
                    help="frames of the batch per GPU and step; default 128 on one GPU, 32 on several (a rank holds the planes of "
                         "ALL N x F frames of a step: 77 MB each at 1080p)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip single_frame_launch and extra_configs (profiling runs)")
    return ap.parse_args()


# ---- launcher: --gpus N without a launcher around us -------------------------------------------------------------------

def spawn_ranks(n):
    """Start n fresh copies of this command, one rank per GPU, BEFORE anything here touches torch or HIP; relay rank 0's
    stdout (the JSON line); a rank that fails takes the others down and its exit code becomes ours."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import threading
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained WHILE the ranks run (a reader thread): a pipe holds about 64 KiB, and a rank 0 that wrote more
    # would block on write while this process waits for it to exit
    chunks = []

    def drain():
        for line in iter(procs[0].stdout.readline, b""):
            chunks.append(line)
    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        time.sleep(0.05)
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = code if code > 0 else 1
                log(f"bench.py: rank {r} exited with code {code}; stopping the other ranks")
                break
    for r in sorted(live):                                    # only after a failure: the exact children started above
        procs[r].terminate()
    for r in sorted(live):
        try:
            procs[r].wait(timeout=10)
        except Exception:
            procs[r].kill()
    reader.join(timeout=10)
    out = b"".join(chunks).decode(errors="replace")
    if rc == 0:
        sys.stdout.write(out)
        sys.stdout.flush()
    elif out:                                                  # a failed run: what rank 0 said goes to stderr, never lost
        log("bench.py: rank 0 wrote before the failure:\n" + out)
    return rc


def launcher_selftest(world, rank):
    """VRT_BENCH_LAUNCH_ONLY=1: the ranks only rendezvous (gloo) and agree on a sum -- exercises spawn_ranks without a GPU
    (tests/test_bench_launcher.py).  VRT_BENCH_FAIL_RANK=k makes rank k fail before the rendezvous."""
    if os.environ.get("VRT_BENCH_FAIL_RANK", "") == str(rank):
        sys.exit(3)
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        total = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        total = 1
    if rank == 0:
        print(json.dumps({"launcher_test": True, "n_gpus": world, "rank_sum": total}), flush=True)


# ---- the measured legs ---------------------------------------------------------------------------------------------------

def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def marched_counts(vrt, torch, engine, scene, st, push, W, H, lookups=False):
    """(DDA iterations of the primary rays, of all rays, rays traced) as the PRODUCT march performs them for one frame:
    VRT_FLAG_MARCHED_COUNTS makes the count planes report the march's own work -- rays end at open cells, untagged blocks are not
    traced, an any-hit ray decided at a look-up reports the iterations it took -- where they otherwise hold the REFERENCE loop's.
    The launch runs the counting twins of the very loops the timed launch runs (threshold runs included: vrt_traverse.h CNT).
    lookups=True (VRT_FLAG_LOOKUP_COUNTS): the BYTES the march asks for instead -- one per clearance look-up of a live lane, one per
    voxel id read: an iteration of a clearance run is an addition in a register, not a fetch."""
    import ctypes as C
    gbm = vrt.GeometryBuffer(engine, W, H, ("steps_primary", "steps_total", "rays_total"))
    stc = st.to_c(); stc.flags |= 16 | (32 if lookups else 0)
    frm = gbm.to_c()
    vrt._capi.check(vrt.lib().vrt_render_geometry(engine.ctx, scene.handle, C.byref(push), C.byref(stc), C.byref(frm), None))
    engine.synchronize()
    out = tuple(int(getattr(gbm, n).to(torch.int64).sum().item()) for n in ("steps_primary", "steps_total", "rays_total"))
    del gbm
    return out


def single_frame_launch(vrt, engine, renderer, pushes, W, H, S_frames, M_frames=None, reps=3):
    """K1 with ONE frame per launch: every launch alone on the device (synchronised before the next), timed by the
    library's HIP events around the kernel; mean over the step's poses, best of `reps` sweeps."""
    stage = renderer._geometryStage
    launch = stage.prepare()
    engine.set_timing(True)
    best = None
    for _ in range(reps):
        tot = 0.0
        for p in pushes:
            launch(p)
            engine.synchronize()
            tot += engine.last_timings()["primary_ms"]
        best = tot if best is None or tot < best else best
    engine.set_timing(False)
    ms = best / len(pushes)
    b_ref = (sum(S_frames) / len(S_frames)) + W * H * B_OUT
    b_alg = ((sum(M_frames) / len(M_frames)) if M_frames else 0) + W * H * B_OUT
    return {"kernel_ms": round(ms, 5), "frac": round(b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if M_frames else None,
            "frac_reference_steps": round(b_ref / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "Mrays_per_s": round(W * H / (ms * 1e-3) / 1e6, 1),
            "sample": f"{len(pushes)} poses of the step, one vrt_render_geometry call each, device idle between launches"}


def pmc_figures(tag, kernel_ms):
    """HBM traffic and the vector unit's share of one launch from the committed PMC summary of this workload (profiles/r04_<tag>_pmc.json;
    quoted only while the kernel sources are the ones it was collected on)."""
    path = os.path.join(ROOT, "profiles", f"r04_{tag}_pmc.json")
    try:
        pj = json.load(open(path))
        if pj.get("csrc_sha16") != csrc_sha16():
            return {"pmc_source": f"dropped: {os.path.basename(path)} was collected on other kernel sources"}
        c = pj["counters"]
        hb = pj["hbm_bytes_per_launch"]
        out = {"pmc_source": os.path.relpath(path, ROOT), "hbm_bytes": int(hb["total_guide_rule"]), "hbm_bytes_raw_fetch": int(hb["total_raw_fetch"]),
               "hbm_frac": round(hb["total_guide_rule"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        if "SQ_INSTS_VALU" in c:
            out["valu_insts"] = int(c["SQ_INSTS_VALU"]["mean_per_launch"])
            out["valu_issue_ms"] = round(c["SQ_INSTS_VALU"]["mean_per_launch"] * VALU_CYCLES / N_SIMD / CLOCK_HZ * 1e3, 5)
            out["valu_issue_frac"] = round(out["valu_issue_ms"] / kernel_ms, 4)
        return out
    except Exception as e:
        return {"pmc_source": f"none: {e}"}


def extra_config(vrt, torch, engine, scene, push, W, H, name, ao, shadows, bounces, iters, reps=9, max_steps=512, kernel="k_primary<DF, megakernel>",
                 batch_pushes=None, pmc=None):
    """One frame of a secondary-ray configuration: kernel times from the library's HIP events (median of `reps` isolated
    frames), ray / step counts from a second render with the count planes attached."""
    st = vrt.VoxelRenderSettings(targetResolution=(W, H))
    st.fsrSetttings.enable = False
    st.occlusionSettings.numSamples = ao
    st.traceSettings.shadows = bool(shadows)
    st.traceSettings.maxReflections = bounces
    st.traceSettings.maxRaySteps = max_steps
    st.denoiserSettings.enable = iters > 0
    st.denoiserSettings.iterations = max(iters, 1)
    geo = vrt.GeometryStage(engine, st, scene)
    den = vrt.DenoiserStage(engine, st)
    engine.set_timing(True)
    tg, td = [], []
    for _ in range(reps + 2):
        gb = geo.record(push)
        if iters > 0:
            den.record(gb.color, gb.normal, gb.position)
        engine.synchronize()
        t = engine.last_timings()
        tg.append(t["geometry_ms"]); td.append(t["denoise_ms"])
    tg, td = tg[2:], td[2:]
    dbg = vrt.GeometryStage(engine, st, scene, debug_planes=True).record(push)
    engine.synchronize()
    S = int(dbg.steps_total.to(torch.int64).sum().item())
    rays = int(dbg.rays_total.to(torch.int64).sum().item())
    engine.set_timing(False)
    g_ms = median(tg)
    _, S_marched, _ = marched_counts(vrt, torch, engine, scene, st, push, W, H)
    _, Q_req, _ = marched_counts(vrt, torch, engine, scene, st, push, W, H, lookups=True)
    b_ref = S + W * H * B_OUT                                  # the reference loop's iterations (SURVEY 8(d)'s count)
    b_geo = S_marched + W * H * B_OUT                          # the iterations the product march takes
    b_req = Q_req + W * H * B_OUT                              # the bytes its lanes ask for + the G-buffer
    out = {"name": name, "ao_samples": ao, "shadows": int(bool(shadows)), "max_bounces": bounces, "denoiser_passes": iters,
           "resolution": [W, H], "max_steps": max_steps, "geometry_kernel": kernel, "geometry_ms": round(g_ms, 5),
           "rays_total": rays, "dda_steps_total": S, "dda_steps_marched": S_marched, "Mrays_total_per_s": round(rays / (g_ms * 1e-3) / 1e6, 1),
           "geometry_algorithmic_bytes": b_geo, "geometry_frac": round(b_geo / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
           "geometry_frac_reference_steps": round(b_ref / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
           "requested_bytes": b_req, "geometry_frac_requested": round(b_req / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
           "write_floor_ms": round(W * H * B_OUT / (STORE_GBS * 1e9) * 1e3, 5)}
    if pmc:
        out.update(pmc_figures(pmc, g_ms))
    if batch_pushes:
        # the same settings with several frames per launch (consecutive poses): a hit wave's chain of secondary traces then has
        # other frames' waves to hide behind, as the headline's primary rays have
        nb = len(batch_pushes)
        launch = vrt.GeometryStage(engine, st, scene).prepare_batch(nb)
        for _ in range(3):
            launch(batch_pushes)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            launch(batch_pushes)
        e1.record(); torch.cuda.synchronize()
        out["geometry_ms_per_frame_batched"] = round(e0.elapsed_time(e1) / 10 / nb, 5)
        out["batched_frames_per_launch"] = nb
        del launch
    if iters > 0:
        d_ms = median(td)
        b_den = W * H * (K3_BYTES_PASS0 + (iters - 1) * K3_BYTES_PASS)
        out.update({"denoise_ms": round(d_ms, 5), "denoise_algorithmic_bytes": b_den,
                    "denoise_frac": round(b_den / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "frame_ms": round(g_ms + d_ms, 5)})
    return out


def frames_in_flight(vrt, scene, push, W, H, configs, slots=(1, 2, 3), n=240):
    """The reference's call pattern with its own MAX_FRAMES_IN_FLIGHT (source/engine/engine.hpp:19): one frame per
    vrt_render_geometry call, k contexts (a stream and a G-buffer each, one shared scene) taking the calls in turn -- a frame's
    tail overlaps the next frame's ramp without any batching API.  Wall-clock us per frame over n calls, host enqueue included."""
    import ctypes as C
    engines = [vrt.Engine(0, use_torch_stream=False) for _ in range(max(slots))]
    for e in engines:
        e.set_timing(False)
    gbs = [vrt.GeometryBuffer(e, W, H) for e in engines]
    for e in engines:
        e.synchronize()
    frs = [g.to_c() for g in gbs]
    out = {}
    for name, st in configs:
        stc = st.to_c()
        row = {}
        for k in slots:
            def go(m):
                for j in range(m):
                    vrt._capi.check(vrt.lib().vrt_render_geometry(engines[j % k].ctx, scene.handle, C.byref(push), C.byref(stc), C.byref(frs[j % k]), None))
                for e in engines:
                    e.synchronize()
            go(12)
            t0 = time.perf_counter()
            go(n)
            row[str(k)] = round((time.perf_counter() - t0) / n * 1e6, 2)
        out[name] = row
    for e in engines:
        e.destroy()
    return {"us_per_frame": out, "what": "one frame per vrt_render_geometry call, k contexts (frames in flight) taking the calls in turn; wall clock over "
                                         f"{n} calls, host enqueue included; the reference's Engine keeps MAX_FRAMES_IN_FLIGHT = 2"}


def primary_4k(vrt, torch, engine, scene, NV, pos0, yaw, pitch, W=3840, H=2160, nb=16):
    """The headline's workload at 3840x2160: K1 batched (nb frames per launch) and one frame per launch."""
    import numpy as np
    st = vrt.VoxelRenderSettings.primary_only((W, H))
    pushes = [vrt.make_push(vrt.CameraController(position=(pos0[0] + 1.5 * t, pos0[1] + 0.5 * t, pos0[2] + 2.0 * t), yaw=yaw, pitch=pitch), (NV, NV, NV), (W, H))
              for t in (8.0 * f / nb for f in range(nb))]
    stage = vrt.GeometryStage(engine, st, scene)
    launch = stage.prepare_batch(nb)
    for _ in range(3):
        launch(pushes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        launch(pushes)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10 / nb
    del launch
    single = stage.prepare()
    engine.set_timing(True)
    t1 = []
    for p in pushes:
        single(p); engine.synchronize()
        t1.append(engine.last_timings()["primary_ms"])
    engine.set_timing(False)
    dbg = vrt.GeometryStage(engine, vrt.VoxelRenderSettings.primary_only((W, H), vrt.TRAVERSAL_BITMASK), scene, debug_planes=True)
    S, M = [], []
    for p in pushes[::4]:
        gb = dbg.record(p); engine.synchronize()
        S.append(int(gb.steps_primary.to(torch.int64).sum().item()))
        M.append(marched_counts(vrt, torch, engine, scene, st, p, W, H)[0])
    del dbg, gb
    px = W * H * B_OUT
    b_alg, b_ref = sum(M) / len(M) + px, sum(S) / len(S) + px
    ms1 = sum(t1) / len(t1)
    return {"name": f"configs[1] at 4K: primary rays only, {W}x{H}, the headline's scene and camera path", "resolution": [W, H],
            "geometry_kernel": "k_tile_tags + k_primary<DF, primary only>", "batched_frames_per_launch": nb,
            "geometry_ms_per_frame_batched": round(ms, 5), "Mrays_per_s_batched": round(W * H / (ms * 1e-3) / 1e6, 1),
            "geometry_frac": round(b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "geometry_frac_reference_steps": round(b_ref / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "geometry_ms": round(ms1, 5), "Mrays_per_s_single_frame": round(W * H / (ms1 * 1e-3) / 1e6, 1),
            "single_frame_frac": round(b_alg / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "dda_steps_per_frame": int(sum(S) / len(S)), "dda_steps_marched_per_frame": int(sum(M) / len(M)),
            "write_floor_ms": round(px / (STORE_GBS * 1e9) * 1e3, 5)}


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"bench.py: WORLD_SIZE={world} does not match --gpus {args.gpus}; refusing to report a {world}-GPU run as {args.gpus}")
        sys.exit(2)
    if os.environ.get("VRT_BENCH_LAUNCH_ONLY") == "1":
        return launcher_selftest(world, rank)

    import numpy as np
    import torch
    import torch.distributed as dist
    import voxel_raytracing_amd as vrt

    # rehearsal of the N > 1 control flow on a box with ONE GPU: VRT_BENCH_BACKEND=gloo VRT_BENCH_DEVICE=0 makes every rank
    # use cuda:0 and sends the gather through host memory (numbers from such a run are not benchmark results)
    backend = os.environ.get("VRT_BENCH_BACKEND", "nccl")
    if "VRT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["VRT_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    engine = vrt.Engine(local_rank)
    W, H, NV = args.width, args.height, args.volume

    # ---- synthetic scene, resident in HBM before the timed region ------------------------------------
    vol = vrt.synthetic.treehouse(NV, seed=2)
    pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
    sky, noise = vrt.synthetic.sky_gradient(512, 256), vrt.synthetic.blue_noise_standin(512)
    scene = vrt.VoxelScene.from_dense(engine, vol, pal, sky=sky, noise=noise)
    trav = getattr(vrt, "TRAVERSAL_" + args.traversal)
    st = vrt.VoxelRenderSettings.primary_only((W, H), trav)
    renderer = vrt.VoxelRenderer(engine, st, scene)
    pos0, yaw, pitch = vrt.synthetic.default_camera_for(NV, NV, NV)
    per_gpu = args.frames_per_gpu if args.frames_per_gpu else (128 if world == 1 else 32)
    F = world * max(1, per_gpu)                               # frames of a batch
    # the same dolly move at every N and batch size (8 units of travel per step), sampled as finely as the batch has frames:
    # the rays per frame and their cost do not drift with N
    poses = [np.array([pos0[0] + 1.5 * t, pos0[1] + 0.5 * t, pos0[2] + 2.0 * t], np.float32) for t in (8.0 * f / F for f in range(F))]
    pushes = []
    for f in range(F):                                        # camera + push constants per pose, marshalled once
        renderer.camera.position = poses[f]
        pushes.append(renderer.push_constants())
    assemble_on = os.environ.get("VRT_ASSEMBLE", "owners")     # where finished frames end up: "owners" | "root"
    sb = vrt.distributed.ShardedBatch(renderer._geometryStage, F, rank, world, host_staged=(backend != "nccl"),
                                      assemble_on=assemble_on, direct="only", in_place=True)
    launches_per_step = (F + 255) // 256                      # K1 launches per rank and step (VRT_MAX_TABLE frames each)

    overlap = os.environ.get("VRT_SYNC_GATHER", "0") != "1"

    def step():
        # K1 over this rank's strips of the F frames, then pack + ONE RCCL collective (+ assembly at the receivers); the
        # collective of a step runs while the next step is traced (VRT_SYNC_GATHER=1: strictly one after the other)
        sb.step(pushes, overlap)

    def barrier():
        sb.finish()                                           # the collective still in flight, and the assembly of its frames
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    engine.set_timing(False)                                  # no per-call event packets inside the library
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    # HIP events on the launch stream (the context runs on torch's current stream), one every EV_EVERY steps so
    # that the event packets themselves do not pace the queue; with N = 1 and the fused primary-only kernel a
    # step is exactly launches_per_step k_primary launches and nothing else, so (event span) / (launches in the
    # span) is the kernel's average duration.
    EV_EVERY = 10
    marks = [torch.cuda.Event(enable_timing=True)]
    marks[0].record()
    for k in range(args.steps):
        step()
        if (k + 1) % EV_EVERY == 0 or k + 1 == args.steps:
            e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=engine.torch_device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    rays_per_step = F * W * H
    value = rays_per_step * args.steps / dt / 1e6
    kern_ms = float(marks[0].elapsed_time(marks[-1])) / (args.steps * launches_per_step)   # per K1 launch

    # N > 1: the frames assembled from everybody's strips must be the frames one GPU renders alone (checked on every rank
    # that holds finished frames, outside the timed region)
    assembled_ok = None
    if world > 1:
        mine = list(sb.owned_frames())
        good = 1
        if mine:
            chk = vrt.GeometryStage(engine, st, scene)
            for j in (0, len(mine) - 1):
                alone = chk.record(pushes[mine[j]]).color
                engine.synchronize()
                good = good and int(bool((sb.finals[j] == alone).all().item()))
        t = torch.tensor([good], dtype=torch.int32, device=engine.torch_device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        assembled_ok = bool(t.item())

    out = None
    if rank == 0:
        # ---- algorithmic bytes of one K1 launch (1 B per DDA iteration + W*H*B_out) ----
        # S_frames: the iterations of the REFERENCE's loop (every ray walks to a hit, the wall or the end of its budget);
        # M_frames: the iterations the product march really takes (rays end at open cells, untagged blocks are not traced) --
        # every frame of the launch, not one extrapolated
        engine.set_timing(True)
        st_dbg = vrt.VoxelRenderSettings.primary_only((W, H), vrt.TRAVERSAL_BITMASK)
        stage = vrt.GeometryStage(engine, st_dbg, scene, debug_planes=True)
        S_frames, M_frames, Q_frames, hit_frac = [], [], [], []
        frames_per_launch = min(F, 256)
        for f in range(frames_per_launch):                    # the frames of the first launch of a step
            gb = stage.record(pushes[f])
            engine.synchronize()
            S_frames.append(int(gb.steps_primary.to(torch.int64).sum().item()))
            hit_frac.append(float((gb.hit_id != 0).float().mean().item()))
            M_frames.append(marched_counts(vrt, torch, engine, scene, st, pushes[f], W, H)[0])
            Q_frames.append(marched_counts(vrt, torch, engine, scene, st, pushes[f], W, H, lookups=True)[0])
        del stage, gb
        scene.trim()                                           # the count planes' second set of clearance fields: not carried through the legs below
        S_frame = S_frames[0]
        # frame 0's hit ids as the TIMED configuration produces them (AUTO: the hand-written loop, open cells, tile tags, the sky
        # fast path), for the CPU leg's comparison
        import ctypes as C
        gbh = vrt.GeometryBuffer(engine, W, H, ("hit_id", "color8"))
        stc, frm = st.to_c(), gbh.to_c()
        vrt._capi.check(vrt.lib().vrt_render_geometry(engine.ctx, scene.handle, C.byref(pushes[0]), C.byref(stc), C.byref(frm), None))
        engine.synchronize()
        hit0 = gbh.hit_id.cpu().numpy()
        color0 = gbh.color8.cpu().numpy()
        del gbh
        # a launch covers this rank's strips (1 / world of the rows) of frames_per_launch frames
        px_bytes = frames_per_launch * W * H * B_OUT / world
        b_alg = sum(M_frames) / world + px_bytes
        b_ref = sum(S_frames) / world + px_bytes
        b_req = sum(Q_frames) / world + px_bytes
        achieved = b_alg / (kern_ms * 1e-3) / 1e9
        tm = engine.last_timings()
        # HBM bytes per K1 launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over
        # this same command; tools/pmc_summary.py -> profiles/*_k_primary_pmc.json).  Counters cannot be read from inside
        # the process, so the committed summary is quoted -- only for the configuration AND the kernel sources it was
        # collected on (the summary carries the digest of csrc/; another digest means the number is stale and is dropped).
        traffic, traffic_src, valu = None, None, None
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_k_primary_pmc.json")))
        if pm and world == 1 and (W, H, NV) == (1920, 1080, 256) and args.traversal in ("AUTO", "DF"):
            try:
                pj = json.load(open(pm[-1]))
                if int(pj.get("frames_per_launch", 1)) != frames_per_launch:
                    raise ValueError("PMC summary was collected at another batch size")
                if pj.get("csrc_sha16") != csrc_sha16():
                    raise ValueError(f"PMC summary {os.path.basename(pm[-1])} was collected on other kernel sources "
                                     f"({pj.get('csrc_sha16')} != {csrc_sha16()})")
                traffic = int(pj["hbm_bytes_per_launch"]["total_guide_rule"])
                traffic_src = os.path.relpath(pm[-1], ROOT)
                valu = pj["counters"].get("SQ_INSTS_VALU", {}).get("mean_per_launch")
            except Exception as e:
                traffic, traffic_src = None, f"dropped: {e}"
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                    "frac_is": "SURVEY 8(d)'s accounting over the work PERFORMED: 1 B per DDA iteration the product march takes (every frame of the launch, counted "
                               "by the twins of the loops that are timed) + 37 B per pixel, / kernel time / peak.  An iteration of a clearance run is an addition in a "
                               "register, not a fetch: this measures arithmetic in byte units; frac_requested is the memory statement",
                    "requested_bytes_per_launch": int(b_req), "lane_lookups_per_launch": int(sum(Q_frames)),
                    "frac_requested": round(b_req / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "frac_requested_is": "bytes the lanes ASK memory for (one per clearance look-up of a live lane, one per voxel id read) + 37 B per pixel stored, "
                                         "/ kernel time / peak",
                    "bound_in_practice": ({"what": "vector-instruction issue", "valu_insts_per_launch": int(valu),
                                           "valu_issue_ms": round(valu * VALU_CYCLES / N_SIMD / CLOCK_HZ * 1e3, 5),
                                           "valu_issue_frac_of_kernel": round(valu * VALU_CYCLES / N_SIMD / CLOCK_HZ * 1e3 / kern_ms, 4),
                                           "model": f"SQ_INSTS_VALU x {VALU_CYCLES} cycles / {N_SIMD} SIMDs / {CLOCK_HZ / 1e9} GHz (same PMC file)"} if valu else None),
                    "hbm_frac": round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
                    "traffic_GBps": round(traffic / (kern_ms * 1e-3) / 1e9, 2) if traffic else None,
                    "write_floor_ms": round(px_bytes / (STORE_GBS * 1e9) * 1e3, 5),
                    "frac_reference_steps": round(b_ref / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "frac_reference_steps_is": "the same formula over the iterations of the REFERENCE's loop (SURVEY 8(d)'s count): the product proves most "
                                               "of them unnecessary (open cells, tile tags: DESIGN.md 5) and does not take them, so this figure passes 1 -- not a bound",
                    "kernel": "k_tile_tags + k_primary (one launch of each per step)", "kernel_ms": round(kern_ms, 5), "csrc_sha16": csrc_sha16(),
                    "frames_per_launch": frames_per_launch, "algorithmic_bytes_per_launch": int(b_alg),
                    "dda_steps_marched_per_launch": int(sum(M_frames)), "dda_steps_reference_per_launch": int(sum(S_frames)),
                    "dda_steps_per_frame": S_frame, "steps_per_ray": round(S_frame / (W * H), 2),
                    "dda_steps_marched_frame0": M_frames[0]}
        extra = None
        if world == 1 and not args.no_extra_configs:
            roofline["single_frame_launch"] = single_frame_launch(vrt, engine, renderer, pushes[:frames_per_launch], W, H, S_frames, M_frames)
            extra = [extra_config(vrt, torch, engine, scene, pushes[0], W, H, *c, batch_pushes=pushes[:16]) for c in (
                ("configs[2]: primary + shadow ray, 1 denoiser pass", 0, True, 0, 1),
                ("configs[2]: primary + shadow ray, 2 denoiser passes", 0, True, 0, 2),
                ("reference defaults: AO 4 x 64 steps, shadow ray, <= 5 bounces, 2 denoiser passes", 4, True, 5, 2))]
            for e, tag in zip(extra, ("config3", "config3", "defaults")):
                e.update(pmc_figures(tag, e["geometry_ms"]))
            # the same three workloads one frame per call with 1, 2 and 3 frames in flight (the reference's own pattern)
            st_c3 = vrt.VoxelRenderSettings.primary_only((W, H)); st_c3.traceSettings.shadows = True
            st_df = vrt.VoxelRenderSettings(targetResolution=(W, H)); st_df.fsrSetttings.enable = False
            roofline["single_frame_launch"]["frames_in_flight"] = frames_in_flight(
                vrt, scene, pushes[0], W, H, (("primary", st), ("config3_geometry", st_c3), ("reference_defaults_geometry", st_df)))
            # north_star's "1080p and 4K": the headline's scene and camera path at 3840x2160, primary rays only -- batched like the
            # headline (16 frames per launch: 77 MB of planes x 4 each) and one frame per launch
            extra.append(primary_4k(vrt, torch, engine, scene, NV, pos0, yaw, pitch))
            # BASELINE configs[3]: Mandelbulb 512^3 at 3840x2160, 2 bounces, AO 4 + shadow ray (one GPU's view of the frame the
            # 8-GPU run cuts into strips)
            t_gen = time.perf_counter()
            volm = vrt.synthetic.mandelbulb(512)
            scm = vrt.VoxelScene.from_dense(engine, volm, pal, sky=sky, noise=noise)
            camm = vrt.CameraController(position=(512 * 0.5 + 0.3, 512 * 0.5 + 0.2, -0.45 * 512))
            pushm = vrt.make_push(camm, (512, 512, 512), (3840, 2160), frame=5)
            batchm = [vrt.make_push(vrt.CameraController(position=(512 * 0.5 + 0.3 + 1.5 * k, 512 * 0.5 + 0.2 + 0.5 * k, -0.45 * 512 + 2.0 * k)),
                                    (512, 512, 512), (3840, 2160), frame=5 + k) for k in range(4)]
            em = extra_config(vrt, torch, engine, scm, pushm, 3840, 2160,
                              "configs[3]: synthetic:mandelbulb(N=512), 3840x2160, 2 bounces, AO 4 x 64 steps, shadow ray", 4, True, 2, 0,
                              reps=5, kernel="k_primary<DF, megakernel>", batch_pushes=batchm, pmc="mandelbulb")
            em["scene_device_bytes"] = scm.memory_bytes()
            em["scene_build_s"] = round(time.perf_counter() - t_gen, 2)
            extra.append(em)
            scm.destroy()
            del volm
            # BASELINE configs[4]: the 2048^3 brick scene at 3840x2160 (one GPU's view of it; the 8-GPU split is the same frame
            # cut into strips) -- the one configuration whose volume does not sit in the caches
            t_gen = time.perf_counter()
            grid, pool = vrt.synthetic.sparse_brick_scene(2048, 0.015, seed=5)
            sc5 = vrt.VoxelScene.from_bricks(engine, grid, pool, pal, sky=sky, noise=noise)
            pos5, yaw5, pitch5 = vrt.synthetic.default_camera_for(2048, 2048, 2048)
            cam5 = vrt.CameraController(position=(pos5[0] + 0.3, pos5[1] + 0.2, pos5[2]), yaw=yaw5, pitch=pitch5)
            push5 = vrt.make_push(cam5, (2048, 2048, 2048), (3840, 2160), frame=17)
            e5 = extra_config(vrt, torch, engine, sc5, push5, 3840, 2160,
                              "configs[4]: synthetic:sparse2048(seed=5) brick scene (1.5 % of 8^3 bricks), 3840x2160, max_steps 6144, 4 bounces, AO 4", 4, True, 4, 0,
                              reps=5, max_steps=6144, kernel="k_primary<BRICK, megakernel>", pmc="brick")
            e5["scene_device_bytes"] = sc5.memory_bytes()
            e5["scene_bricks"] = int(pool.shape[0])
            e5["scene_build_s"] = round(time.perf_counter() - t_gen, 2)
            extra.append(e5)
            sc5.destroy()
            del grid, pool
        cpu = None
        if not args.no_cpu_baseline and world == 1:           # the CPU leg is reported at N = 1 only
            from oracle import oracle                          # checker / CPU baseline only
            osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
            # every hardware thread this process may run on; the counts are in the line.  (A GPU box of the pool gives a
            # one-GPU job a share of the host -- os.sched_getaffinity -- not all of os.cpu_count(); above 128 threads the oracle gains nothing.)
            try:
                usable = len(os.sched_getaffinity(0))
            except Exception:
                usable = os.cpu_count() or 1
            quota = None                                       # (a cgroup CPU quota below the affinity mask: more threads than that only contend)
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if q != "max":
                    quota = max(1, int(-(-int(q) // int(per))))
            except Exception:
                pass
            ncores = max(1, min(usable, quota or usable, 256))
            # a bounded sample of the same workload: every (F / 8)-th frame of the step, i.e. 8 full frames spread over the
            # camera path (about 10 CPU-seconds at 1080p: 0.6 s of wall time on 16 threads)
            sample = list(range(0, F, max(1, F // 8)))[:8]
            c0 = time.perf_counter()
            same, same_color = True, True
            for f in sample:
                exp = oracle.render(osn, pushes[f], oracle.params_from(st.to_c()), planes=["hit_id", "steps_primary", "color8"], nthreads=ncores)
                if f == 0:                                     # the oracle's frame against what the TIMED kernel configuration renders
                    same = bool((exp["hit_id"] == hit0).all()) and int(exp["steps_primary"].sum()) == S_frame
                    same_color = bool((exp["color8"] == color0).all())
            cdt = time.perf_counter() - c0
            cpu = {"value": round(len(sample) * W * H / cdt / 1e6, 3), "unit": "Mrays/s", "cores": ncores, "hardware_threads": os.cpu_count(), "usable_threads": usable, "cgroup_cpu_quota": quota, "kind": "port",
                   "sample": f"{len(sample)} full {W}x{H} frames of the same workload (every {max(1, F // 8)}th pose of the step), scalar C oracle, "
                             f"rows interleaved over {ncores} threads, {cdt:.2f} s",
                   "hit_ids_match_gpu": same, "color8_matches_gpu": same_color,
                   "compared_with": "frame 0 as the timed configuration renders it (AUTO: hand-written loop, open cells, tile tags, sky fast path)"}
            # one thread on a band of the same frame (every 8th row group would bias towards sky; a centred band does not)
            r0, r1 = H // 2 - 60, H // 2 + 60
            c1 = time.perf_counter()
            oracle.render_band(osn, pushes[0], oracle.params_from(st.to_c()), r0, r1, planes=["hit_id"], nthreads=1)
            cpu["single_thread"] = {"value": round(W * (r1 - r0) / (time.perf_counter() - c1) / 1e6, 3), "unit": "Mrays/s",
                                    "sample": f"rows {r0}..{r1 - 1} of the same frame, one thread"}
        out = {"metric": "Mrays/sec at 1080p treehouse.vox; achieved HBM GB/s vs MI355X peak",
               "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"synthetic:treehouse(seed=2) {NV}^3 stand-in for treehouse.vox, {W}x{H}, primary rays only "
                                      f"(BASELINE configs[1]); {F} frame(s)/step ({F // world} per GPU, consecutive poses, one K1 launch per step), "
                                      f"{sb.strip_rows}-row strips over {world} GPU(s)"
                                      + ((", one RCCL all-to-all/step (frame block b assembled on rank b), strip assignment rotated per block"
                                          if sb.owners else ", one RCCL gather/step to rank 0") if world > 1 else ""),
                          "traversal": args.traversal, "frames_per_step": F, "assembled_frames_match_single_gpu": assembled_ok, "hit_fraction": round(hit_frac[0], 4), "bytes_out_per_px": B_OUT,
                          "device": engine.device_info()[0]},
               "roofline": roofline, "cpu_baseline": cpu, "extra_configs": extra}
        log(f"timings of last call: {tm}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
