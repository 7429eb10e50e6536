#!/usr/bin/env python3
"""bench.py -- BASELINE metric: Mrays/s at 1080p on the treehouse scene (primary rays), plus the achieved
algorithmic GB/s of the primary-ray DDA kernel against the MI355X HBM peak.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: a batch of N x F frames (N = number of GPUs,
F = --frames-per-gpu, default 32; consecutive camera poses of the same 8-unit dolly move at every N and F), every frame cut into 16-row screen strips
that are dealt round-robin to the N ranks.  A rank traces its strips of ALL frames of the batch with ONE K1 launch
(vrt_render_geometry_batch / _slots: the next frame's tiles are dispatched while the previous frame drains), packs
them, and ONE RCCL collective per step moves the strips to where the frames are assembled: frame block b (F frames) is
gathered to rank b, the N gathers issued as a single all-to-all so that every GPU receives over all of its xGMI links
(VRT_ASSEMBLE=root: everything to rank 0 with one dist.gather instead -- bound by rank 0's inbound links).  The strip
assignment is rotated per block, so every rank traces the same number of rows per step although 1080 rows are 67.5
strips.  The collective of a step runs while the next step is traced.  Per-GPU work per step is therefore F frames'
worth of rays at every N ("weak" scaling); value = total primary rays of all ranks / wall time.  At N = 1 there is no
collective and no copy.

Workload = BASELINE.json configs[1]: treehouse stand-in (synthetic:treehouse(seed=2), 256^3 -- the real
treehouse.vox is a git-LFS pointer in the reference checkout), 1920x1080, primary rays only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured-achievable
B_OUT = 37                     # bytes stored per pixel: the reference's 6-target G-buffer (geometry_stage.cpp:22-33)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--volume", type=int, default=256)
    ap.add_argument("--traversal", default="AUTO", choices=["AUTO", "DENSE", "BITMASK", "JUMP", "DF", "DFJ"])
    ap.add_argument("--frames-per-gpu", type=int, default=32, help="frames of the batch per GPU and step (<= 256 / GPUs for one launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import voxel_raytracing_amd as vrt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 control flow on a box with ONE GPU: VRT_BENCH_BACKEND=gloo VRT_BENCH_DEVICE=0 makes every rank
    # use cuda:0 and sends the gather through host memory (numbers from such a run are not benchmark results)
    backend = os.environ.get("VRT_BENCH_BACKEND", "nccl")
    if "VRT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["VRT_BENCH_DEVICE"])
    if world != args.gpus:
        log(f"note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
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
    F = world * max(1, args.frames_per_gpu)                   # frames of a batch
    # the same dolly move at every N and batch size (8 units of travel per step), sampled as finely as the batch has frames:
    # the rays per frame and their cost do not drift with N
    poses = [np.array([pos0[0] + 1.5 * t, pos0[1] + 0.5 * t, pos0[2] + 2.0 * t], np.float32) for t in (8.0 * f / F for f in range(F))]
    pushes = []
    for f in range(F):                                        # camera + push constants per pose, marshalled once
        renderer.camera.position = poses[f]
        pushes.append(renderer.push_constants())
    assemble_on = os.environ.get("VRT_ASSEMBLE", "owners")     # where finished frames end up: "owners" | "root"
    sb = vrt.distributed.ShardedBatch(renderer._geometryStage, F, rank, world, host_staged=(backend != "nccl"),
                                      assemble_on=assemble_on, direct="only")
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
        # ---- algorithmic bytes of one K1 launch: S fetches (1 B each) + W*H*B_out ----
        engine.set_timing(True)
        st_dbg = vrt.VoxelRenderSettings.primary_only((W, H), vrt.TRAVERSAL_BITMASK)
        stage = vrt.GeometryStage(engine, st_dbg, scene, debug_planes=True)
        S_frames, hit_frac = [], []
        frames_per_launch = min(F, 256)
        for f in range(frames_per_launch):                    # the frames of the first launch of a step
            gb = stage.record(pushes[f])
            engine.synchronize()
            S_frames.append(int(gb.steps_primary.to(torch.int64).sum().item()))
            hit_frac.append(float((gb.hit_id != 0).float().mean().item()))
            if f == 0:
                hit0 = gb.hit_id.cpu().numpy()
        S_frame = S_frames[0]
        # a launch covers this rank's strips (1 / world of the rows) of frames_per_launch frames
        b_alg = (sum(S_frames) + frames_per_launch * W * H * B_OUT) / world
        achieved = b_alg / (kern_ms * 1e-3) / 1e9
        tm = engine.last_timings()
        # HBM bytes per K1 launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over
        # this same command; tools/pmc_summary.py -> profiles/*_k_primary_pmc.json).  Counters cannot be read from inside
        # the process, so the committed summary is quoted, and only for the configuration it was collected on.
        traffic, traffic_src = None, None
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_k_primary_pmc.json")))
        if pm and world == 1 and (W, H, NV) == (1920, 1080, 256) and args.traversal in ("AUTO", "DF"):
            try:
                pj = json.load(open(pm[-1]))
                if int(pj.get("frames_per_launch", 1)) != frames_per_launch:
                    raise ValueError("PMC summary was collected at another batch size")
                traffic = int(pj["hbm_bytes_per_launch"]["total_guide_rule"])
                traffic_src = os.path.relpath(pm[-1], ROOT)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "k_primary", "kernel_ms": round(kern_ms, 5),
                    "frames_per_launch": frames_per_launch, "algorithmic_bytes_per_launch": int(b_alg),
                    "dda_steps_per_frame": S_frame, "steps_per_ray": round(S_frame / (W * H), 2)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:           # the CPU leg is reported at N = 1 only
            from oracle import oracle                          # checker / CPU baseline only
            osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
            ncores = min(os.cpu_count() or 1, 16)
            # a bounded sample of the same workload: every (F / 8)-th frame of the step, i.e. 8 full frames spread over the
            # camera path (about 10 CPU-seconds at 1080p: 0.6 s of wall time on 16 threads)
            sample = list(range(0, F, max(1, F // 8)))[:8]
            c0 = time.perf_counter()
            same = True
            for f in sample:
                exp = oracle.render(osn, pushes[f], oracle.params_from(st.to_c()), planes=["hit_id", "steps_primary"], nthreads=ncores)
                if f == 0:
                    cdt0 = time.perf_counter() - c0
                    same = bool((exp["hit_id"] == hit0).all()) and int(exp["steps_primary"].sum()) == S_frame
            cdt = time.perf_counter() - c0
            cpu = {"value": round(len(sample) * W * H / cdt / 1e6, 3), "unit": "Mrays/s", "cores": ncores, "kind": "port",
                   "sample": f"{len(sample)} full {W}x{H} frames of the same workload (every {max(1, F // 8)}th pose of the step), scalar C oracle, "
                             f"rows interleaved over {ncores} threads, {cdt:.2f} s",
                   "hit_ids_match_gpu": same}
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
                                      f"{sb.strip_rows}-row strips round-robin over {world} GPU(s)"
                                      + ((", one RCCL all-to-all/step (frame block b assembled on rank b), strip assignment rotated per block"
                                          if sb.owners else ", one RCCL gather/step to rank 0") if world > 1 else ""),
                          "traversal": args.traversal, "frames_per_step": F, "assembled_frames_match_single_gpu": assembled_ok, "hit_fraction": round(hit_frac[0], 4), "bytes_out_per_px": B_OUT,
                          "device": engine.device_info()[0]},
               "roofline": roofline, "cpu_baseline": cpu}
        log(f"timings of last call: {tm}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
