"""Randomised parity sweep: HIP path (C-ABI) vs the CPU oracle on random scenes, cameras and settings.
   python tests/fuzz_parity.py [cases] [seed]      -- prints one line per failing case and a summary; exit code 1 on mismatch.
Test infrastructure (uses oracle/); not part of the product."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import voxel_raytracing_amd as vrt
from oracle import oracle
from helpers import compare_planes

GB = ["color8", "depth", "motion", "mask8", "position", "normal8"]
DBG = ["color_f", "hit_id", "hit_voxel", "hit_mask", "steps_primary", "steps_total", "rays_total"]


def run(cases, seed, eng, verbose=True):
  """Returns the number of mismatching cases."""
  rng = np.random.default_rng(seed)
  bad = 0
  t0 = time.time()
  for case in range(cases):
      kind = rng.integers(0, 5)
      dims = [int(rng.integers(5, 70)) for _ in range(3)] if kind == 4 else None
      if kind == 0:   vol = vrt.synthetic.floating_cubes(int(rng.integers(16, 72)), seed=int(rng.integers(1, 1 << 30)), count=int(rng.integers(1, 200)))
      elif kind == 1: vol = vrt.synthetic.sparse_bricks(int(rng.choice([32, 48, 64])), int(rng.choice([2, 4, 8])), float(rng.uniform(0.005, 0.3)), seed=int(rng.integers(1, 1 << 30)))
      elif kind == 2: vol = vrt.synthetic.treehouse(int(rng.choice([32, 64])), seed=int(rng.integers(1, 1 << 30)))
      elif kind == 3: vol = (rng.random((int(rng.integers(4, 40)),) * 3) < rng.uniform(0.0, 0.2)).astype(np.uint8) * rng.integers(1, 256, dtype=np.uint8)
      else:           vol = (rng.random((dims[2], dims[1], dims[0])) < rng.uniform(0.0, 0.1)).astype(np.uint8) * np.uint8(rng.integers(1, 256))   # non-cubic
      D, H, W = vol.shape
      pal = vrt.synthetic.default_palette(metallic_ids=range(int(rng.integers(1, 256)), 256), metallic_value=float(rng.choice([0.0, 0.5, 0.8, 1.0])))
      sky = vrt.synthetic.sky_gradient(int(rng.choice([1, 7, 64])), int(rng.choice([1, 5, 32])))
      noise = vrt.synthetic.blue_noise_standin(int(rng.choice([1, 16, 64])))
      sc = vrt.VoxelScene.from_dense(eng, vol, pal, sky=sky, noise=noise)
      osn = oracle.OracleScene(vol, pal, sky=sky, noise=noise)
      res = (int(rng.integers(1, 160)), int(rng.integers(1, 120)))
      st = vrt.VoxelRenderSettings(targetResolution=res)
      st.fsrSetttings.enable = False
      st.occlusionSettings.numSamples = int(rng.integers(0, 5))
      st.occlusionSettings.intensity = float(rng.choice([1.0, 0.5, 2.0]))
      st.traceSettings.shadows = bool(rng.integers(0, 2))
      st.traceSettings.maxReflections = int(rng.integers(0, 6))
      st.traceSettings.maxRaySteps = int(rng.choice([0, 1, 7, 64, 512, 2000]))
      st.traceSettings.aoSteps = int(rng.choice([1, 16, 64]))
      trav = str(rng.choice(["DF", "DF", "DF", "DENSE", "BITMASK", "JUMP", "DFJ"]))
      st.traceSettings.traversal = getattr(vrt, "TRAVERSAL_" + trav)
      st.traceSettings.splitKernels = bool(rng.integers(0, 4) == 0)
      ld = rng.normal(size=3); st.lightSettings.direction = tuple((ld / np.linalg.norm(ld)).astype(np.float32).tolist())
      mode = rng.integers(0, 4)
      if mode == 0:   pos = (W / 2 + rng.uniform(-1, 1), H / 2 + rng.uniform(-1, 1), -rng.uniform(0.2, 2.0) * D)                 # outside, looking in
      elif mode == 1: pos = tuple(rng.uniform(0, 1, 3) * np.array([W, H, D]))                                               # inside the volume
      elif mode == 2: pos = (float(rng.integers(0, W + 1)), float(rng.integers(0, H + 1)), -float(rng.integers(0, 40)))    # lattice positions: exact ties
      else:           pos = tuple(rng.uniform(-2, 3, 3) * np.array([W, H, D]))                                              # anywhere, often missing the box
      yaw = float(rng.choice([90.0, 0.0, 45.0, rng.uniform(0, 360)])); pitch = float(rng.choice([0.0, 45.0, -30.0, rng.uniform(-89, 89)]))
      cam = vrt.CameraController(position=pos, yaw=yaw, pitch=pitch)
      jit = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5))) if rng.integers(0, 2) else (0.0, 0.0)
      push = vrt.make_push(cam, (W, H, D), res, frame=int(rng.integers(0, 100)), jitter=jit)
      gb = vrt.GeometryStage(eng, st, sc, debug_planes=True).record(push)
      eng.synchronize()
      exp = oracle.render(osn, push, oracle.params_from(st.to_c()), nthreads=8)
      names = GB + DBG if trav not in ("JUMP", "DFJ") else GB + ["color_f", "hit_id", "hit_voxel", "hit_mask", "rays_total"]
      b = compare_planes(gb.numpy(), exp, names)
      # ... and the launch a caller of the product makes: the reference's six targets and nothing else (the sky-texel fast path, no
      # diagnostic march)
      plain = vrt.GeometryStage(eng, st, sc).record(push)
      eng.synchronize()
      b += compare_planes(plain.numpy(), exp, GB)
      # ... and the same content as a brick scene (8^3 bricks; volumes whose sides are multiples of 8): the brick march and its
      # threshold runs (AUTO is the one traversal a brick scene renders with; the split form has no brick kernel of its own)
      if W % 8 == 0 and H % 8 == 0 and D % 8 == 0 and not st.traceSettings.splitKernels:
          grid, pool = vrt.synthetic.bricks_from_dense(vol)
          sb = vrt.VoxelScene.from_bricks(eng, grid, pool, pal, sky=sky, noise=noise)
          trav0 = st.traceSettings.traversal
          st.traceSettings.traversal = vrt.TRAVERSAL_AUTO
          expb = exp if trav in ("DF", "DENSE", "BITMASK") else oracle.render(osn, push, oracle.params_from(st.to_c()), planes=GB, nthreads=8)
          pb = vrt.GeometryStage(eng, st, sb).record(push)
          eng.synchronize()
          b += compare_planes(pb.numpy(), expb, GB)
          st.traceSettings.traversal = trav0
          sb.destroy()
      it = int(rng.integers(0, 4)); sw = float(rng.choice([2.0, 1.0, 1.5, 0.0, 3.0])); dmode = int(rng.integers(0, 2))
      st.denoiserSettings.iterations = it; st.denoiserSettings.stepWidth = sw; st.denoiserSettings.mode = dmode
      den = vrt.DenoiserStage(eng, st).record(gb.color, gb.normal, gb.position)
      eng.synchronize()
      e = oracle.denoise(exp["color8"], exp["normal8"], exp["position"], iterations=it, step_width0=sw, mode=dmode)
      dbad = int((den.cpu().numpy() != e).sum())
      if b or dbad:
          bad += 1
          print(f"MISMATCH case {case} seed {seed}: kind {kind} vol {W}x{H}x{D} res {res} trav {trav} split {st.traceSettings.splitKernels} ao {st.occlusionSettings.numSamples} "
                f"sh {st.traceSettings.shadows} b {st.traceSettings.maxReflections} steps {st.traceSettings.maxRaySteps} cam {pos} {yaw} {pitch}: {b[:2]} denoise diffs {dbad}", flush=True)
      sc.destroy()
      if verbose and (case + 1) % 50 == 0:
          print(f"{case + 1} cases, {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
  return bad


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = run(cases, seed, vrt.Engine(0))
    print(f"done: {cases} cases, {bad} mismatching")
    sys.exit(1 if bad else 0)
