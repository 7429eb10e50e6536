"""Fuzz of the verified denoiser pass (k_denoise_ver) against the oracle: random frame sizes, phi parameters over the whole
range the guard admits, step widths, modes, pass counts, and G-buffers of several kinds (rendered-like smooth colours with
noise, flat patches, hard edges, hostile random codes, non-finite positions).  Every case must be bit-identical.

    python tests/fuzz_denoise.py [cases] [seed]        (on a GPU box; tests/test_gpu_denoise.py runs a short sweep of it)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def make_gbuffer(rng, W, H, kind):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    if kind == 0:                                                     # smooth shading + noise, a sky region, face normals
        base = 128 + 100 * np.sin(xx / 17.0 + rng.uniform(0, 6)) * np.cos(yy / 23.0 + rng.uniform(0, 6))
        color = np.clip(base[..., None] + rng.normal(0, rng.uniform(0, 25), (H, W, 4)), 0, 255).astype(np.uint8)
        color[..., 3] = 0
        faces = np.array([[127, 0, 0, 0], [-127, 0, 0, 0], [0, 127, 0, 0], [0, -127, 0, 0], [0, 0, 127, 0], [0, 0, -127, 0], [90, 90, 0, 0], [73, 73, 73, 0]], np.int8)
        nrm = faces[rng.integers(0, len(faces), (H // 8 + 1, W // 8 + 1))].repeat(8, 0).repeat(8, 1)[:H, :W]
        pos = np.zeros((H, W, 4), np.float32)
        pos[..., 0] = xx * 0.07 + rng.normal(0, 0.02, (H, W)); pos[..., 1] = yy * 0.05; pos[..., 2] = 30 + 3 * np.sin(xx / 9.0)
        sky = yy < H * rng.uniform(0, 0.6)
        pos[sky] = 0.0; nrm = np.where(sky[..., None], 0, nrm).astype(np.int8)
    elif kind == 1:                                                   # hard edges and flat patches
        color = (rng.integers(0, 4, (H // 5 + 1, W // 7 + 1, 4)) * 85).astype(np.uint8).repeat(5, 0).repeat(7, 1)[:H, :W].copy()
        color[..., 3] = 0
        nrm = rng.choice(np.array([-127, 0, 127], np.int8), (H, W, 4)); nrm[..., 3] = 0
        pos = (np.round(rng.uniform(0, 16, (H, W, 4)) * 2) / 2).astype(np.float32); pos[..., 3] = 0
    else:                                                             # hostile: every channel random, odd values
        color = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        nrm = rng.integers(-128, 128, (H, W, 4)).astype(np.int8)
        pos = (rng.normal(0, 1, (H, W, 4)) * 10.0 ** rng.integers(-3, 3)).astype(np.float32)
        odd = rng.random((H, W))
        pos[odd < 0.003] = np.float32(np.nan); pos[(odd >= 0.003) & (odd < 0.006)] = np.float32(np.inf); pos[(odd >= 0.006) & (odd < 0.009)] = np.float32(-1e30)
    return np.ascontiguousarray(color), np.ascontiguousarray(nrm), np.ascontiguousarray(pos)


def run(vrt, oracle, engine, cases, seed, verbose=False):
    import torch
    rng = np.random.default_rng(seed)
    dev = engine.torch_device
    verified = redone = 0
    engine.set_option("denoise_count", 1)
    try:
        for case in range(cases):
            W, H = int(rng.integers(1, 300)), int(rng.integers(1, 200))
            kind = int(rng.integers(0, 3))
            color, nrm, pos = make_gbuffer(rng, W, H, kind)
            iterations = int(rng.integers(1, 5))
            step = float(rng.choice([0.0, 1.0, 1.0, 2.0, 2.0, 2.0, 4.0, 1.5]))
            mode = int(rng.integers(0, 2))
            phis = tuple(float(10.0 ** rng.uniform(-3.5, 3.5)) for _ in range(3)) if rng.random() < 0.7 else (20.4, 0.01, 0.1)
            st = vrt.VoxelRenderSettings(targetResolution=(W, H))
            d = st.denoiserSettings
            d.iterations, d.stepWidth, d.mode = iterations, step, mode
            d.phiColor0, d.phiNormal0, d.phiPos0 = phis
            c, n, p = (torch.from_numpy(a).to(dev) for a in (color, nrm, pos))
            stage = vrt.DenoiserStage(engine, st)
            got = stage.record(c, n, p).cpu().numpy()
            r = [stage.redone(i) for i in range(iterations)]
            verified += sum(1 for i in range(iterations) if np.isfinite(stage.guard(i)) and stage.guard(i) <= 0.02)
            redone += sum(r)
            exp = oracle.denoise(color, nrm, pos, iterations=iterations, phi_color0=phis[0], phi_normal0=phis[1], phi_pos0=phis[2], step_width0=step, mode=mode)
            bad = int((got != exp).sum())
            if verbose and case % 100 == 0:
                print(f"case {case}: {W}x{H} kind {kind} iterations {iterations} step {step} mode {mode} phis {phis} redone {r}", flush=True)
            assert bad == 0, f"case {case} (seed {seed}): {W}x{H} kind {kind} iterations {iterations} step {step} mode {mode} phis {phis}: {bad} bytes differ"
    finally:
        engine.set_option("denoise_count", 0)
    return verified, redone


if __name__ == "__main__":
    import voxel_raytracing_amd as vrt
    from oracle import oracle
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    v, r = run(vrt, oracle, vrt.Engine(0), cases, seed, verbose=True)
    print(f"{cases} cases bit-identical; {v} passes took the verified form, {r} pixels were evaluated twice")
