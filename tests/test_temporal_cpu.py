"""Host-side temporal / presentation logic (no GPU): the jitter sequence of UpscalerStage::update
(upscaler_stage.cpp:59-70) and the scripted camera (camera_controller.cpp:15-68)."""
import ctypes as C
import math

import numpy as np
import pytest


def test_jitter_phase_counts_match_fsr2_table(vrt, oracle):
    # ffx_fsr2.h documents 18 / 23 / 32 / 72 phases for the quality / balanced / performance / ultra-performance modes
    L = vrt.lib()
    for scaling, phases in [(vrt.FsrScaling.QUALITY, 18), (vrt.FsrScaling.BALANCED, 23),
                            (vrt.FsrScaling.PERFORMANCE, 32), (vrt.FsrScaling.ULTRA_PERFORMANCE, 72),
                            (vrt.FsrScaling.NONE, 8)]:
        st = vrt.VoxelRenderSettings(targetResolution=(1920, 1080))
        st.fsrSetttings.scaling = scaling
        rw = st.renderResolution()[0]
        assert L.vrt_jitter_phase_count(rw, 1920) == phases
        assert oracle.jitter(0, rw, 1920)[0] == phases
    assert L.vrt_jitter_phase_count(0, 1920) == 0


@pytest.mark.parametrize("rw,dw", [(1280, 1920), (1129, 1920), (960, 1920), (640, 1920), (1920, 1920), (333, 1000)])
def test_jitter_offsets_match_oracle(vrt, oracle, rw, dw):
    L = vrt.lib()
    phases = L.vrt_jitter_phase_count(rw, dw)
    seen = set()
    for i in range(3 * phases + 5):
        jx, jy = C.c_float(), C.c_float()
        assert L.vrt_jitter_offset(i, phases, C.byref(jx), C.byref(jy)) == 0
        ep, ex, ey = oracle.jitter(i, rw, dw)
        assert ep == phases
        assert np.float32(jx.value).view(np.uint32) == np.float32(ex).view(np.uint32)
        assert np.float32(jy.value).view(np.uint32) == np.float32(ey).view(np.uint32)
        assert -0.5 <= jx.value < 0.5 and -0.5 <= jy.value < 0.5
        seen.add((jx.value, jy.value))
    assert len(seen) == phases                        # Halton points of one cycle are distinct
    jx, jy = C.c_float(), C.c_float()
    assert L.vrt_jitter_offset(-1, phases, C.byref(jx), C.byref(jy)) != 0
    assert L.vrt_jitter_offset(0, 0, C.byref(jx), C.byref(jy)) != 0


def test_jitter_known_answers(vrt):
    # Halton(2,3) at indices 1..4: (1/2,1/3) (1/4,2/3) (3/4,1/9) (1/8,4/9), minus 0.5
    L = vrt.lib()
    exp = [(0.0, 1 / 3 - 0.5), (-0.25, 2 / 3 - 0.5), (0.25, 1 / 9 - 0.5), (-0.375, 4 / 9 - 0.5)]
    for i, (ex, ey) in enumerate(exp):
        jx, jy = C.c_float(), C.c_float()
        assert L.vrt_jitter_offset(i, 23, C.byref(jx), C.byref(jy)) == 0
        assert jx.value == ex and abs(jy.value - ey) < 1e-6


def test_upscaler_update_sequence_keeps_reference_wrap(vrt, oracle):
    """frameCount++ ; if (frameCount > phaseCount) frameCount = 0  (upscaler_stage.cpp:67-69): the counter runs
    0..phaseCount inclusive, so index phaseCount % phaseCount = 0 repeats the first offset once per cycle and the
    `frame` push constant takes phaseCount + 1 distinct values."""
    st = vrt.VoxelRenderSettings(targetResolution=(1920, 1080))          # BALANCED -> 1129 wide, 23 phases
    up = vrt.UpscalerStage(None, st)
    phases = up.phaseCount()
    assert phases == 23
    frames, jit = [], []
    for _ in range(2 * (phases + 1) + 3):
        frames.append(up.frameCount)                                     # value used to index the sequence
        up.update(1.0 / 60.0)
        jit.append((up.jitterX, up.jitterY))
    assert frames[:phases + 2] == list(range(phases + 1)) + [0]
    for f, (jx, jy) in zip(frames, jit):
        _, ex, ey = oracle.jitter(f % phases, st.renderResolution()[0], 1920)
        assert (np.float32(jx), np.float32(jy)) == (np.float32(ex), np.float32(ey))
    assert jit[phases] == jit[0] == jit[phases + 1]                      # the repeated sample
    assert abs(up._deltaMsec - 1000.0 / 60.0) < 1e-9


def test_camera_mouse_and_path(vrt):
    cam = vrt.CameraController(position=(8.0, 8.0, -50.0), yaw=90.0, pitch=0.0)
    cam.mouse(10.0, 100.0)                                               # yaw -= 10, pitch clamps at -90
    assert cam.yaw == 80.0 and cam.pitch == -90.0
    cam.mouse(0.0, -500.0)
    assert cam.pitch == 90.0
    cam = vrt.CameraController(position=(8.0, 8.0, -50.0), yaw=90.0, pitch=0.0)
    keys = [vrt.CameraKey(frames=3, forward=1.0), vrt.CameraKey(frames=2, strafe=-1.0, mouseX=1.5),
            vrt.CameraKey(frames=1, forward=-1.0, mouseY=2.0)]
    poses = [(c.position.copy(), c.yaw, c.pitch) for c in vrt.camera_path(cam, keys, delta=1.0 / 60.0)]
    assert len(poses) == 6
    # replay by hand in float32 (camera_controller.cpp:30-44: position += 50 * normalDir * delta per pressed key)
    ref = vrt.CameraController(position=(8.0, 8.0, -50.0), yaw=90.0, pitch=0.0)
    d = np.float32(1.0 / 60.0)
    for k in keys:
        for _ in range(k.frames):
            if k.mouseX or k.mouseY:
                ref.yaw -= k.mouseX
                ref.pitch = min(max(ref.pitch - k.mouseY, -90.0), 90.0)
                ref.updateDirectionVectors()
            ref.position = (ref.position + np.float32(50.0) * ref.normalDir * d * np.float32(k.forward)
                            + ref.right * np.float32(50.0) * d * np.float32(k.strafe)).astype(np.float32)
    assert np.array_equal(poses[-1][0], ref.position) and poses[-1][1] == ref.yaw and poses[-1][2] == ref.pitch
    assert math.isclose(float(poses[2][0][2]), -50.0 + 3 * 50.0 / 60.0, rel_tol=1e-5)   # three frames forward along +z
    # basis stays orthonormal
    assert abs(float(np.dot(cam.right, cam.up))) < 1e-6 and abs(float(np.dot(cam.right, cam.normalDir))) < 1e-6


def test_renderer_state_mirrors_upscaler_stage(vrt):
    """VoxelRenderer takes frame / cameraJitter from its UpscalerStage (voxel_renderer.cpp:80-82) -- checked on the
    host objects only (no engine calls)."""
    st = vrt.VoxelRenderSettings(targetResolution=(640, 360))
    up = vrt.UpscalerStage(None, st)
    r = vrt.VoxelRenderer.__new__(vrt.VoxelRenderer)
    r._upscalerStage = up
    assert r.frameCount == 0 and r.jitter == (0.0, 0.0)
    up.update(0.016)
    assert r.frameCount == 1 and r.jitter == (up.jitterX, up.jitterY)
    r.frameCount, r.jitter = 7, (0.25, -0.125)
    assert (up.frameCount, up.jitterX, up.jitterY) == (7, 0.25, -0.125)
