"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol include/vrt.h
declares, the host-only entry points work, and compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(vrt):
    hdr = open(os.path.join(ROOT, "include", "vrt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vrt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = C.CDLL(vrt._capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/vrt.h but not exported"
    assert declared == set(vrt._capi.SYMBOLS), declared ^ set(vrt._capi.SYMBOLS)


def test_struct_layouts(vrt):
    # ScreenQuadPush offsets (screen_quad_push.hpp:5-15): camPos@0 camDir@16 camRight@32 camUp@48 bounds@64 frame@76 size@80 jitter@88
    P = vrt._capi.Push
    assert [getattr(P, f).offset for f in ("cam_pos", "cam_dir", "cam_right", "cam_up", "volume_bounds", "frame", "screen_size", "camera_jitter")] == \
        [0, 16, 32, 48, 64, 76, 80, 88]
    assert C.sizeof(P) == 96 and C.sizeof(vrt._capi.Material) == 32


def test_defaults_match_reference(vrt):
    s = vrt._capi.Settings(); vrt.lib().vrt_settings_default(C.byref(s))
    assert (s.ao_samples, s.max_steps, s.ao_steps, s.max_bounces, s.shadows) == (4, 512, 64, 5, 1)
    assert s.ambient_intensity == 1.0 and s.light_intensity == 1.0 and list(s.light_color) == [1.0] * 4
    assert np.allclose(list(s.light_dir), 1 / np.sqrt(3), rtol=1e-7)
    d = vrt._capi.DenoiserSettings(); vrt.lib().vrt_denoiser_settings_default(C.byref(d))
    assert (d.iterations, d.mode) == (2, 0) and np.allclose([d.phi_color0, d.phi_normal0, d.phi_pos0, d.step_width], [20.4, 0.01, 0.1, 2.0])
    assert vrt.lib().vrt_denoise_halo_rows(C.byref(d)) == 4
    py = vrt.VoxelRenderSettings().to_c()
    for f, _ in vrt._capi.Settings._fields_:
        a, b = getattr(py, f), getattr(s, f)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else a == b, f
    assert vrt.VoxelRenderSettings().renderResolution() == (1129, 635)        # FSR "Balanced" of 1920x1080


def test_no_cpu_fallback(vrt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = vrt.lib().vrt_ctx_create(0, C.byref(ctx))
    assert rc == 5 and b"no CPU fallback" in vrt.lib().vrt_last_error()      # VRT_ERR_NO_DEVICE
    with pytest.raises(vrt.VrtError):
        vrt.Engine(0, use_torch_stream=False)


def test_camera_controller_defaults(vrt):
    c = vrt.CameraController()          # voxel_renderer.cpp:20 + camera_controller.cpp:15-28
    assert np.allclose(c.normalDir, [0, 0, 1], atol=1e-7) and np.allclose(c.right, [1, 0, 0], atol=1e-7)
    assert np.allclose(c.up, [0, -1, 0], atol=1e-7)
    assert np.allclose(np.linalg.norm(c.direction), 1 / np.tan(np.radians(27.5)), rtol=1e-6)
    c = vrt.CameraController(yaw=0.0, pitch=30.0)
    assert np.allclose(c.normalDir, [np.cos(np.radians(30)), 0.5, 0], atol=1e-6)
    assert abs(np.dot(c.right, c.up)) < 1e-6 and abs(np.dot(c.right, c.normalDir)) < 1e-6


def test_shard_row_maps(vrt):
    D = vrt.distributed
    lib = vrt.lib()
    for H, n, sr in [(1080, 8, 16), (1080, 3, 16), (2160, 8, 32), (100, 4, 16), (30, 8, 16), (16, 2, 16)]:
        seen = np.zeros(H, int)
        for r in range(n):
            rm = D.packed_row_map(H, r, n, sr)
            sh = vrt._capi.Shard(r, n, sr)
            assert len(rm) == D.packed_rows(H, n, sr) == lib.vrt_shard_rows(H, C.byref(sh))
            rows = rm[rm >= 0]
            assert (np.diff(rows) > 0).all()
            seen[rows] += 1
            assert ((rows // sr) % n == r).all()
        assert (seen == 1).all()                                  # every row owned exactly once
    assert lib.vrt_shard_rows(1080, None) == 1080
