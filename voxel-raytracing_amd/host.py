"""Host-side mirror of the reference's call surface for the hot path (Python over the C-ABI).

Same names, argument meaning and error behaviour as the reference objects (paths relative to the
reference root):

  VoxelRenderSettings   source/voxels/voxel_render_settings.hpp:6-59, .cpp:3-13
  CameraController      source/voxels/resource/camera_controller.cpp:15-28
  VoxelScene            source/voxels/resource/voxel_scene.hpp:18-34, .cpp:33-133
  GeometryStage         source/voxels/stages/geometry_stage.hpp:19-53, .cpp:106-153
  DenoiserStage         source/voxels/stages/denoiser_stage.hpp:22-47, .cpp:143-258
  VoxelRenderer         source/voxels/voxel_renderer.hpp:17-38, .cpp:16-94
  Engine                source/engine/engine.hpp:71-76 (reduced to: device + stream)

torch is used only for device buffers and the stream; every computation goes through libvrt_hip.so.
"""
import ctypes as C
import enum
import math
import os
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

from . import _capi
from ._capi import check, lib


# --------------------------------------------------------------------------------------------------
# settings (voxel_render_settings.hpp)
# --------------------------------------------------------------------------------------------------

class FsrScaling(enum.IntEnum):          # :6-13
    NONE = 10
    QUALITY = 15
    BALANCED = 17
    PERFORMANCE = 20
    ULTRA_PERFORMANCE = 30


@dataclass
class FsrSettings:                       # :15-19  (FSR2 itself is out of scope; only the render scale applies)
    enable: bool = True
    scaling: FsrScaling = FsrScaling.BALANCED


@dataclass
class DenoiserSettings:                  # :21-29
    enable: bool = True
    iterations: int = 2
    phiColor0: float = 20.4
    phiNormal0: float = 1e-2
    phiPos0: float = 1e-1
    stepWidth: float = 2.0
    mode: int = _capi.DENOISE_CANONICAL  # build extension: as-shipped std140 aliasing mode


@dataclass
class AmbientOcclusionSettings:          # :31-35
    numSamples: int = 4
    intensity: float = 1.0


def _norm111():
    v = np.float32(1.0) / np.sqrt(np.float32(3.0))
    return (float(v), float(v), float(v))


@dataclass
class LightSettings:                     # :37-42
    direction: Tuple[float, float, float] = field(default_factory=_norm111)
    color: Tuple[float, float, float, float] = (1.0, 1.0, 1.0, 1.0)
    intensity: float = 1.0


@dataclass
class TraceSettings:
    """The shader's compile-time constants (voxel_volume.frag:68-69,219) as runtime knobs."""
    maxRaySteps: int = 512
    aoSteps: int = 64
    maxReflections: int = 5
    shadows: bool = True
    traversal: int = _capi.TRAVERSAL_AUTO
    splitKernels: bool = False         # K1 + K2 over the compacted hit list instead of the megakernel


def _scale(scaling: FsrScaling, dim: int) -> int:      # voxel_render_settings.cpp:3-6
    return int(np.float32(np.float32(10.0) / np.float32(int(scaling))) * np.float32(dim))


@dataclass
class VoxelRenderSettings:               # :44-59
    targetResolution: Tuple[int, int] = (1920, 1080)
    fsrSetttings: FsrSettings = field(default_factory=FsrSettings)       # (sic) reference spelling
    denoiserSettings: DenoiserSettings = field(default_factory=DenoiserSettings)
    occlusionSettings: AmbientOcclusionSettings = field(default_factory=AmbientOcclusionSettings)
    lightSettings: LightSettings = field(default_factory=LightSettings)
    traceSettings: TraceSettings = field(default_factory=TraceSettings)
    voxPath: str = "../resource/treehouse.vox"
    skyboxPath: str = "../resource/rustig_koppie.hdr"

    def renderResolution(self) -> Tuple[int, int]:     # voxel_render_settings.cpp:8-13
        if self.fsrSetttings.enable:
            return (_scale(self.fsrSetttings.scaling, self.targetResolution[0]),
                    _scale(self.fsrSetttings.scaling, self.targetResolution[1]))
        return tuple(self.targetResolution)

    def to_c(self) -> _capi.Settings:
        s = _capi.Settings()
        s.ao_samples = int(self.occlusionSettings.numSamples)          # geometry_stage.cpp:135
        s.ambient_intensity = float(self.occlusionSettings.intensity)  # :136
        s.light_dir[:] = [float(x) for x in self.lightSettings.direction]   # :140
        s.light_intensity = float(self.lightSettings.intensity)        # :139
        s.light_color[:] = [float(x) for x in self.lightSettings.color]     # :141
        t = self.traceSettings
        s.max_steps, s.ao_steps, s.max_bounces = int(t.maxRaySteps), int(t.aoSteps), int(t.maxReflections)
        s.shadows = 1 if t.shadows else 0
        s.traversal = int(t.traversal)
        s.flags = 4 if t.splitKernels else 0
        return s

    def denoiser_to_c(self) -> _capi.DenoiserSettings:
        d = self.denoiserSettings
        return _capi.DenoiserSettings(int(d.iterations), float(d.phiColor0), float(d.phiNormal0),
                                      float(d.phiPos0), float(d.stepWidth), int(d.mode))

    @staticmethod
    def primary_only(resolution=(1920, 1080), traversal=_capi.TRAVERSAL_AUTO) -> "VoxelRenderSettings":
        """BASELINE "primary rays only": ao_samples = 0 (ambient = 1), no shadow ray, no bounces, no FSR scale."""
        s = VoxelRenderSettings(targetResolution=tuple(resolution))
        s.fsrSetttings.enable = False
        s.denoiserSettings.enable = False
        s.occlusionSettings.numSamples = 0
        s.traceSettings.shadows = False
        s.traceSettings.maxReflections = 0
        s.traceSettings.traversal = traversal
        return s


# --------------------------------------------------------------------------------------------------
# camera (camera_controller.cpp:15-28)
# --------------------------------------------------------------------------------------------------

def _f32(x):
    return np.float32(x)


def _normalize(v):
    v = np.asarray(v, dtype=np.float32)
    n = np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))
    return (v / n).astype(np.float32)


class CameraController:
    def __init__(self, position=(8.0, 8.0, -50.0), yaw=90.0, pitch=0.0,
                 focalLength=1.0 / math.tan(math.radians(55.0 / 2))):       # voxel_renderer.cpp:20
        self.position = np.asarray(position, dtype=np.float32)
        self.yaw = float(yaw)
        self.pitch = float(pitch)
        self.focalLength = float(focalLength)
        self.updateDirectionVectors()

    def updateDirectionVectors(self):                                       # :15-28
        worldUp = np.array([0.0, -1.0, 0.0], dtype=np.float32)
        yaw, pitch = _f32(math.radians(self.yaw)), _f32(math.radians(self.pitch))
        nd = np.array([np.cos(yaw) * np.cos(pitch), np.sin(pitch), np.sin(yaw) * np.cos(pitch)], dtype=np.float32)
        self.normalDir = _normalize(nd)
        self.right = _normalize(np.cross(self.normalDir, worldUp).astype(np.float32))
        self.up = _normalize(np.cross(self.right, self.normalDir).astype(np.float32))
        self.direction = (self.normalDir * _f32(self.focalLength)).astype(np.float32)

    def update(self, delta: float, forward=0.0, strafe=0.0):                # :30-44 (keys -> scripted axes)
        cameraSpeed = _f32(50.0)
        self.position = (self.position + cameraSpeed * self.normalDir * _f32(delta) * _f32(forward)
                         + self.right * cameraSpeed * _f32(delta) * _f32(strafe)).astype(np.float32)
        self.updateDirectionVectors()

    def mouse(self, offsetX: float, offsetY: float):                        # :46-68 (right button held)
        self.yaw -= float(_f32(offsetX))
        self.pitch = float(min(max(_f32(self.pitch) - _f32(offsetY), _f32(-90.0)), _f32(90.0)))
        self.updateDirectionVectors()


@dataclass
class CameraKey:
    """One scripted input sample: what the keyboard / mouse callbacks of camera_controller.cpp:30-68 would have seen
    during `frames` consecutive frames (W/S -> forward +-1, D/A -> strafe +-1, cursor motion in pixels per frame)."""
    frames: int = 1
    forward: float = 0.0
    strafe: float = 0.0
    mouseX: float = 0.0
    mouseY: float = 0.0


def camera_path(camera: CameraController, keys, delta: float = 1.0 / 60.0):
    """Scripted fly-through: replays `keys` through CameraController.update / mouse and yields the controller after every
    frame (the same object, mutated), so a benchmark can render an animated sequence without a window."""
    for k in keys:
        for _ in range(int(k.frames)):
            if k.mouseX or k.mouseY:
                camera.mouse(k.mouseX, k.mouseY)
            camera.update(delta, k.forward, k.strafe)
            yield camera


# --------------------------------------------------------------------------------------------------
# engine / scene
# --------------------------------------------------------------------------------------------------

def _torch():
    import torch
    return torch


class Engine:
    """Device + stream holder (Engine::init, engine.cpp:14).  Work is enqueued on torch's current
    stream for the device so that torch ops (and RCCL collectives) order with the kernels."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        l = lib()
        self.device = int(device)
        torch = _torch()
        try:
            torch.cuda.init()                    # torch's HIP runtime first (see _capi.lib)
        except Exception:
            pass                                 # no GPU: vrt_ctx_create below reports it (VRT_ERR_NO_DEVICE, no CPU fallback)
        self._ctx = C.c_void_p()
        check(l.vrt_ctx_create(self.device, C.byref(self._ctx)))
        self.torch_device = torch.device("cuda", self.device)      # where the stages allocate their images
        if use_torch_stream:
            stream = torch.cuda.current_stream(self.torch_device)
            check(l.vrt_ctx_set_stream(self._ctx, C.c_void_p(stream.cuda_stream)))
        # use_torch_stream=False: the context keeps its own non-blocking stream (a second frame in flight); torch
        # allocations and fills on torch's stream must then be synchronised by the caller before the first use

    @property
    def ctx(self):
        if not self._ctx:
            raise RuntimeError("Engine was destroyed")
        return self._ctx

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int()
        check(lib().vrt_device_info(self.ctx, name, 256, C.byref(cus)))
        return name.value.decode(), cus.value

    def synchronize(self):
        check(lib().vrt_ctx_synchronize(self.ctx))

    def set_timing(self, enabled: bool):
        """Per-call HIP event recording inside the library (vrt_last_timings); off for throughput runs."""
        check(lib().vrt_ctx_set_timing(self.ctx, 1 if enabled else 0))

    def set_option(self, name: str, value) -> None:
        """Development switch of this context (include/vrt.h vrt_ctx_set_option): speed only, never a result."""
        check(lib().vrt_ctx_set_option(self.ctx, name.encode(), int(value)))

    def option(self, name: str) -> int:
        v = C.c_int32()
        check(lib().vrt_ctx_get_option(self.ctx, name.encode(), C.byref(v)))
        return int(v.value)

    def options(self, **kv):
        """with engine.options(tile_tags=0, sky_fast=0): ...  -- the switches are put back on exit."""
        eng = self

        class _Scope:
            def __enter__(self_):
                self_.old = {k: eng.option(k) for k in kv}
                for k, v in kv.items():
                    eng.set_option(k, v)
                return eng

            def __exit__(self_, *a):
                for k, v in self_.old.items():
                    eng.set_option(k, v)
        return _Scope()

    def last_timings(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        check(lib().vrt_last_timings(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        return {"primary_ms": a.value, "geometry_ms": b.value, "denoise_ms": c.value}

    def destroy(self):
        if self._ctx:
            lib().vrt_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def materials_from_numpy(pal: np.ndarray):
    """pal: (256, 5) float32 [r, g, b, a, metallic] already linear -> ctypes Material[256]."""
    arr = (_capi.Material * 256)()
    pal = np.asarray(pal, dtype=np.float32)
    for i in range(256):
        arr[i].diffuse[:] = [float(x) for x in pal[i, :4]]
        arr[i].metallic = float(pal[i, 4])
    return arr


def materials_to_numpy(arr) -> np.ndarray:
    out = np.zeros((256, 5), dtype=np.float32)
    for i in range(256):
        out[i, :4] = list(arr[i].diffuse)
        out[i, 4] = arr[i].metallic
    return out


class VoxelScene:
    """VoxelScene(engine, filename, skyboxFilename) (voxel_scene.cpp:33).  `skyboxFilename` may be an
    (h, w, 4) float32 array (decoders for .hdr are a later-round item); None keeps the 1x1 white default.
    Raises RuntimeError with the reference's messages on load failures."""

    def __init__(self, engine: Engine, filename: Optional[str] = None, skyboxFilename=None, *, _handle=None):
        self.engine = engine
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            if filename is None:
                raise ValueError("VoxelScene needs a filename (or use VoxelScene.from_dense / from_memory)")
            rc = lib().vrt_scene_load_vox_file(engine.ctx, str(filename).encode(), C.byref(self._h))
            self._raise(rc)
        self._update_dims()
        if skyboxFilename is not None:
            self.set_sky(skyboxFilename)

    @staticmethod
    def _raise(rc):
        if rc != _capi.VRT_OK:
            msg = lib().vrt_last_error().decode()
            if rc in (2, 3, 4):
                raise RuntimeError(msg)          # reference: std::runtime_error, voxel_scene.cpp:42,46,50
            raise _capi.VrtError(rc, msg)

    @classmethod
    def from_memory(cls, engine: Engine, buf: bytes, sky=None):
        h = C.c_void_p()
        b = (C.c_uint8 * len(buf)).from_buffer_copy(buf)
        cls._raise(lib().vrt_scene_load_vox_mem(engine.ctx, b, len(buf), C.byref(h)))
        return cls(engine, skyboxFilename=sky, _handle=h)

    @classmethod
    def from_dense(cls, engine: Engine, voxels: np.ndarray, palette: np.ndarray, sky=None, noise=None):
        """voxels: uint8 array indexed [z, y, x] (C order == x + y*W + z*W*H); palette: (256,5) float32."""
        v = np.ascontiguousarray(voxels, dtype=np.uint8)
        D, H, W = v.shape
        h = C.c_void_p()
        cls._raise(lib().vrt_scene_from_dense(engine.ctx, v.ctypes.data_as(C.c_void_p), W, H, D,
                                              materials_from_numpy(palette), C.byref(h)))
        s = cls(engine, skyboxFilename=sky, _handle=h)
        if noise is not None:
            s.set_blue_noise(noise)
        return s

    @classmethod
    def from_bricks(cls, engine: Engine, grid: np.ndarray, pool: np.ndarray, palette: np.ndarray, sky=None, noise=None):
        """The volume in 8^3 bricks (vrt_scene_from_bricks): grid uint32 [nbz, nby, nbx] (0 = empty, else 1 + index into pool),
        pool uint8 [n, 8, 8, 8] indexed [z, y, x] within the brick.  Renders like from_dense of the same content."""
        g = np.ascontiguousarray(grid, dtype=np.uint32)
        p = np.ascontiguousarray(pool, dtype=np.uint8).reshape(-1, 512)
        nbz, nby, nbx = g.shape
        h = C.c_void_p()
        cls._raise(lib().vrt_scene_from_bricks(engine.ctx, g.ctypes.data_as(C.c_void_p), nbx, nby, nbz,
                                               p.ctypes.data_as(C.c_void_p) if p.size else None, p.shape[0],
                                               materials_from_numpy(palette), C.byref(h)))
        s = cls(engine, skyboxFilename=sky, _handle=h)
        if noise is not None:
            s.set_blue_noise(noise)
        return s

    def trim(self) -> None:
        """Drop the diagnostic copy of the clearance fields a launch with count planes built (vrt_scene_trim)."""
        check(lib().vrt_scene_trim(self.engine.ctx, self._h))

    def memory_bytes(self) -> int:
        n = C.c_uint64()
        check(lib().vrt_scene_memory(self._h, C.byref(n)))
        return int(n.value)

    def _update_dims(self):
        d = (C.c_uint32 * 3)()
        check(lib().vrt_scene_info(self._h, d))
        self.width, self.height, self.depth = int(d[0]), int(d[1]), int(d[2])

    def set_sky(self, rgba):
        if isinstance(rgba, (str, bytes, os.PathLike)):       # Texture2D(skyboxPath, 4, RGBA32F): decode on the host
            rc = lib().vrt_scene_set_sky_file(self.engine.ctx, self._h, os.fsencode(rgba))
            if rc != _capi.VRT_OK:
                raise RuntimeError(lib().vrt_last_error().decode())      # "Could not load image <path>" (texture_2d.cpp:42)
            return
        self._set_sky_array(rgba)

    def _set_sky_array(self, rgba: np.ndarray):
        a = np.ascontiguousarray(rgba, dtype=np.float32)
        assert a.ndim == 3 and a.shape[2] == 4
        check(lib().vrt_scene_set_sky(self.engine.ctx, self._h, a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0]))

    def set_blue_noise(self, rgba8):
        if isinstance(rgba8, (str, bytes, os.PathLike)):      # Texture2D("blue_noise_rgba.png", 4, RGBA8_UNORM)
            rc = lib().vrt_scene_set_blue_noise_file(self.engine.ctx, self._h, os.fsencode(rgba8))
            if rc != _capi.VRT_OK:
                raise RuntimeError(lib().vrt_last_error().decode())
            return
        self._set_noise_array(rgba8)

    def _set_noise_array(self, rgba8: np.ndarray):
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 4
        check(lib().vrt_scene_set_blue_noise(self.engine.ctx, self._h, a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0]))

    def download(self):
        v = np.empty((self.depth, self.height, self.width), dtype=np.uint8)
        pal = (_capi.Material * 256)()
        check(lib().vrt_scene_download(self.engine.ctx, self._h, v.ctypes.data_as(C.c_void_p), pal))
        return v, materials_to_numpy(pal)

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("VoxelScene was destroyed")
        return self._h

    def destroy(self):
        if self._h:
            lib().vrt_scene_free(self.engine.ctx, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def load_image(path):
    """Host-side decoder behind Texture2D (Radiance .hdr -> float32 RGBA, PNG -> uint8 RGBA)."""
    hdr, w, h, px = C.c_int(), C.c_uint32(), C.c_uint32(), C.c_void_p()
    rc = lib().vrt_image_load(os.fsencode(path), C.byref(hdr), C.byref(w), C.byref(h), C.byref(px))
    if rc != _capi.VRT_OK:
        raise RuntimeError(lib().vrt_last_error().decode())
    n = w.value * h.value * 4
    if hdr.value:
        a = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_float)), shape=(n,)).copy().reshape(h.value, w.value, 4)
    else:
        a = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint8)), shape=(n,)).copy().reshape(h.value, w.value, 4)
    lib().vrt_host_free(px)
    return a


def write_image(path, array):
    """PNG / PPM from uint8 RGBA (H,W,4); PFM from float32 (H,W,3|4)."""
    a = np.ascontiguousarray(array)
    p = os.fsencode(path)
    H, W = a.shape[:2]
    if a.dtype == np.uint8:
        fn = lib().vrt_image_write_ppm if str(path).lower().endswith(".ppm") else lib().vrt_image_write_png
        check(fn(p, a.ctypes.data_as(C.c_void_p), W, H))
    else:
        a = np.ascontiguousarray(a, dtype=np.float32)
        check(lib().vrt_image_write_pfm(p, a.ctypes.data_as(C.c_void_p), W, H, a.shape[2]))


def vox_flatten_host(buf: bytes):
    """Host-only .vox parse + flatten (no device): returns (voxels[z,y,x], palette(256,5), n_instances, dropped)."""
    dims = (C.c_uint32 * 3)()
    vox = C.c_void_p()
    pal = (_capi.Material * 256)()
    ninst, dropped = C.c_uint32(), C.c_uint64()
    b = (C.c_uint8 * len(buf)).from_buffer_copy(buf)
    rc = lib().vrt_vox_flatten_host(b, len(buf), dims, C.byref(vox), pal, C.byref(ninst), C.byref(dropped))
    VoxelScene._raise(rc)
    W, H, D = int(dims[0]), int(dims[1]), int(dims[2])
    arr = np.ctypeslib.as_array(C.cast(vox, C.POINTER(C.c_uint8)), shape=(D * H * W,)).copy().reshape(D, H, W)
    lib().vrt_host_free(vox)
    return arr, materials_to_numpy(pal), ninst.value, dropped.value


# --------------------------------------------------------------------------------------------------
# push constants (voxel_renderer.cpp:72-83)
# --------------------------------------------------------------------------------------------------

def make_push(camera: CameraController, scene_dims, resolution, frame=0, jitter=(0.0, 0.0)) -> _capi.Push:
    p = _capi.Push()
    p.screen_size[:] = [int(resolution[0]), int(resolution[1])]
    p.volume_bounds[:] = [int(scene_dims[0]), int(scene_dims[1]), int(scene_dims[2])]
    p.cam_pos[:] = [float(camera.position[0]), float(camera.position[1]), float(camera.position[2]), 1.0]
    p.cam_dir[:] = [float(camera.direction[0]), float(camera.direction[1]), float(camera.direction[2]), 0.0]
    p.cam_up[:] = [float(camera.up[0]), float(camera.up[1]), float(camera.up[2]), 0.0]
    p.cam_right[:] = [float(camera.right[0]), float(camera.right[1]), float(camera.right[2]), 0.0]
    p.frame = int(frame) & 0xFFFFFFFF
    p.camera_jitter[:] = [float(jitter[0]), float(jitter[1])]
    return p


# --------------------------------------------------------------------------------------------------
# stages
# --------------------------------------------------------------------------------------------------

_PLANE_SPECS = {
    # name: (torch dtype name, trailing shape)
    "color8": ("uint8", (4,)), "depth": ("float32", ()), "motion": ("float32", (2,)), "mask8": ("uint8", ()),
    "position": ("float32", (4,)), "normal8": ("int8", (4,)),
    "color_f": ("float32", (3,)), "hit_id": ("uint8", ()), "hit_voxel": ("int16", (3,)), "hit_mask": ("uint8", ()),
    "steps_primary": ("int32", ()), "steps_total": ("int32", ()), "rays_total": ("int32", ()),
    "color8_strips": ("uint8", (4,)),      # (allocated at the full frame's size: the packed strips of a rank fill its first rows)
}
GBUFFER_PLANES = ("color8", "depth", "motion", "mask8", "position", "normal8")
DEBUG_PLANES = ("color_f", "hit_id", "hit_voxel", "hit_mask", "steps_primary", "steps_total", "rays_total")
# Planes whose presence changes which march a launch runs: the count planes select the clearance fields without open cells
# (the iterations of the reference's loop), hit_voxel the look-up loop that keeps mapPos.  A stage with debug planes renders
# them in a launch of their own, so that every other plane comes from the march a caller without debug planes gets.
DIAGNOSTIC_MARCH_PLANES = ("hit_voxel", "steps_primary", "steps_total")


class GeometryBuffer:
    """GeometryBuffer (geometry_stage.hpp:19-27): color, depth, motion, mask, normal, position (+ debug planes).
    Each plane is a torch tensor on the engine's device, shape (H, W, ...)."""

    def __init__(self, engine: Engine, W: int, H: int, planes=GBUFFER_PLANES):
        torch = _torch()
        self.W, self.H = int(W), int(H)
        self.planes = {}
        for n in planes:
            dt, tail = _PLANE_SPECS[n]
            self.planes[n] = torch.zeros((self.H, self.W) + tail, dtype=getattr(torch, dt), device=engine.torch_device)

    def __getattr__(self, n):
        alias = {"color": "color8", "mask": "mask8", "normal": "normal8"}
        planes = self.__dict__.get("planes", {})
        n2 = alias.get(n, n)
        if n2 in planes:
            return planes[n2]
        raise AttributeError(n)

    def to_c(self, only=None, without=()) -> _capi.Frame:
        f = _capi.Frame()
        for n in _capi.FRAME_PLANES:
            t = self.planes.get(n)
            if (only is not None and n not in only) or n in without:
                t = None
            setattr(f, n, t.data_ptr() if t is not None else None)
        return f

    def to_c_split(self):
        """[frame of the product march, frame of the diagnostic march or None] (DIAGNOSTIC_MARCH_PLANES)."""
        if not any(n in self.planes for n in DIAGNOSTIC_MARCH_PLANES):
            return [self.to_c(), None]
        return [self.to_c(without=DIAGNOSTIC_MARCH_PLANES), self.to_c(only=DIAGNOSTIC_MARCH_PLANES)]

    def numpy(self):
        return {n: t.cpu().numpy() for n, t in self.planes.items()}


def make_shard(rank=0, nranks=1, strip_rows=16):
    if nranks <= 1:
        return None
    return _capi.Shard(int(rank), int(nranks), int(strip_rows))


class GeometryStage:
    """GeometryStage(engine, settings, scene, noise) / record() (geometry_stage.cpp:14,106)."""

    def __init__(self, engine: Engine, settings: VoxelRenderSettings, scene: VoxelScene, noise=None,
                 debug_planes: bool = False):
        self.engine, self._settings, self._scene = engine, settings, scene
        if noise is not None:
            scene.set_blue_noise(noise)
        self._debug = debug_planes
        self._buffer = None

    def _targets(self):
        W, H = self._settings.renderResolution()
        if self._buffer is None or (self._buffer.W, self._buffer.H) != (W, H):     # RENDER_RESIZE recreation
            planes = GBUFFER_PLANES + (DEBUG_PLANES if self._debug else ())
            self._buffer = GeometryBuffer(self.engine, W, H, planes)
        return self._buffer

    def record(self, push: _capi.Push, shard: Optional[_capi.Shard] = None) -> GeometryBuffer:
        gb = self._targets()
        st = self._settings.to_c()
        for fr in gb.to_c_split():
            if fr is not None:
                check(lib().vrt_render_geometry(self.engine.ctx, self._scene.handle, C.byref(push), C.byref(st), C.byref(fr),
                                                C.byref(shard) if shard is not None else None))
        return gb

    def prepare(self, shard: Optional[_capi.Shard] = None):
        """Frame loops whose settings do not change between frames: marshal the settings / target structs once and
        return launch(push) -> GeometryBuffer (a Python-side economy only; the same C-ABI call is made)."""
        gb = self._targets()
        st, fr, fr_diag = self._settings.to_c(), *gb.to_c_split()
        fn, ctx, scene = lib().vrt_render_geometry, self.engine.ctx, self._scene.handle
        pst, pfr = C.byref(st), C.byref(fr)
        psh = C.byref(shard) if shard is not None else None

        def launch(push: _capi.Push) -> GeometryBuffer:
            rc = fn(ctx, scene, C.byref(push), pst, pfr, psh)
            if rc == 0 and fr_diag is not None:
                rc = fn(ctx, scene, C.byref(push), pst, C.byref(fr_diag), psh)
            if rc != 0:
                check(rc)
            return gb
        launch._keepalive = (st, fr, fr_diag, shard, gb)
        return launch

    def prepare_batch(self, n: int, shard: Optional[_capi.Shard] = None, shards=None):
        """n frames per launch (vrt_render_geometry_batch): consecutive poses of an animation or the frames of a multi-GPU
        batch.  Returns launch(pushes) -> [GeometryBuffer] * n; every frame has its own planes, allocated here once.
        shards: one strip assignment per frame instead of one for all (vrt_render_geometry_slots)."""
        W, H = self._settings.renderResolution()
        planes = GBUFFER_PLANES + (DEBUG_PLANES if self._debug else ())
        gbs = [GeometryBuffer(self.engine, W, H, planes) for _ in range(int(n))]
        st = self._settings.to_c()
        split = [g.to_c_split() for g in gbs]
        frs = (_capi.Frame * int(n))(*[sp[0] for sp in split])
        frs_diag = (_capi.Frame * int(n))(*[sp[1] for sp in split]) if split and split[0][1] is not None else None
        arr = (_capi.Push * int(n))()
        fn, ctx, scene = lib().vrt_render_geometry_batch, self.engine.ctx, self._scene.handle
        pst = C.byref(st)
        psh = C.byref(shard) if shard is not None else None
        if shards is not None:
            if len(shards) != int(n):
                raise ValueError("prepare_batch: one shard per frame is required")
            shard = (_capi.Shard * int(n))(*shards)
            fn, psh = lib().vrt_render_geometry_slots, shard

        def launch(pushes, frames=None) -> list:
            """frames: another (_capi.Frame * n) table to render into (e.g. a copy of launch._keepalive[1] with other
            color8_strips pointers); default: the planes allocated above."""
            for k, p in enumerate(pushes):
                arr[k] = p
            rc = fn(ctx, scene, len(pushes), arr, pst, frs if frames is None else frames, psh)
            if rc == 0 and frs_diag is not None and frames is None:
                rc = fn(ctx, scene, len(pushes), arr, pst, frs_diag, psh)
            if rc != 0:
                check(rc)
            return gbs[:len(pushes)]
        launch._keepalive = (st, frs, arr, shard, gbs, frs_diag)
        return launch

    def record_batch(self, pushes, shard: Optional[_capi.Shard] = None) -> list:
        return self.prepare_batch(len(pushes), shard)(pushes)


class DenoiserStage:
    """DenoiserStage(engine, settings) / record(color, normal, pos) (denoiser_stage.cpp:22,156)."""

    def __init__(self, engine: Engine, settings: VoxelRenderSettings):
        self.engine, self._settings = engine, settings
        self._targets = None

    def record(self, colorInput, normalInput, posInput, shard: Optional[_capi.Shard] = None):
        torch = _torch()
        H, W = colorInput.shape[0], colorInput.shape[1]
        if self._targets is None or self._targets[0].shape != colorInput.shape:
            self._targets = [torch.zeros_like(colorInput), torch.zeros_like(colorInput)]     # _colorTargets ring
        ds = self._settings.denoiser_to_c()
        res = C.c_void_p()
        check(lib().vrt_denoise(self.engine.ctx, W, H, C.byref(ds), colorInput.data_ptr(), normalInput.data_ptr(),
                                posInput.data_ptr(), self._targets[0].data_ptr(), self._targets[1].data_ptr(),
                                C.byref(shard) if shard is not None else None, C.byref(res)))
        for t in self._targets:
            if t.data_ptr() == res.value:
                return t
        return colorInput

    def guard(self, pass_index: int) -> float:
        """Guard (RGBA8 codes) of weighted pass `pass_index` under the current settings (vrt_denoise_guard); inf: computed literally."""
        ds = self._settings.denoiser_to_c()
        g = C.c_float()
        check(lib().vrt_denoise_guard(C.byref(ds), int(pass_index), C.byref(g)))
        return float(g.value)

    def redone(self, pass_index: int) -> int:
        """Pixels of pass `pass_index` that the latest record() on this engine evaluated a second time, literally."""
        n = C.c_uint32()
        check(lib().vrt_debug_denoise_redone(self.engine.ctx, int(pass_index), C.byref(n)))
        return int(n.value)


class UpscalerStage:
    """UpscalerStage (upscaler_stage.cpp).  update() reproduces the jitter / frame sequence of :59-70 bit for bit
    (including the `frameCount > phaseCount` wrap, which lets frameCount reach phaseCount once per cycle).  The FSR2
    dispatch of record() (:72-161, a prebuilt third-party library) is out of scope; its stand-in is an exact N-frame
    accumulation of the jittered frames followed by the bilinear upscale to targetResolution (vrt_accumulate /
    vrt_resolve / vrt_blit)."""

    def __init__(self, engine: Engine, settings: VoxelRenderSettings):
        self.engine, self._settings = engine, settings
        self.jitterX = 0.0
        self.jitterY = 0.0
        self.frameCount = 0
        self._deltaMsec = 0.0
        self._accum = self._resolved = self._target = None
        self.accumulated = 0

    def phaseCount(self) -> int:
        return int(lib().vrt_jitter_phase_count(self._settings.renderResolution()[0], self._settings.targetResolution[0]))

    def update(self, delta: float):                                         # :59-70
        self._deltaMsec = delta * 1000
        phases = self.phaseCount()
        jx, jy = C.c_float(), C.c_float()
        check(lib().vrt_jitter_offset(self.frameCount % phases, phases, C.byref(jx), C.byref(jy)))
        self.jitterX, self.jitterY = jx.value, jy.value
        self.frameCount += 1
        if self.frameCount > phases:
            self.frameCount = 0

    def reset(self):
        """Drop the accumulated history (camera or scene changed)."""
        self.accumulated = 0

    def record(self, color):
        torch = _torch()
        H, W = color.shape[0], color.shape[1]
        TW, TH = self._settings.targetResolution
        if self._accum is None or self._accum.shape[:2] != (H, W):
            self._accum = torch.zeros((H, W, 4), dtype=torch.int32, device=color.device)
            self._resolved = torch.zeros((H, W, 4), dtype=torch.uint8, device=color.device)
            self.accumulated = 0
        if self._target is None or self._target.shape[:2] != (TH, TW):
            self._target = torch.zeros((TH, TW, 4), dtype=torch.uint8, device=color.device)
        check(lib().vrt_accumulate(self.engine.ctx, color.data_ptr(), self._accum.data_ptr(), W, H, 1 if self.accumulated == 0 else 0))
        self.accumulated += 1
        check(lib().vrt_resolve(self.engine.ctx, self._accum.data_ptr(), self._resolved.data_ptr(), W, H, self.accumulated))
        check(lib().vrt_blit(self.engine.ctx, self._resolved.data_ptr(), W, H, self._target.data_ptr(), TW, TH))
        return self._target


class BlitStage:
    """BlitStage::record + shader/blit.frag (blit_stage.cpp:41-75): centre-cropped bilinear copy into a window-sized
    RGBA8 target (the swapchain image of the reference; here a device tensor)."""

    def __init__(self, engine: Engine, settings: VoxelRenderSettings):
        self.engine, self._settings = engine, settings
        self._target = None

    def record(self, source, windowSize):
        torch = _torch()
        TW, TH = int(windowSize[0]), int(windowSize[1])
        if self._target is None or self._target.shape[:2] != (TH, TW):
            self._target = torch.zeros((TH, TW, 4), dtype=torch.uint8, device=source.device)
        check(lib().vrt_blit(self.engine.ctx, source.data_ptr(), source.shape[1], source.shape[0],
                             self._target.data_ptr(), TW, TH))
        return self._target


class VoxelRenderer:
    """VoxelRenderer (voxel_renderer.cpp:16-94): scene + camera + settings -> GeometryStage -> DenoiserStage ->
    UpscalerStage stand-in -> BlitStage -> RGBA8 image.  update() advances the camera and the jitter sequence as
    :33-39 does; until update() is called frame = 0 and cameraJitter = (0, 0).  GUI / swapchain are out of scope.

    temporal=False (default) keeps the frame graph at the hot path: the FSR branch of :86-87 is skipped and the
    image stays at renderResolution(); temporal=True takes that branch through the accumulation stand-in.
    windowSize=(w, h) appends the blit of :89."""

    def __init__(self, engine: Engine, settings: Optional[VoxelRenderSettings] = None, scene: Optional[VoxelScene] = None,
                 noise=None, debug_planes: bool = False, temporal: bool = False, windowSize=None):
        self.engine = engine
        self._settings = settings or VoxelRenderSettings()
        self._camera = CameraController()
        self._scene = scene if scene is not None else VoxelScene(engine, self._settings.voxPath)
        self._geometryStage = GeometryStage(engine, self._settings, self._scene, noise, debug_planes)
        self._denoiserStage = DenoiserStage(engine, self._settings)
        self._upscalerStage = UpscalerStage(engine, self._settings)
        self._blitStage = BlitStage(engine, self._settings)
        self.temporal = bool(temporal)
        self.windowSize = windowSize
        self._time = 0.0

    @property
    def settings(self):
        return self._settings

    @property
    def camera(self):
        return self._camera

    @property
    def scene(self):
        return self._scene

    @property
    def upscaler(self):
        return self._upscalerStage

    # frame / jitter live in the upscaler stage as in the reference (voxel_renderer.cpp:80-82)
    @property
    def frameCount(self):
        return self._upscalerStage.frameCount

    @frameCount.setter
    def frameCount(self, v):
        self._upscalerStage.frameCount = int(v)

    @property
    def jitter(self):
        return (self._upscalerStage.jitterX, self._upscalerStage.jitterY)

    @jitter.setter
    def jitter(self, v):
        self._upscalerStage.jitterX, self._upscalerStage.jitterY = float(v[0]), float(v[1])

    def update(self, delta: float, forward=0.0, strafe=0.0):          # :33-39
        self._time += delta
        self._camera.update(delta, forward, strafe)
        self._upscalerStage.update(delta)

    def push_constants(self) -> _capi.Push:                           # :72-83
        return make_push(self._camera, (self._scene.width, self._scene.height, self._scene.depth),
                         self._settings.renderResolution(), self.frameCount, self.jitter)

    def recordCommands(self, shard: Optional[_capi.Shard] = None):    # :55-94
        push = self.push_constants()
        gBuffer = self._geometryStage.record(push, shard)
        if self._settings.denoiserSettings.enable:
            color = self._denoiserStage.record(gBuffer.color, gBuffer.normal, gBuffer.position, shard)
        else:
            color = gBuffer.color
        self.gBuffer = gBuffer
        if self.temporal and self._settings.fsrSetttings.enable:      # :86-87
            color = self._upscalerStage.record(color)
        if self.windowSize is not None:                               # :89
            color = self._blitStage.record(color, self.windowSize)
        return color

    render = recordCommands
