"""ctypes wrapper of the CPU ORACLE (oracle/libvrt_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED (see vrt_oracle.h): the reference ships no golden vectors for this path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvrt_oracle.so")
REFVOX_PATH = os.path.join(_HERE, "_ref", "libvrt_refvox.so")


class Push(C.Structure):
    _fields_ = [("cam_pos", C.c_float * 4), ("cam_dir", C.c_float * 4), ("cam_right", C.c_float * 4),
                ("cam_up", C.c_float * 4), ("volume_bounds", C.c_uint32 * 3), ("frame", C.c_uint32),
                ("screen_size", C.c_int32 * 2), ("camera_jitter", C.c_float * 2)]


class Scene(C.Structure):
    _fields_ = [("voxels", C.c_void_p), ("dims", C.c_uint32 * 3), ("palette", C.c_void_p),
                ("sky", C.c_void_p), ("sky_w", C.c_uint32), ("sky_h", C.c_uint32),
                ("noise", C.c_void_p), ("noise_w", C.c_uint32), ("noise_h", C.c_uint32),
                ("brick_grid", C.c_void_p), ("brick_pool", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("ao_samples", C.c_uint32), ("ambient_intensity", C.c_float), ("light_dir", C.c_float * 3),
                ("light_intensity", C.c_float), ("light_color", C.c_float * 4), ("max_steps", C.c_uint32),
                ("ao_steps", C.c_uint32), ("max_bounces", C.c_uint32), ("shadows", C.c_uint32)]


PLANES = [("color_f", np.float32, (3,)), ("color8", np.uint8, (4,)), ("depth", np.float32, ()),
          ("motion", np.float32, (2,)), ("mask8", np.uint8, ()), ("position", np.float32, (4,)),
          ("normal8", np.int8, (4,)), ("hit_id", np.uint8, ()), ("hit_voxel", np.int16, (3,)),
          ("hit_mask", np.uint8, ()), ("steps_primary", np.uint32, ()), ("steps_total", np.uint32, ()),
          ("rays_total", np.uint32, ())]


class Frame(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _, _ in PLANES]


class Hit(C.Structure):
    _fields_ = [("material", C.c_uint32), ("pos", C.c_float * 3), ("normal", C.c_float * 3), ("dir", C.c_float * 3),
                ("voxel", C.c_int32 * 3), ("mask", C.c_uint32), ("steps", C.c_uint32), ("p0", C.c_float * 3),
                ("side", C.c_float * 3), ("delta", C.c_float * 3)]


class DenoiseParams(C.Structure):
    _fields_ = [("phi_color", C.c_float), ("phi_normal", C.c_float), ("phi_pos", C.c_float), ("step_width", C.c_float)]


_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        l.vo_atan2f.restype = C.c_float; l.vo_atan2f.argtypes = [C.c_float, C.c_float]
        l.vo_asinf.restype = C.c_float; l.vo_asinf.argtypes = [C.c_float]
        l.vo_expf.restype = C.c_float; l.vo_expf.argtypes = [C.c_float]
        l.vo_unorm8.restype = C.c_uint8; l.vo_unorm8.argtypes = [C.c_float]
        l.vo_snorm8.restype = C.c_int8; l.vo_snorm8.argtypes = [C.c_float]
        l.vo_render_rows.restype = None
        l.vo_render_rows.argtypes = [C.POINTER(Scene), C.POINTER(Push), C.POINTER(Params), C.POINTER(Frame), C.c_int, C.c_int]
        l.vo_render_mt.restype = None
        l.vo_render_mt.argtypes = [C.POINTER(Scene), C.POINTER(Push), C.POINTER(Params), C.POINTER(Frame), C.c_int]
        l.vo_trace_ray.restype = None
        l.vo_trace_ray.argtypes = [C.POINTER(Scene), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.POINTER(Hit)]
        l.vo_primary_ray.restype = None
        l.vo_primary_ray.argtypes = [C.POINTER(Push), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.vo_denoise_pass_params.restype = None
        l.vo_denoise_pass_params.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(DenoiseParams)]
        l.vo_denoise_pass.restype = None
        l.vo_denoise_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.POINTER(DenoiseParams), C.c_int, C.c_int, C.c_int]
        l.vo_denoise.restype = C.c_int
        l.vo_denoise.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]
        l.vo_blit.restype = None
        l.vo_blit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        l.vo_jitter_phase_count.restype = C.c_int; l.vo_jitter_phase_count.argtypes = [C.c_int, C.c_int]
        l.vo_jitter_offset.restype = None
        l.vo_jitter_offset.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib = l
    return _lib


def blit(src, tw, th):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((th, tw, 4), np.uint8)
    lib().vo_blit(src.ctypes.data, src.shape[1], src.shape[0], dst.ctypes.data, tw, th)
    return dst


def jitter(index, render_width, display_width):
    n = lib().vo_jitter_phase_count(render_width, display_width)
    x, y = C.c_float(), C.c_float()
    lib().vo_jitter_offset(index, n, C.byref(x), C.byref(y))
    return n, x.value, y.value


class OracleScene:
    """Host copy of a scene for the oracle.  voxels[z,y,x] uint8, palette (256,5) float32 (r,g,b,a,metallic)."""

    def __init__(self, voxels, palette, sky=None, noise=None, bricks=None):
        """bricks = (grid uint32 [nbz, nby, nbx], pool uint8 [n, 8, 8, 8]): the volume in 8^3 bricks instead of `voxels`
        (pass voxels=None); get_voxel then reads through the brick grid."""
        if bricks is not None:
            self.grid = np.ascontiguousarray(bricks[0], dtype=np.uint32)
            self.pool = np.ascontiguousarray(bricks[1], dtype=np.uint8)
            self.voxels = None
            D, H, W = (8 * n for n in self.grid.shape)
        else:
            self.voxels = np.ascontiguousarray(voxels, dtype=np.uint8)
            D, H, W = self.voxels.shape
        pal = np.zeros((256, 8), dtype=np.float32)
        pal[:, :5] = np.asarray(palette, dtype=np.float32)
        self.palette = pal
        self.sky = np.ascontiguousarray(sky if sky is not None else np.ones((1, 1, 4), np.float32), dtype=np.float32)
        self.noise = np.ascontiguousarray(noise if noise is not None else np.array([[[128, 128, 128, 255]]], np.uint8), dtype=np.uint8)
        s = Scene()
        if self.voxels is not None:
            s.voxels = self.voxels.ctypes.data
        else:
            s.voxels = None
            s.brick_grid = self.grid.ctypes.data
            s.brick_pool = self.pool.ctypes.data if self.pool.size else None
        s.dims[:] = [W, H, D]
        s.palette = self.palette.ctypes.data
        s.sky = self.sky.ctypes.data; s.sky_w = self.sky.shape[1]; s.sky_h = self.sky.shape[0]
        s.noise = self.noise.ctypes.data; s.noise_w = self.noise.shape[1]; s.noise_h = self.noise.shape[0]
        self.c = s
        self.dims = (W, H, D)


def params_from(settings_c) -> Params:
    """Copy the shared fields of a product vrt_settings ctypes struct (or anything with the same names)."""
    p = Params()
    p.ao_samples = settings_c.ao_samples
    p.ambient_intensity = settings_c.ambient_intensity
    p.light_dir[:] = list(settings_c.light_dir)
    p.light_intensity = settings_c.light_intensity
    p.light_color[:] = list(settings_c.light_color)
    p.max_steps, p.ao_steps, p.max_bounces, p.shadows = settings_c.max_steps, settings_c.ao_steps, settings_c.max_bounces, settings_c.shadows
    return p


def push_from(push_c) -> Push:
    p = Push()
    C.memmove(C.byref(p), C.byref(push_c), C.sizeof(Push))
    return p


def render(scene: OracleScene, push, params: Params, planes=None, nthreads=1, rows=None):
    """Render with the oracle; returns dict name -> numpy array (H, W, ...)."""
    push = push_from(push)
    W, H = push.screen_size[0], push.screen_size[1]
    out, f = {}, Frame()
    for n, dt, tail in PLANES:
        if planes is None or n in planes:
            out[n] = np.zeros((H, W) + tail, dtype=dt)
            setattr(f, n, out[n].ctypes.data)
    if rows is not None:
        lib().vo_render_rows(C.byref(scene.c), C.byref(push), C.byref(params), C.byref(f), rows[0], rows[1])
    else:
        lib().vo_render_mt(C.byref(scene.c), C.byref(push), C.byref(params), C.byref(f), int(nthreads))
    return out


def render_band(scene: OracleScene, push, params: Params, row0, row1, planes=None, nthreads=8):
    """Rows [row0, row1) only, split over `nthreads` Python threads (ctypes releases the GIL).  Returns (row1-row0, W, ...)
    arrays: for full-size frames whose other rows are checked through size-independent properties."""
    import threading
    push = push_from(push)
    W, H = push.screen_size[0], push.screen_size[1]
    out, f = {}, Frame()
    for n, dt, tail in PLANES:
        if planes is None or n in planes:
            out[n] = np.zeros((H, W) + tail, dtype=dt)
            setattr(f, n, out[n].ctypes.data)
    edges = np.linspace(row0, row1, int(nthreads) + 1).astype(int)
    ts = [threading.Thread(target=lib().vo_render_rows, args=(C.byref(scene.c), C.byref(push), C.byref(params), C.byref(f), int(a), int(b)))
          for a, b in zip(edges[:-1], edges[1:]) if b > a]
    for t in ts: t.start()
    for t in ts: t.join()
    return {k: v[row0:row1] for k, v in out.items()}


def trace_ray(scene: OracleScene, start, direction, max_steps=512) -> Hit:
    h = Hit()
    s = (C.c_float * 3)(*[float(x) for x in start])
    d = (C.c_float * 3)(*[float(x) for x in direction])
    lib().vo_trace_ray(C.byref(scene.c), s, d, int(max_steps), C.byref(h))
    return h


def primary_ray(push, px, py):
    push = push_from(push)
    s, d = (C.c_float * 3)(), (C.c_float * 3)()
    lib().vo_primary_ray(C.byref(push), int(px), int(py), s, d)
    return np.array(list(s), np.float32), np.array(list(d), np.float32)


def denoise(color8, normal8, position, iterations=2, phi_color0=20.4, phi_normal0=1e-2, phi_pos0=1e-1,
            step_width0=2.0, mode=0):
    color8 = np.ascontiguousarray(color8, np.uint8); normal8 = np.ascontiguousarray(normal8, np.int8)
    position = np.ascontiguousarray(position, np.float32)
    H, W = color8.shape[:2]
    t0, t1 = np.zeros_like(color8), np.zeros_like(color8)
    last = lib().vo_denoise(color8.ctypes.data, normal8.ctypes.data, position.ctypes.data, t0.ctypes.data, t1.ctypes.data,
                            W, H, int(iterations), phi_color0, phi_normal0, phi_pos0, step_width0, int(mode))
    return color8 if last < 0 else (t0 if last == 0 else t1)


# ---- reference .vox parser (oracle/_ref, built from /root/reference in the build container) -------

_ref = None


def refvox():
    global _ref
    if _ref is None:
        if not os.path.exists(REFVOX_PATH):
            if os.path.exists("/root/reference/thirdparty/opengametools/include/ogt_vox.h"):
                subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
            else:
                return None
        l = C.CDLL(REFVOX_PATH)
        l.refvox_flatten.restype = C.c_int
        l.refvox_flatten.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p),
                                     C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        l.refvox_free.restype = None; l.refvox_free.argtypes = [C.c_void_p]
        l.refvox_write.restype = C.c_int
        _ref = l
    return _ref


def refvox_flatten(buf: bytes):
    """Flatten with the reference's own ogt_vox.h parser.  Returns (rc, voxels[z,y,x], palette(256,5), ninst, dropped)."""
    l = refvox()
    dims = (C.c_uint32 * 3)(); vox = C.c_void_p(); pal = np.zeros((256, 8), np.float32)
    ninst, dropped = C.c_uint32(), C.c_uint64()
    b = (C.c_uint8 * len(buf)).from_buffer_copy(buf)
    rc = l.refvox_flatten(b, len(buf), dims, C.byref(vox), pal.ctypes.data, C.byref(ninst), C.byref(dropped))
    if rc != 0:
        return rc, None, None, 0, 0
    W, H, D = dims[0], dims[1], dims[2]
    arr = np.ctypeslib.as_array(C.cast(vox, C.POINTER(C.c_uint8)), shape=(D * H * W,)).copy().reshape(D, H, W)
    l.refvox_free(vox)
    return 0, arr, pal[:, :5].copy(), ninst.value, dropped.value


def refvox_write(models, groups, instances, palette_rgba8, metal=None) -> bytes:
    """Serialise with the reference's ogt_vox_write_scene.
    models: list of uint8 arrays indexed [z,y,x] (ogt order x + y*sx + z*sx*sy);
    groups: list of (xform16, parent_index) with group 0 the root (parent 0xFFFFFFFF);
    instances: list of (model_index, group_index, xform16, hidden)."""
    l = refvox()
    nm = len(models)
    sizes = (C.c_uint32 * (3 * nm))()
    ptrs = (C.c_void_p * nm)()
    keep = []
    for i, m in enumerate(models):
        m = np.ascontiguousarray(m, np.uint8); keep.append(m)
        sizes[i * 3 + 0], sizes[i * 3 + 1], sizes[i * 3 + 2] = m.shape[2], m.shape[1], m.shape[0]
        ptrs[i] = m.ctypes.data
    ng = len(groups)
    gx = (C.c_float * (16 * ng))(*[float(v) for g in groups for v in g[0]])
    gp = (C.c_uint32 * ng)(*[int(g[1]) for g in groups])
    ni = len(instances)
    im = (C.c_uint32 * ni)(*[int(i[0]) for i in instances])
    ig = (C.c_uint32 * ni)(*[int(i[1]) for i in instances])
    ix = (C.c_float * (16 * ni))(*[float(v) for i in instances for v in i[2]])
    ih = (C.c_uint8 * ni)(*[1 if i[3] else 0 for i in instances])
    pal = np.ascontiguousarray(palette_rgba8, np.uint8)
    met = np.ascontiguousarray(metal if metal is not None else -np.ones(256), np.float32)
    out, n = C.c_void_p(), C.c_uint32()
    rc = l.refvox_write(nm, sizes, ptrs, ng, gx, gp, ni, im, ig, ix, ih, pal.ctypes.data_as(C.c_void_p),
                        met.ctypes.data_as(C.c_void_p), C.byref(out), C.byref(n))
    assert rc == 0
    data = C.string_at(out, n.value)
    l.refvox_free(out)
    return data
