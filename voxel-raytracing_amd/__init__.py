"""voxel_raytracing_amd -- MI355X-native backend for the per-pixel voxel traversal + denoise hot path of
ectucker1/voxel-raytracing (shader/voxel_volume.frag + shader/denoiser.frag), behind the reference's
.vox-load -> render-to-RGBA call surface.  HIP kernels + C-ABI: csrc/ (libvrt_hip.so, include/vrt.h);
this package is the Python host side (ctypes) mirroring the reference's objects."""
from . import _capi
from ._capi import (TRAVERSAL_AUTO, TRAVERSAL_DENSE, TRAVERSAL_BITMASK, TRAVERSAL_JUMP, TRAVERSAL_DF, TRAVERSAL_DFJ,
                    DENOISE_CANONICAL, DENOISE_AS_SHIPPED, DENOISE_FAST, VrtError, lib)
from .host import (AmbientOcclusionSettings, BlitStage, CameraController, CameraKey, DenoiserSettings, DenoiserStage, Engine,
                   FsrScaling, FsrSettings, GeometryBuffer, GeometryStage, LightSettings, TraceSettings,
                   UpscalerStage, camera_path,                   VoxelRenderSettings, VoxelRenderer, VoxelScene, load_image, make_push, make_shard, vox_flatten_host,
                   write_image)
from . import synthetic
from . import distributed

__all__ = [n for n in dir() if not n.startswith("_")]
