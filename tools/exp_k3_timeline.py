"""Where a weighted denoiser pass spends its 27 us (development build: make -C voxel-raytracing_amd/csrc variant NAME=k3st
EXTRA=-DVRT_K3_STAMPS; VRT_LIB=.../libvrt_hip_k3st.so): every workgroup of k_denoise_ver / k_denoise_pair leaves its start / ring filled / rows done /
end stamps (100 MHz) in the first words of the output image."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import voxel_raytracing_amd as vrt
from helpers import camera_push
eng = vrt.Engine(0)
pal = vrt.synthetic.default_palette(metallic_ids=range(200, 256))
sc = vrt.VoxelScene.from_dense(eng, vrt.synthetic.treehouse(256, seed=2), pal, sky=vrt.synthetic.sky_gradient(512, 256), noise=vrt.synthetic.blue_noise_standin(512))
res = (1920, 1080)
st = vrt.VoxelRenderSettings(targetResolution=res); st.fsrSetttings.enable = False; st.occlusionSettings.numSamples = 0
gb = vrt.GeometryStage(eng, st, sc).record(camera_push(vrt, (256, 256, 256), res)); eng.synchronize()
st.denoiserSettings.iterations = 2
den = vrt.DenoiserStage(eng, st)
for rep in range(4):
    out = den.record(gb.color, gb.normal, gb.position); eng.synchronize()
gx = int(sys.argv[1]) if len(sys.argv) > 1 else 30          # strips and segments of the launch (k_denoise_ver: 30 x 34; k_denoise_pair at 1080p: 34 x 40)
gy = int(sys.argv[2]) if len(sys.argv) > 2 else 34
n = gx * gy
w = out.cpu().numpy().view(np.uint32).reshape(-1)[: 4 * n].reshape(n, 4).astype(np.int64)
s0 = w[:, 0]; t0 = s0[np.abs(s0 - np.median(s0)) < 5000].min()      # (a stamp overwritten by a pixel of the first rows is far off: not the launch's start)
w = (w - t0) / 100.0                                       # us
good = (w[:, 3] > 0) & (w[:, 3] < 1000)
print("workgroups", n, "kernel span %.1f us" % (w[good, 3].max()))
for name, col in (("start", 0), ("ring filled", 1), ("rows done", 2), ("end", 3)):
    v = np.sort(w[:, col]); print(f"{name:12s}: min {v[0]:.1f}  10% {v[n//10]:.1f}  median {v[n//2]:.1f}  90% {v[9*n//10]:.1f}  max {v[-1]:.1f}")
print("prologue (ring filled - start): median %.1f max %.1f;  rows: median %.1f max %.1f;  tail: median %.1f max %.1f" % (
      np.median(w[:, 1] - w[:, 0]), (w[:, 1] - w[:, 0]).max(), np.median(w[:, 2] - w[:, 1]), (w[:, 2] - w[:, 1]).max(), np.median(w[:, 3] - w[:, 2]), (w[:, 3] - w[:, 2]).max()))

tail = np.sort((w[:, 3] - w[:, 2])[(w[:, 3] - w[:, 2] >= 0) & (w[:, 3] - w[:, 2] < 100)]); m = len(tail)
print("tail (end - rows done): 50%% %.1f 70%% %.1f 90%% %.1f 99%% %.1f max %.1f; over 1 us: %d of %d" % (tail[m // 2], tail[7 * m // 10], tail[9 * m // 10], tail[99 * m // 100], tail[-1], int((tail > 1.0).sum()), m))
pro = np.sort((w[:, 1] - w[:, 0])[(w[:, 1] - w[:, 0] >= 0) & (w[:, 1] - w[:, 0] < 100)]); m = len(pro)
print("ring fill (+ in-loop redo rounds of k_denoise_pair): 50%% %.1f 70%% %.1f 90%% %.1f 99%% %.1f max %.1f" % (pro[m // 2], pro[7 * m // 10], pro[9 * m // 10], pro[99 * m // 100], pro[-1]))
ok = np.abs(w[:, 0] - np.median(w[:, 0])) < 2.0              # (stamps of a few workgroups are overwritten by pixels of the first rows)
ok &= (w[:, 3] > w[:, 0]) & (w[:, 3] < w[:, 0] + 100)
rows = w[:, 2] - w[:, 1]; life = w[:, 3] - w[:, 0]
ids = np.arange(n); bx, by = ids % gx, ids // gx
print("usable", int(ok.sum()), "of", n)
print("life by wg_id % 8 (XCD):", [round(float(np.median(life[ok & (ids % 8 == k)])), 1) for k in range(8)])
print("life by segment row:", [round(float(np.median(life[ok & (by == k)])), 1) for k in range(gy)])
print("life by strip:", [round(float(np.median(life[ok & (bx == k)])), 1) for k in range(gx)])
print("life by (wg_id // 8) % 32 (CU within XCD?):", [round(float(np.median(life[ok & ((ids // 8) % 32 == k)])), 1) for k in range(32)])
