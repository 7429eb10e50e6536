#!/usr/bin/env python3
"""Register budget of the hand-scheduled kernels (csrc/vrt_device.hip), checked at build time.

K1's look-up loop pins physical registers and the kernel sits at two occupancy cliffs that the compiler's own remark does
not show: past 80 scalar registers a SIMD holds seven of these waves, not eight (measured: DESIGN.md 5, "Tile tags"), and
past 64 vector registers likewise.  `make -C voxel-raytracing_amd/csrc resources` (and tests/test_kernel_resources.py)
compile the device code with -Rpass-analysis=kernel-resource-usage and fail when a product kernel leaves its budget: one more
live scalar then breaks the build instead of silently costing 7 %.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "voxel-raytracing_amd", "csrc")

# mangled-name fragment -> (what it is, max VGPRs, max SGPRs, max scratch bytes per lane)
BUDGET = {
    "k_primaryILi7ELb0ELi1ELb0ELi0EE": ("K1 primary-only, look-up loop, one frame per launch (rows dealt to the XCDs)", 64, 80, 0),
    "k_primaryILi7ELb0ELi1ELb0ELi2EE": ("K1 primary-only, look-up loop, 8 frames in the kernel arguments (XCD regions)", 64, 80, 0),
    "k_primaryILi7ELb0ELi1ELb1ELi0EE": ("K1 primary-only, look-up loop, slots in the table, rows dealt to the XCDs (the bench line's kernel)", 64, 80, 0),
    "k_primaryILi7ELb0ELi1ELb1ELi2EE": ("K1 primary-only, look-up loop, slots in the table, XCD regions", 64, 80, 0),
    # (round 4: with the AO rays' pool the kernel needs 65 VGPRs, i.e. 7 waves per SIMD; forced into 64 it spills 12 B per lane and runs
    # the same 124-128 us on the reference defaults: tools/exp_r4_breakdown.py, libvrt_hip_m8.so -- no scratch is the invariant kept)
    "k_primaryILi7ELb0ELi4ELb0ELi0EE": ("megakernel without its bounce loop, one frame per launch", 72, 80, 0),
    "k_primaryILi7ELb0ELi4ELb1ELi2EE": ("megakernel without its bounce loop, table", 72, 96, 0),
    "k_primaryILi7ELb0ELi2ELb0ELi0EE": ("megakernel with the stack of hits (context option packed_bounces = 0), one frame per launch", 72, 96, 512),
    "k_primaryILi7ELb0ELi2ELb1ELi2EE": ("megakernel with the stack of hits, table", 72, 96, 512),
    "k_primaryILi7ELb0ELi5ELb0ELi0EE": ("megakernel, bounce chain as one word per hit, <= 2 bounces (BASELINE configs[3]), one frame per launch", 72, 96, 160),
    "k_primaryILi7ELb0ELi6ELb0ELi0EE": ("megakernel, bounce chain as one word per hit, <= 5 bounces (the reference's defaults), one frame per launch", 72, 96, 192),
    "k_primaryILi6ELb0ELi2ELb0ELi0EE": ("megakernel over bricks (config 5), one frame per launch", 80, 96, 576),
    "k_denoise_ldsILb0ELb0ELb0ELb1EE": ("K3 weighted pass, exact, packed", 128, 96, 0),
    # (the verified pass: four workgroups of four waves per compute unit, i.e. at most 128 VGPRs; 96 leaves it five)
    "k_denoise_verILb0ELb1ELi3ELb0EE": ("K3 verified weighted pass, tap offset 3 (the reference's pass 1)", 96, 96, 0),
    "k_denoise_verILb0ELb1ELi5ELb0EE": ("K3 verified weighted pass, tap offset 5 (the reference's pass 2)", 96, 96, 0),
    "k_denoise_verILb0ELb1ELi1ELb1EE": ("K3 verified pass 0", 64, 96, 0),
    # (round 4, every weight computed once: R waves per workgroup, five to six workgroups per compute unit by LDS: 96 VGPRs leave five waves per SIMD)
    "k_denoise_pairILb1ELi3EE": ("K3 verified weighted pass, every weight once, tap offset 3 (the reference's pass 1)", 96, 96, 0),
    "k_denoise_pairILb1ELi5EE": ("K3 verified weighted pass, every weight once, tap offset 5 (the reference's pass 2)", 96, 96, 0),
    "k_denoise_pairILb0ELi3EE": ("K3 VRT_DENOISE_FAST weighted pass, every weight once, tap offset 3", 96, 96, 0),
    "k_denoise_p0ILb1EE": ("K3 verified pass 0, a wave to itself (no LDS ring, no barrier)", 64, 96, 0),
}


def report():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed",
             "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", "vrt_device.hip"]
    p = subprocess.run([hipcc] + flags, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + p.stdout[-4000:])
    kernels, cur = {}, None
    for line in p.stdout.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return kernels


def check(kernels=None):
    kernels = kernels if kernels is not None else report()
    bad, rows = [], []
    for frag, (what, vmax, smax, scr) in BUDGET.items():
        hits = [(n, k) for n, k in kernels.items() if frag in n]
        if not hits:
            bad.append(f"{frag}: kernel not found in the resource report ({what})")
            continue
        for n, k in hits:
            rows.append((frag, k.get("VGPRs"), k.get("TotalSGPRs"), k.get("ScratchSize"), what))
            if k.get("VGPRs", 1 << 30) > vmax: bad.append(f"{frag}: {k.get('VGPRs')} VGPRs > {vmax} ({what})")
            if k.get("TotalSGPRs", 1 << 30) > smax: bad.append(f"{frag}: {k.get('TotalSGPRs')} SGPRs > {smax} ({what})")
            if k.get("ScratchSize", 1 << 30) > scr: bad.append(f"{frag}: {k.get('ScratchSize')} B scratch per lane > {scr} ({what})")
    return bad, rows


if __name__ == "__main__":
    bad, rows = check()
    for frag, v, s, sc, what in rows:
        print(f"{frag:36s} VGPR {v:3d}  SGPR {s:3d}  scratch {sc:4d}   {what}")
    if bad:
        print("\nregister budget exceeded:\n  " + "\n  ".join(bad), file=sys.stderr)
        sys.exit(1)
    print("register budgets hold")
