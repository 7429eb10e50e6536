"""AddressSanitizer + UBSan builds (CPU only) of the product's host-side parsers -- csrc/vox_reader.cpp and csrc/image_io.cpp --
driven over the committed fixtures and a few thousand random mutations of them (truncations, bit flips, wild 32-bit
fields): every input must end in a decoded scene / image or in an error code, never in a sanitizer report."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "voxel-raytracing_amd", "csrc")
SAN = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + CSRC]


def _build(tmp_path, name, sources):
    exe = str(tmp_path / name)
    r = subprocess.run(["g++"] + SAN + ["-o", exe, os.path.join(ROOT, "tests", "native", name + ".cpp")] + sources + ["-lz"],
                       capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("no sanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stderr
    return exe


def test_vox_reader_under_sanitizers(tmp_path):
    exe = _build(tmp_path, "vox_fuzz", [os.path.join(CSRC, "vox_reader.cpp"), os.path.join(CSRC, "image_io.cpp")])
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.vox")))
    r = subprocess.run([exe, "400"] + files, capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "parsed" in r.stdout and "ERROR" not in r.stderr


def test_image_decoders_under_sanitizers(tmp_path):
    exe = _build(tmp_path, "img_fuzz", [os.path.join(CSRC, "image_io.cpp")])
    scratch = tmp_path / "scratch"; scratch.mkdir()
    r = subprocess.run([exe, "600", str(scratch)], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "decoded" in r.stdout and "ERROR" not in r.stderr
