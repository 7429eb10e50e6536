"""The product's own .vox reader (csrc/vox_reader.cpp, via the host-only C-ABI entry vrt_vox_flatten_host)
against (a) the committed golden fixtures produced by the reference's ogt_vox.h and (b), when
oracle/_ref is present, the reference parser itself on freshly generated scenes."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLD, "vox_*.vox")))


def test_fixture_inventory():
    assert {"single", "multi", "groups", "raw_default_palette", "raw_imap_matl", "err_no_instance",
            "err_bad_magic", "err_version", "raw_graph_empty_model"} <= set(NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_reader_matches_golden(vrt, name):
    data = open(os.path.join(GOLD, f"vox_{name}.vox"), "rb").read()
    exp = np.load(os.path.join(GOLD, f"vox_{name}.npz"))
    rc = int(exp["rc"])
    if rc != 0:
        # reference: throws std::runtime_error (voxel_scene.cpp:46,50)
        with pytest.raises(RuntimeError) as ei:
            vrt.vox_flatten_host(data)
        want = "Could not parse voxel scene" if rc == 1 else "Voxel scene does not contain an instance."
        assert want in str(ei.value)
        return
    vox, pal, ninst, dropped = vrt.vox_flatten_host(data)
    assert vox.shape == exp["voxels"].shape
    assert (vox == exp["voxels"]).all()
    assert ninst == int(exp["ninst"]) and dropped == int(exp["dropped"])
    assert np.allclose(pal, exp["palette"], rtol=1e-6, atol=0)      # powf on the same libm: equal in practice


def test_reader_matches_reference_parser_live(vrt, oracle):
    if oracle.refvox() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference; the GPU box has only the prebuilt .so)")
    import sys
    sys.path.insert(0, GOLD)
    from make_vox_fixtures import xform, rnd_model
    rng = np.random.default_rng(99)
    perms = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((0, 1, 0), (1, 0, 0), (0, 0, 1)), ((0, 0, 1), (0, 1, 0), (1, 0, 0)),
             ((1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (0, 0, 1), (1, 0, 0)), ((0, 0, 1), (1, 0, 0), (0, 1, 0))]
    for case in range(25):
        nm = int(rng.integers(1, 4))
        models = [rnd_model(rng, *rng.integers(1, 9, 3), fill=float(rng.uniform(0.2, 0.9))) for _ in range(nm)]
        models = [m if m.any() else np.ones_like(m) for m in models]

        def rx():
            rows = np.array(perms[int(rng.integers(0, 6))]) * rng.choice([-1, 1], (3, 1))
            return xform(rows.tolist(), rng.integers(-12, 13, 3).tolist())

        ng = int(rng.integers(1, 4))
        groups = [(xform(), 0xFFFFFFFF)] + [(rx(), int(rng.integers(0, g))) for g in range(1, ng)]
        insts = [(int(rng.integers(0, nm)), int(rng.integers(0, ng)), rx(), bool(rng.integers(0, 2))) for _ in range(int(rng.integers(1, 6)))]
        pal = rng.integers(0, 256, (256, 4)).astype(np.uint8)
        met = np.where(rng.random(256) < 0.1, rng.random(256), -1).astype(np.float32)
        data = oracle.refvox_write(models, groups, insts, pal, met)
        rc, evox, epal, eninst, edrop = oracle.refvox_flatten(data)
        assert rc == 0
        vox, gpal, ninst, dropped = vrt.vox_flatten_host(data)
        assert vox.shape == evox.shape and (vox == evox).all(), f"case {case}"
        assert ninst == eninst and dropped == edrop
        assert np.allclose(gpal, epal, rtol=1e-6, atol=0)


def test_truncated_and_garbage_inputs_do_not_crash(vrt):
    data = open(os.path.join(GOLD, "vox_multi.vox"), "rb").read()
    rng = np.random.default_rng(4)
    for cut in list(range(0, 64)) + [len(data) // 2, len(data) - 1]:
        try:
            vrt.vox_flatten_host(data[:cut] if cut else b"\x00")
        except (RuntimeError, vrt.VrtError):
            pass
    for _ in range(50):
        b = bytearray(data)
        for _ in range(8):
            b[int(rng.integers(8, len(b)))] = int(rng.integers(0, 256))
        try:
            vrt.vox_flatten_host(bytes(b))
        except (RuntimeError, vrt.VrtError):
            pass
