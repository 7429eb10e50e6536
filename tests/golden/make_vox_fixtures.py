"""Generate the .vox golden fixtures (run in the build container, where /root/reference exists).

* The structured files are written with the reference's own ogt_vox_write_scene (through oracle/_ref);
  the raw files are assembled chunk by chunk here to cover what the writer never emits (no scene graph,
  default palette, IMAP, legacy MATT, version 200, errors).
* The expected flattened volume / palette of every file comes from the reference's ogt_vox_read_scene plus
  the restated voxel_scene.cpp flatten (oracle/ref_vox/ref_vox.cpp).
Outputs: tests/golden/vox_<name>.vox (input bytes) and vox_<name>.npz (rc, voxels, palette, instances, dropped).
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle  # noqa: E402


def xform(rows=((1, 0, 0), (0, 1, 0), (0, 0, 1)), t=(0, 0, 0)):
    r = np.array(rows, dtype=np.float32)
    # ogt_vox_transform: column i = (m_i0, m_i1, m_i2, m_i3)
    return [r[0][0], r[1][0], r[2][0], 0, r[0][1], r[1][1], r[2][1], 0, r[0][2], r[1][2], r[2][2], 0, t[0], t[1], t[2], 1]


def rnd_model(rng, sx, sy, sz, fill=0.3):
    m = (rng.random((sz, sy, sx)) < fill) * rng.integers(1, 256, (sz, sy, sx))
    return m.astype(np.uint8)


def chunk(cid, body=b"", children=b""):
    return cid + struct.pack("<II", len(body), len(children)) + body + children


def vdict(d):
    out = struct.pack("<I", len(d))
    for k, v in d.items():
        out += struct.pack("<I", len(k)) + k.encode() + struct.pack("<I", len(v)) + v.encode()
    return out


def xyzi(model):
    zs, ys, xs = np.nonzero(model)
    body = struct.pack("<I", len(xs))
    for x, y, z in zip(xs, ys, zs):
        body += bytes([x, y, z, model[z, y, x]])
    return chunk(b"SIZE", struct.pack("<III", model.shape[2], model.shape[1], model.shape[0])) + chunk(b"XYZI", body)


def raw_file(children, version=150):
    return b"VOX " + struct.pack("<I", version) + chunk(b"MAIN", b"", children)


def build():
    rng = np.random.default_rng(20261003)
    pal = rng.integers(0, 256, (256, 4)).astype(np.uint8); pal[:, 3] = 255; pal[0] = 0
    ROOT = 0xFFFFFFFF
    fx = {}
    # 1. one model, identity
    fx["single"] = oracle.refvox_write([rnd_model(rng, 5, 4, 3)], [(xform(), ROOT)], [(0, 0, xform(), False)], pal)
    # 2. several instances: rotations with negative signs, negative translations, overlap, hidden, shared model
    m0, m1, m2 = rnd_model(rng, 6, 3, 4), rnd_model(rng, 3, 7, 2), rnd_model(rng, 4, 4, 4, 0.6)
    insts = [
        (0, 0, xform(t=(0, 0, 0)), False),
        (1, 0, xform(((0, 1, 0), (-1, 0, 0), (0, 0, 1)), (5, -3, 2)), False),
        (2, 0, xform(((0, 0, -1), (0, 1, 0), (1, 0, 0)), (-4, 6, -1)), True),
        (0, 0, xform(((-1, 0, 0), (0, -1, 0), (0, 0, 1)), (2, 2, 5)), False),
        (1, 0, xform(((1, 0, 0), (0, 0, -1), (0, 1, 0)), (-7, -2, 3)), False),
    ]
    met = -np.ones(256, np.float32); met[3] = 0.5; met[200] = 1.0; met[17] = 0.25
    fx["multi"] = oracle.refvox_write([m0, m1, m2], [(xform(), ROOT)], insts, pal, met)
    # 3. nested groups with their own transforms
    groups = [(xform(), ROOT), (xform(((0, -1, 0), (1, 0, 0), (0, 0, 1)), (10, 0, 0)), 0),
              (xform(((1, 0, 0), (0, 0, 1), (0, -1, 0)), (0, -6, 4)), 1)]
    insts = [(0, 1, xform(t=(1, 2, 3)), False), (1, 2, xform(((0, 1, 0), (1, 0, 0), (0, 0, -1)), (-2, 0, 1)), False),
             (2, 0, xform(t=(-9, -9, -9)), False)]
    fx["groups"] = oracle.refvox_write([m0, m1, m2], groups, insts, pal)
    # 4. raw: no scene graph, no RGBA chunk (default palette), version 200
    fx["raw_default_palette"] = raw_file(xyzi(rnd_model(rng, 7, 5, 6)), version=200)
    # 5. raw: IMAP + MATL + legacy MATT, no scene graph
    imap = rng.permutation(256).astype(np.uint8).tobytes()
    body = xyzi(rnd_model(rng, 4, 4, 4, 0.5)) + chunk(b"RGBA", pal.tobytes()) + chunk(b"IMAP", imap)
    body += chunk(b"MATL", struct.pack("<i", 9) + vdict({"_type": "_metal", "_metal": "0.75", "_rough": "0.1"}))
    body += chunk(b"MATL", struct.pack("<i", 256) + vdict({"_type": "_metal", "_metal": "0.3"}))
    body += chunk(b"MATT", struct.pack("<iifI", 40, 1, 0.6, 0))
    body += chunk(b"MATT", struct.pack("<iifI", 41, 2, 0.9, 0))
    body += chunk(b"rOBJ", vdict({"_type": "_bloom"})) + chunk(b"NOTE", b"\x00" * 12)
    fx["raw_imap_matl"] = raw_file(body)
    # 6. errors: two models without a scene graph -> no instance; bad magic; unsupported version; empty file
    fx["err_no_instance"] = raw_file(xyzi(rnd_model(rng, 2, 2, 2, 0.9)) + xyzi(rnd_model(rng, 3, 2, 2, 0.9)))
    fx["err_bad_magic"] = b"VOY " + raw_file(b"")[4:]
    fx["err_version"] = raw_file(xyzi(rnd_model(rng, 2, 2, 2, 0.9)), version=151)
    # 7. an empty model (XYZI with 0 voxels) referenced by a shape node is skipped
    mE = np.zeros((2, 2, 2), np.uint8)
    kids = struct.pack("<I", 0) + vdict({}) + struct.pack("<III", 1, 0xFFFFFFFF, 0) + struct.pack("<I", 1) + vdict({})
    graph = chunk(b"nTRN", kids)
    graph += chunk(b"nGRP", struct.pack("<I", 1) + vdict({}) + struct.pack("<III", 2, 2, 4))
    graph += chunk(b"nTRN", struct.pack("<I", 2) + vdict({"_name": "a"}) + struct.pack("<IIII", 3, 0xFFFFFFFF, 0, 1) + vdict({"_t": "3 -2 1", "_r": "40"}))
    graph += chunk(b"nSHP", struct.pack("<I", 3) + vdict({}) + struct.pack("<I", 1) + struct.pack("<I", 0) + vdict({}))
    graph += chunk(b"nTRN", struct.pack("<I", 4) + vdict({"_hidden": "1"}) + struct.pack("<IIII", 5, 0xFFFFFFFF, 0, 1) + vdict({"_t": "0 0 7"}))
    graph += chunk(b"nSHP", struct.pack("<I", 5) + vdict({}) + struct.pack("<I", 1) + struct.pack("<I", 1) + vdict({}))
    fx["raw_graph_empty_model"] = raw_file(xyzi(rnd_model(rng, 3, 4, 5, 0.5)) + xyzi(mE) + graph + chunk(b"RGBA", pal.tobytes()))
    return fx


def main():
    for name, data in build().items():
        rc, vox, palette, ninst, dropped = oracle.refvox_flatten(data)
        open(os.path.join(HERE, f"vox_{name}.vox"), "wb").write(data)
        if rc == 0:
            np.savez_compressed(os.path.join(HERE, f"vox_{name}.npz"), rc=rc, voxels=vox, palette=palette, ninst=ninst, dropped=dropped)
        else:
            np.savez_compressed(os.path.join(HERE, f"vox_{name}.npz"), rc=rc)
        print(name, "rc", rc, None if vox is None else vox.shape, "instances", ninst, "dropped", dropped, len(data), "bytes")


if __name__ == "__main__":
    main()
