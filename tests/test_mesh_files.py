"""collision-mesh files: the vertex loaders behind add_convex_collision_from_file (STL, OBJ, PLY -- the formats of the
reference's robot and object assets) and the 64-vertex hull budget."""
import struct

import numpy as np

from maniskill_amd.model import mesh


def _cloud(n=200, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.normal(size=(n, 3))
    return p / np.linalg.norm(p, axis=1, keepdims=True) * np.array([0.05, 0.03, 0.02])


def test_ply_ascii_and_binary_and_obj_give_the_same_vertices(tmp_path):
    v = _cloud(40).astype(np.float32)
    tris = [(0, 1, 2), (1, 2, 3)]
    a = tmp_path / "a.ply"
    with open(a, "w") as f:
        f.write(f"ply\nformat ascii 1.0\ncomment synthetic\nelement vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\n"
                f"property uchar red\nelement face {len(tris)}\nproperty list uchar int vertex_indices\nend_header\n")
        for p in v:
            f.write(f"{p[0]!r} {p[1]!r} {p[2]!r} 255\n".replace("np.float32(", "").replace(")", ""))
        for t in tris:
            f.write(f"3 {t[0]} {t[1]} {t[2]}\n")
    b = tmp_path / "b.ply"
    with open(b, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\n"
                 f"property float nx\nelement face {len(tris)}\nproperty list uchar int vertex_indices\nend_header\n").encode())
        for p in v:
            f.write(struct.pack("<4f", p[0], p[1], p[2], 0.0))
        for t in tris:
            f.write(struct.pack("<B3i", 3, *t))
    o = tmp_path / "c.obj"
    with open(o, "w") as f:
        f.write("# synthetic\n")
        for p in v:
            f.write(f"v {float(p[0])!r} {float(p[1])!r} {float(p[2])!r}\n")
        f.write("vn 0 0 1\nf 1 2 3\n")
    ref = np.unique(v.astype(np.float64), axis=0)
    for path in (a, b, o):
        got = mesh.load_mesh_vertices(str(path))
        assert got.shape == ref.shape and np.allclose(got, ref, atol=1e-7), path


def test_cooked_hull_respects_the_vertex_budget(tmp_path):
    v = _cloud(300, seed=1)
    o = tmp_path / "round.obj"
    with open(o, "w") as f:
        for p in v:
            f.write(f"v {p[0]} {p[1]} {p[2]}\n")
    hull = mesh.cook_convex_mesh(str(o), scale=(2.0, 1.0, 1.0))
    assert len(hull) <= 64 and len(hull) >= 32
    assert np.abs(hull[:, 0]).max() > 0.08  # scaled along x
    vol, com, I = mesh.hull_volume_com_inertia(hull)
    full = 4 / 3 * np.pi * 0.1 * 0.03 * 0.02
    assert 0.85 * full < vol < full and np.linalg.norm(com) < 2e-3 and np.all(np.linalg.eigvalsh(I) > 0)
