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


def test_convex_decomposition_file_gives_one_hull_per_part(tmp_path):
    """`add_multiple_convex_collisions_from_file` (actor_builder.py:121-135): an L-shaped bracket written as two boxes in one
    OBJ (one `o` group per part, the layout V-HACD / CoACD write) becomes two convex shapes whose union has the bracket's
    mass -- one hull over all vertices would fill the notch"""
    import itertools

    from maniskill_amd.model.compile import ShapeRecord

    def box(lo, hi):
        return np.array(list(itertools.product(*zip(lo, hi))), dtype=np.float64)

    parts = [box((0, 0, 0), (0.1, 0.02, 0.02)), box((0, 0, 0.02), (0.02, 0.02, 0.1))]
    faces = [(0, 1, 3), (0, 3, 2), (4, 5, 7), (4, 7, 6), (0, 1, 5), (0, 5, 4), (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 3, 7), (1, 7, 5)]
    f = tmp_path / "bracket.obj"
    with open(f, "w") as fh:
        base = 0
        for k, v in enumerate(parts):
            fh.write(f"o part{k}\n")
            for p in v:
                fh.write(f"v {p[0]} {p[1]} {p[2]}\n")
            for t in faces:
                fh.write(f"f {t[0] + 1 + base} {t[1] + 1 + base} {t[2] + 1 + base}\n")
            base += len(v)
    hulls = mesh.cook_convex_parts(str(f))
    assert len(hulls) == 2 and all(len(h) == 8 for h in hulls)
    vol = sum(ShapeRecord("convex", vertices=h, density=1.0).mass_properties()[0] for h in hulls)
    assert abs(vol - (0.1 * 0.02 * 0.02 + 0.02 * 0.02 * 0.08)) < 1e-9
    one = mesh.cook_convex_mesh(str(f))
    assert ShapeRecord("convex", vertices=one, density=1.0).mass_properties()[0] > 1.5 * vol

    from maniskill_amd.physx.components import PhysxCollisionShapeConvexMesh

    shapes = PhysxCollisionShapeConvexMesh.load_multiple(str(f), scale=(2, 2, 2))
    assert len(shapes) == 2 and abs(shapes[0].vertices[:, 0].max() - 0.2) < 1e-12
