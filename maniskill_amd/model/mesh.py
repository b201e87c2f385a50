"""Collision-mesh cooking on the host: STL / OBJ / PLY vertices -> convex hull with a vertex budget.

Counterpart of PhysX convex-mesh cooking that SAPIEN runs inside
`add_convex_collision_from_file` (mani_skill/utils/building/actor_builder.py:113-131 calls
`PhysxCollisionShapeConvexMesh`). PhysX GPU-compatible hulls are limited to 64 vertices; the
same budget is applied here (MSSIM_MAX_HULL_VERTS).
"""
import struct
from functools import lru_cache

import numpy as np
from scipy.spatial import ConvexHull


def load_stl_vertices(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        data = f.read()
    is_ascii = data[:5].lower() == b"solid" and b"facet" in data[:1024]
    if is_ascii:
        verts = []
        for line in data.decode("ascii", errors="ignore").splitlines():
            s = line.strip().split()
            if len(s) == 4 and s[0] == "vertex":
                verts.append([float(s[1]), float(s[2]), float(s[3])])
        return np.unique(np.array(verts, dtype=np.float64), axis=0)
    (n,) = struct.unpack("<I", data[80:84])
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    arr = np.frombuffer(data[84 : 84 + 50 * n], dtype=rec)
    return np.unique(arr["v"].reshape(-1, 3).astype(np.float64), axis=0)


def load_obj_vertices(path: str) -> np.ndarray:
    verts = []
    with open(path, "r", errors="ignore") as f:
        for line in f:
            if line.startswith("v "):
                t = line.split()
                verts.append([float(t[1]), float(t[2]), float(t[3])])
    return np.unique(np.array(verts, dtype=np.float64).reshape(-1, 3), axis=0)


def load_obj_parts(path: str):
    """the convex parts of a decomposition file (V-HACD / CoACD style OBJ: one `o` / `g` group per part): list of vertex
    arrays, each the vertices its group's faces use (all of the group's `v` lines when it has no faces)"""
    verts, parts, cur = [], [], None
    with open(path, "r", errors="ignore") as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append([float(t[1]), float(t[2]), float(t[3])])
                if cur is not None:
                    cur["own"].append(len(verts) - 1)
            elif t[0] in ("o", "g"):
                cur = dict(own=[], used=set())
                parts.append(cur)
            elif t[0] == "f":
                if cur is None:
                    cur = dict(own=[], used=set())
                    parts.append(cur)
                for w in t[1:]:
                    i = int(w.split("/")[0])
                    cur["used"].add(i - 1 if i > 0 else len(verts) + i)
    V = np.array(verts, dtype=np.float64).reshape(-1, 3)
    out = []
    for part in parts:
        idx = sorted(part["used"]) if part["used"] else part["own"]
        if len(idx) >= 4:
            out.append(np.unique(V[idx], axis=0))
    return out if out else [np.unique(V, axis=0)]


def cook_convex_parts(path: str, scale=(1.0, 1.0, 1.0), max_verts: int = 64):
    """one cooked hull per convex part of the file (`add_multiple_convex_collisions_from_file`)"""
    if str(path).lower().endswith(".obj"):
        parts = load_obj_parts(path)
    else:
        parts = [load_mesh_vertices(path)]
    return [simplify_hull(v * np.array(scale), max_verts) for v in parts]


def load_ply_vertices(path: str) -> np.ndarray:
    """vertex positions of an ASCII or binary PLY file (the YCB `collision.ply` format): the x, y, z properties of the
    `vertex` element, whatever else it carries"""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header") + len(b"end_header")
    end = data.index(b"\n", end) + 1
    header = data[:end].decode("ascii", errors="ignore").splitlines()
    fmt = next(l.split()[1] for l in header if l.startswith("format"))
    types = dict(char="i1", uchar="u1", short="i2", ushort="u2", int="i4", uint="u4", float="f4", double="f8",
                 int8="i1", uint8="u1", int16="i2", uint16="u2", int32="i4", uint32="u4", float32="f4", float64="f8")
    n_vert, props, in_vertex, before = 0, [], False, 0
    for l in header:
        t = l.split()
        if not t:
            continue
        if t[0] == "element":
            in_vertex = t[1] == "vertex"
            if in_vertex:
                n_vert = int(t[2])
            elif not props:
                before += 1  # (an element in front of the vertices: not produced by any exporter we know)
        elif t[0] == "property" and in_vertex:
            if t[1] == "list":
                raise ValueError(f"{path}: list property inside the vertex element")
            props.append((t[2], types[t[1]]))
    if before:
        raise ValueError(f"{path}: elements in front of the vertex element are not supported")
    names = [p[0] for p in props]
    if fmt == "ascii":
        rows = np.array([[float(x) for x in l.split()] for l in data[end:].decode("ascii", errors="ignore").splitlines()[:n_vert]])
        v = rows[:, [names.index("x"), names.index("y"), names.index("z")]]
    else:
        order = "<" if fmt == "binary_little_endian" else ">"
        rec = np.dtype([(n_, order + t_) for n_, t_ in props])
        arr = np.frombuffer(data[end : end + rec.itemsize * n_vert], dtype=rec)
        v = np.stack([arr["x"], arr["y"], arr["z"]], axis=1)
    return np.unique(np.asarray(v, dtype=np.float64), axis=0)


def load_mesh_vertices(path: str) -> np.ndarray:
    ext = str(path).lower().rsplit(".", 1)[-1]
    if ext == "obj":
        return load_obj_vertices(path)
    if ext == "ply":
        return load_ply_vertices(path)
    return load_stl_vertices(path)


def hull_volume_com_inertia(verts: np.ndarray):
    """Volume, centre of mass and unit-density inertia about the COM of conv(verts)."""
    hull = ConvexHull(verts)
    c0 = verts[hull.vertices].mean(0)
    vol = 0.0
    com = np.zeros(3)
    # covariance-style second moments integrated over tetrahedra (c0, a, b, c)
    C = np.zeros((3, 3))
    canon = np.full((3, 3), 1 / 120.0) + np.eye(3) / 120.0  # integral of x_i x_j over unit tet
    for tri, eq in zip(hull.simplices, hull.equations):
        a, b, c = verts[tri] - c0
        if np.dot(np.cross(b - a, c - a), eq[:3]) < 0:
            b, c = c, b
        A = np.stack([a, b, c], axis=1)
        det = np.linalg.det(A)
        vol += det / 6.0
        com += det / 24.0 * (a + b + c)
        C += det * A @ canon @ A.T
    com = com / vol
    # second moment about c0 -> about com
    C = C - vol * np.outer(com, com)
    I = np.trace(C) * np.eye(3) - C
    return vol, com + c0, I


def simplify_hull(verts: np.ndarray, max_verts: int) -> np.ndarray:
    """Reduce a convex point set to at most `max_verts` hull vertices.

    Greedy: start from the 6 axis-extreme points, then repeatedly add the input vertex that is
    farthest outside the current hull (largest plane violation). Deterministic.
    """
    hull = ConvexHull(verts)
    pts = verts[hull.vertices]
    if len(pts) <= max_verts:
        return pts
    chosen = []
    for ax in range(3):
        for idx in (int(np.argmin(pts[:, ax])), int(np.argmax(pts[:, ax]))):
            if idx not in chosen:
                chosen.append(idx)
    # make sure the seed is full-dimensional
    while len(chosen) < 4:
        chosen.append(next(i for i in range(len(pts)) if i not in chosen))
    while len(chosen) < max_verts:
        try:
            h = ConvexHull(pts[chosen], qhull_options="QJ")
        except Exception:
            rest = [i for i in range(len(pts)) if i not in chosen]
            chosen.append(rest[0])
            continue
        # distance of every point outside the current hull
        d = (pts @ h.equations[:, :3].T + h.equations[:, 3]).max(axis=1)
        d[chosen] = -np.inf
        i = int(np.argmax(d))
        if d[i] <= 1e-9:
            break
        chosen.append(i)
    return pts[sorted(chosen)]


@lru_cache(maxsize=64)
def _cook_cached(path: str, scale: tuple, max_verts: int):
    v = load_mesh_vertices(path) * np.array(scale)
    return simplify_hull(v, max_verts)


def cook_convex_mesh(path: str, scale=(1.0, 1.0, 1.0), max_verts: int = 64) -> np.ndarray:
    return _cook_cached(path, tuple(float(s) for s in scale), int(max_verts)).copy()


def bounding_sphere(verts: np.ndarray):
    lo, hi = verts.min(0), verts.max(0)
    c = 0.5 * (lo + hi)
    r = float(np.linalg.norm(verts - c, axis=1).max())
    return c, r


# ---------------------------------------------------------------------------------------------- triangle meshes
def load_triangles(path: str):
    """(vertices [V,3], triangles [T,3]) of a mesh file: OBJ (polygons are fanned) or STL (every facet)"""
    ext = str(path).lower().rsplit(".", 1)[-1]
    if ext == "obj":
        verts, tris = [], []
        with open(path, "r", errors="ignore") as f:
            for line in f:
                t = line.split()
                if not t:
                    continue
                if t[0] == "v":
                    verts.append([float(t[1]), float(t[2]), float(t[3])])
                elif t[0] == "f":
                    idx = [int(w.split("/")[0]) for w in t[1:]]
                    idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                    for k in range(1, len(idx) - 1):
                        tris.append([idx[0], idx[k], idx[k + 1]])
        return np.asarray(verts, dtype=np.float64).reshape(-1, 3), np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    with open(path, "rb") as f:
        data = f.read()
    if data[:5].lower() == b"solid" and b"facet" in data[:1024]:
        pts = [[float(x) for x in l.split()[1:4]] for l in data.decode("ascii", errors="ignore").splitlines() if l.strip().startswith("vertex")]
        v = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    else:
        (n,) = struct.unpack("<I", data[80:84])
        rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
        v = np.frombuffer(data[84 : 84 + 50 * n], dtype=rec)["v"].reshape(-1, 3).astype(np.float64)
    return v, np.arange(len(v), dtype=np.int64).reshape(-1, 3)


def triangle_soup(verts: np.ndarray, tris: np.ndarray) -> np.ndarray:
    """[T, 12] float32 per triangle: centroid, then the three corners relative to it (a triangle is handed to the
    narrowphase as a 3-vertex hull in a frame at its centroid). Degenerate triangles are dropped."""
    P = np.asarray(verts, dtype=np.float64)[np.asarray(tris, dtype=np.int64)]  # [T,3,3]
    area2 = np.linalg.norm(np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]), axis=1)
    P = P[area2 > 1e-14]
    c = P.mean(axis=1)
    return np.concatenate([c, (P - c[:, None, :]).reshape(-1, 9)], axis=1).astype(np.float32)


def build_bvh16(soup: np.ndarray) -> np.ndarray:
    """16-wide bounding-volume hierarchy over the triangles of `soup`: [n_nodes, 112] float32, node 0 the root. A node
    holds 16 children, one per lane of the 16-lane group that traverses it: child c's box in words 6c .. 6c+5 (min xyz,
    max xyz; an empty child has min > max) and its reference in word 96 + c as an int32 bit pattern -- >= 0: node index,
    < 0: ~triangle index (a leaf child is ONE triangle, its box the triangle's own)."""
    T = len(soup)
    cen = soup[:, :3].astype(np.float64)
    corners = (soup[:, None, :3] + soup[:, 3:].reshape(T, 3, 3)).astype(np.float32)  # f32 corners as the kernels see them
    tmin, tmax = corners.min(axis=1), corners.max(axis=1)
    nodes = []

    def split(ids, parts):
        """ids -> `parts` groups of nearly equal size by recursive median splits along the longest axis"""
        if parts == 1 or len(ids) <= 1:
            return [ids]
        ext = cen[ids].max(axis=0) - cen[ids].min(axis=0)
        order = ids[np.argsort(cen[ids, int(np.argmax(ext))], kind="stable")]
        half = len(order) // 2
        return split(order[:half], parts // 2) + split(order[half:], parts - parts // 2)

    def make(ids):
        me = len(nodes)
        node = np.zeros(112, dtype=np.float32)
        node[:96].reshape(16, 6)[:] = [1, 1, 1, -1, -1, -1]
        refs = np.full(16, -2**31, dtype=np.int32)
        nodes.append(node)
        # (few, full leaves: a set of up to 256 triangles is cut into ceil(n / 16) groups of at most 16)
        groups = [ids[k : k + 1] for k in range(len(ids))] if len(ids) <= 16 else [g for g in split(ids, min(16, -(-len(ids) // 16))) if len(g)]
        for k, g in enumerate(groups):
            node[6 * k : 6 * k + 3] = tmin[g].min(axis=0)
            node[6 * k + 3 : 6 * k + 6] = tmax[g].max(axis=0)
            refs[k] = ~int(g[0]) if len(g) == 1 else make(g)
        node[96:112] = refs.view(np.float32)
        return me

    if T:
        make(np.arange(T))
    return np.asarray(nodes, dtype=np.float32).reshape(-1, 112)
