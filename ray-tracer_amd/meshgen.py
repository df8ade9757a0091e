"""Seeded procedural stand-ins for the Stanford scans (no PLY models exist in this pipeline).

SURVEY.md 8(d): a cube-sphere grid of the needed resolution with a seeded multi-octave radial
displacement (surface-like coherence, as a scanned statue has), stretched per axis, scaled to the
model's span, and topped up to the exact face count.  Inputs only -- nothing here is on the hot path.
"""
import numpy as np

# name -> (faces, span of the longest axis, per-axis stretch, seed, orientation, fov, light z)
# counts: Stanford 3D Scanning Repository; cameras / lights: RayTracer.java:106-140
STANFORD = {
    "bunny": dict(faces=69451, span=0.156, stretch=(1.0, 0.99, 0.78), seed=1, orientation=(0, 0, 0), fov=15.0),
    "dragon": dict(faces=871414, span=0.205, stretch=(1.0, 0.70, 0.45), seed=2, orientation=(0, 0, 0), fov=23.0),
    "buddha": dict(faces=1087716, span=0.2, stretch=(0.41, 1.0, 0.41), seed=3, orientation=(180, 0, 0), fov=10.0),
    "thai": dict(faces=10000000, span=400.0, stretch=(0.59, 1.0, 0.52), seed=4, orientation=(0, 0, 0), fov=10.0),
    "lucy": dict(faces=28055742, span=1500.0, stretch=(0.58, 0.34, 1.0), seed=5, orientation=(0, -90, 0), fov=10.0),
}


def _cube_sphere(n):
    """Unit cube-sphere: each cube face is an n x n grid of quads split into two triangles, projected
    onto the sphere.  6 n^2 + 2 vertices, 12 n^2 faces, every vertex of valence 4 to 6 (scanned meshes
    have no high-valence poles; a lat-long sphere would put ~1000 slivers on each pole vertex)."""
    g = np.arange(n + 1, dtype=np.int64)
    U, V = np.meshgrid(g, g, indexing="ij")
    U, V = U.ravel(), V.ravel()
    Z, N = np.zeros_like(U), np.full_like(U, n)
    # integer cube coordinates of each face's grid, oriented so that (du x dv) points outward
    grids = [np.stack(c, axis=1) for c in ((N, U, V), (Z, V, U), (V, N, U), (U, Z, V), (U, V, N), (V, U, Z))]
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    ii, jj = ii.ravel(), jj.ravel()
    a, b, c, d = ii * (n + 1) + jj, (ii + 1) * (n + 1) + jj, (ii + 1) * (n + 1) + jj + 1, ii * (n + 1) + jj + 1
    quad = np.concatenate([np.stack([a, b, c], axis=1), np.stack([a, c, d], axis=1)])
    pts_i = np.concatenate(grids)
    faces = np.concatenate([quad + k * (n + 1) ** 2 for k in range(6)])
    # weld the shared edge / corner vertices of the six grids
    key = (pts_i[:, 0] * (n + 1) + pts_i[:, 1]) * (n + 1) + pts_i[:, 2]
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    pts = pts_i[first].astype(np.float64) / n * 2.0 - 1.0
    # equal-angle-ish warp before projection keeps triangle areas within ~2x of each other
    pts = np.tan(pts * (np.pi / 4.0))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    return pts, inv[faces]


def blob(n_faces, span=1.0, stretch=(1.0, 1.0, 1.0), seed=0, octaves=5, bumpiness=0.12, center=(0.0, 0.0, 0.0)):
    """A closed bumpy blob with exactly n_faces triangles: returns (vertices f32 [n,3], indices i32 [f,3])."""
    n_faces = int(n_faces)
    if n_faces < 12:
        raise ValueError("need at least 12 faces")
    n = max(1, int(np.sqrt(n_faces / 12.0)))
    unit, faces = _cube_sphere(n)
    rs = np.random.RandomState(seed)
    radius = np.ones(unit.shape[0])
    for k in range(octaves):
        freq = rs.randint(1, 4 + 3 * k, size=3).astype(np.float64)
        axes = rs.standard_normal((3, 3))
        axes /= np.linalg.norm(axes, axis=1, keepdims=True)
        phase = rs.uniform(0, 2 * np.pi, size=3)
        amp = bumpiness / (1.6 ** k)
        radius += amp * np.sin(freq[0] * np.pi * (unit @ axes[0]) + phase[0]) * np.cos(freq[1] * np.pi * (unit @ axes[1]) + phase[1]) \
            * np.cos(freq[2] * np.pi * (unit @ axes[2]) + phase[2])
    pts = unit * radius[:, None] * np.asarray(stretch, dtype=np.float64)

    # top up to the exact count: centroid splits (+2 faces, +1 vertex each), one edge split (+1) if odd
    missing = n_faces - faces.shape[0]
    if missing > 0:
        n_tri_split, odd = missing // 2, missing % 2
        pick = rs.permutation(faces.shape[0])[: n_tri_split + odd]
        nv = pts.shape[0]
        faces = faces.copy()
        tri = faces[pick[:n_tri_split]]
        cen = pts[tri].mean(axis=1)
        m = nv + np.arange(n_tri_split)
        extra = [np.stack([tri[:, 1], tri[:, 2], m], axis=1), np.stack([tri[:, 2], tri[:, 0], m], axis=1)]
        faces[pick[:n_tri_split], 2] = m
        new_pts = [cen]
        if odd:
            fa, fb, fc = faces[pick[-1]]
            new_pts.append(((pts[fb] + pts[fc]) / 2.0)[None, :])
            mid = nv + n_tri_split
            faces[pick[-1]] = (fa, fb, mid)
            extra.append(np.array([[fa, mid, fc]]))
        pts = np.concatenate([pts] + new_pts)
        faces = np.concatenate([faces] + extra)
    assert faces.shape[0] == n_faces, (faces.shape[0], n_faces)

    ext = pts.max(axis=0) - pts.min(axis=0)
    pts = (pts - (pts.max(axis=0) + pts.min(axis=0)) / 2.0) * (span / ext.max()) + np.asarray(center, dtype=np.float64)
    return np.ascontiguousarray(pts, dtype=np.float32), np.ascontiguousarray(faces, dtype=np.int32)


def stanford_standin(name, faces=None):
    """The blob standing in for one of the Stanford models (optionally at a reduced face count)."""
    cfg = STANFORD[name]
    return blob(faces or cfg["faces"], span=cfg["span"], stretch=cfg["stretch"], seed=cfg["seed"])


def soup(n_faces, span=1.0, tri_size=0.02, seed=0):
    """Uniform random triangle soup (the adversarial input): small triangles anywhere in a cube."""
    rs = np.random.RandomState(seed)
    c = rs.uniform(-0.5, 0.5, size=(n_faces, 1, 3)) * span
    v = c + rs.uniform(-0.5, 0.5, size=(n_faces, 3, 3)) * (tri_size * span)
    verts = np.ascontiguousarray(v.reshape(-1, 3), dtype=np.float32)
    idx = np.arange(3 * n_faces, dtype=np.int32).reshape(-1, 3)
    return verts, idx


def quad(y, half, center=(0.0, 0.0)):
    """A two-triangle horizontal floor (a Plane's stand-in, RayTracer.java:156-157), normal +y."""
    cx, cz = center
    verts = np.array([[cx - half, y, cz - half], [cx - half, y, cz + half], [cx + half, y, cz + half], [cx + half, y, cz - half]],
                     dtype=np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    return verts, idx


def write_ply(path, vertices, indices, normals=None, binary=False):
    """Writes a Stanford-style PLY (ASCII, or big-endian binary as the reference's readBinary expects)."""
    v = np.asarray(vertices, dtype=np.float32)
    i = np.asarray(indices, dtype=np.int32)
    props = ["x", "y", "z"] + (["nx", "ny", "nz"] if normals is not None else [])
    head = ["ply", "format binary_big_endian 1.0" if binary else "format ascii 1.0", "comment mi355rt test mesh",
            "element vertex %d" % v.shape[0]]
    head += ["property float %s" % p for p in props]
    head += ["element face %d" % i.shape[0], "property list uchar int vertex_indices", "end_header"]
    data = v if normals is None else np.concatenate([v, np.asarray(normals, dtype=np.float32)], axis=1)
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        if binary:
            f.write(data.astype(">f4").tobytes())
            rec = np.zeros(i.shape[0], dtype=[("n", "u1"), ("idx", ">i4", (3,))])
            rec["n"] = 3
            rec["idx"] = i
            f.write(rec.tobytes())
        else:
            lines = [" ".join(repr(float(x)) for x in row) + " " for row in data]
            lines += ["3 %d %d %d " % tuple(row) for row in i]
            f.write("\n".join(lines).encode())
