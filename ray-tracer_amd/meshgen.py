"""Seeded procedural stand-ins for the Stanford scans (no PLY models exist in this pipeline).

SURVEY.md 8(d): a lat-long sphere of the needed resolution with a seeded multi-octave radial
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


def _uv_sphere(rings, segs):
    """Unit lat-long sphere: 2 poles + (rings-1)*segs vertices, 2*segs*(rings-1) faces (CCW outward)."""
    theta = np.pi * np.arange(1, rings, dtype=np.float64) / rings        # polar angle of each ring
    phi = 2.0 * np.pi * np.arange(segs, dtype=np.float64) / segs
    T, P = np.meshgrid(theta, phi, indexing="ij")
    ang = np.concatenate([[[0.0, 0.0]], np.stack([T.ravel(), P.ravel()], axis=1), [[np.pi, 0.0]]])
    south = ang.shape[0] - 1

    def vid(r, s):   # ring r in [0, rings-2], segment s
        return 1 + r * segs + (s % segs)

    s = np.arange(segs)
    faces = [np.stack([np.zeros(segs, dtype=np.int64), vid(0, s), vid(0, s + 1)], axis=1)]
    for r in range(rings - 2):
        a, b, c, d = vid(r, s), vid(r + 1, s), vid(r + 1, s + 1), vid(r, s + 1)
        faces.append(np.stack([a, b, c], axis=1))
        faces.append(np.stack([a, c, d], axis=1))
    faces.append(np.stack([np.full(segs, south, dtype=np.int64), vid(rings - 2, s + 1), vid(rings - 2, s)], axis=1))
    return ang, np.concatenate(faces)


def blob(n_faces, span=1.0, stretch=(1.0, 1.0, 1.0), seed=0, octaves=5, bumpiness=0.12, center=(0.0, 0.0, 0.0)):
    """A closed bumpy blob with exactly n_faces triangles: returns (vertices f32 [n,3], indices i32 [f,3])."""
    n_faces = int(n_faces)
    if n_faces < 8:
        raise ValueError("need at least 8 faces")
    rings = max(3, int(np.sqrt(n_faces / 4.0)))
    segs = max(3, (n_faces // (2 * (rings - 1))))
    while 2 * segs * (rings - 1) > n_faces:
        segs -= 1
    ang, faces = _uv_sphere(rings, segs)
    rs = np.random.RandomState(seed)
    radius = np.ones(ang.shape[0])
    for k in range(octaves):
        f1, f2 = rs.randint(1, 4 + 3 * k, size=2)
        p1, p2 = rs.uniform(0, 2 * np.pi, size=2)
        amp = bumpiness / (1.6 ** k)
        radius += amp * np.sin(f1 * ang[:, 0] + p1) * np.cos(f2 * ang[:, 1] + p2) * np.sin(ang[:, 0])
    st, ct = np.sin(ang[:, 0]), np.cos(ang[:, 0])
    pts = np.stack([radius * st * np.cos(ang[:, 1]), radius * ct, radius * st * np.sin(ang[:, 1])], axis=1)
    pts *= np.asarray(stretch, dtype=np.float64)

    # top up to the exact count: centroid splits (+2 faces, +1 vertex each), one edge split (+1) if odd
    missing = n_faces - faces.shape[0]
    extra_pts, extra_faces = [], []
    faces = faces.copy()
    nv = pts.shape[0]
    pick = rs.permutation(faces.shape[0])[: (missing + 1) // 2]
    for t, fidx in enumerate(pick):
        a, b, c = faces[fidx]
        if missing - 2 * t >= 2:
            extra_pts.append((pts[a] + pts[b] + pts[c]) / 3.0)
            m = nv + len(extra_pts) - 1
            faces[fidx] = (a, b, m)
            extra_faces += [(b, c, m), (c, a, m)]
        else:
            extra_pts.append((pts[b] + pts[c]) / 2.0)
            m = nv + len(extra_pts) - 1
            faces[fidx] = (a, b, m)
            extra_faces += [(a, m, c)]
    if extra_pts:
        pts = np.concatenate([pts, np.asarray(extra_pts)])
        faces = np.concatenate([faces, np.asarray(extra_faces, dtype=np.int64)])
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
