"""A second, independent restatement of the reference's core arithmetic in numpy float32 scalars
(pure-Python loops, small cases only).  It exists to cross-check oracle/rt_oracle.c: two restatements
written separately from the Java text should agree bit for bit.  Citations: /root/reference/src/stephen/ranger/ar/.
"""
import math

import numpy as np

f32 = np.float32


def vdot(a, b):          # Vector3f.dot
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def vcross(a, b):        # Vector3f.cross(a, b)
    return [f32(f32(a[1] * b[2]) - f32(a[2] * b[1])), f32(f32(b[0] * a[2]) - f32(b[2] * a[0])), f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))]


def vsub(a, b):
    return [f32(a[0] - b[0]), f32(a[1] - b[1]), f32(a[2] - b[2])]


def vnormalize(a):       # Vector3f.normalize
    s = f32(f32(f32(a[0] * a[0]) + f32(a[1] * a[1])) + f32(a[2] * a[2]))
    with np.errstate(all="ignore"):
        n = f32(np.float64(1.0) / np.sqrt(np.float64(s)))
        return [f32(a[0] * n), f32(a[1] * n), f32(a[2] * n)]


def aabb_intersection(o, d, mm, near=f32(0.01), far=f32(3000.0)):   # RTStatics.java:102-158
    with np.errstate(all="ignore"):
        div = [f32(1.0) / f32(x) for x in d]
        t = []
        for c in range(3):
            lo, hi = f32(mm[c]), f32(mm[3 + c])
            if div[c] >= 0:
                t.append((f32(f32(lo - o[c]) * div[c]), f32(f32(hi - o[c]) * div[c])))
            else:
                t.append((f32(f32(hi - o[c]) * div[c]), f32(f32(lo - o[c]) * div[c])))
        txmin, txmax = t[0]
        tymin, tymax = t[1]
        if txmin > tymax or tymin > txmax:
            return False
        if tymin > txmin:
            txmin = tymin
        if tymax < txmax:
            txmax = tymax
        tzmin, tzmax = t[2]
        if txmin > tzmax or tzmin > txmax:
            return False
        if tzmin > txmin:
            txmin = tzmin
        if tzmax < txmax:
            txmax = tzmax
        return bool(txmin < far and txmax > near)


def intersects_triangle(o, d, p, n, eps):   # Triangle.java:85-133 ; p, n = 3x3
    with np.errstate(all="ignore"):
        p0, p1, p2 = [[f32(x) for x in row] for row in p]
        e1, e2 = vsub(p1, p0), vsub(p2, p0)
        pv = vcross(d, e2)
        divisor = vdot(pv, e1)
        if divisor < eps and divisor > -eps:
            return None
        T = vsub(o, p0)
        q = vcross(T, e1)
        u = f32(vdot(pv, T) / divisor)
        if u < 0 or u > 1:
            return None
        v = f32(vdot(q, d) / divisor)
        if v < 0 or f32(v + u) > 1:
            return None
        w = f32(f32(f32(1.0) - u) - v)
        out = []
        for arr in (p, n):
            for c in range(3):
                out.append(f32(f32(f32(w * f32(arr[0][c])) + f32(u * f32(arr[1][c]))) + f32(v * f32(arr[2][c]))))
        dx, dy, dz = f32(o[0] - out[0]), f32(o[1] - out[1]), f32(o[2] - out[2])
        d2 = f32(f32(f32(dx * dx) + f32(dy * dy)) + f32(dz * dz))
        out.append(f32(np.sqrt(np.float64(d2))))
        return out


def reflection_direction(n, dr):   # RTStatics.java:67-76
    N = [f32(x * f32(2.0)) for x in n]
    s = vdot(dr, n)
    N = [f32(x * s) for x in N]
    return vnormalize(vsub(dr, N))


def quat_axis_angle(axis, angle_deg):   # RTStatics.java:356-361 + Quat4f.set(AxisAngle4f)
    angle = f32(np.float64(f32(angle_deg)) / 180.0 * math.pi)
    amag = f32(np.sqrt(np.float64(f32(f32(f32(axis[0] * axis[0]) + f32(axis[1] * axis[1])) + f32(axis[2] * axis[2])))))
    if np.float64(amag) < 0.000001:
        return [f32(0)] * 4
    amag = f32(f32(1.0) / amag)
    mag = f32(math.sin(np.float64(angle) / 2.0))
    w = f32(math.cos(np.float64(angle) / 2.0))
    return [f32(f32(axis[0] * amag) * mag), f32(f32(axis[1] * amag) * mag), f32(f32(axis[2] * amag) * mag), w]


def quat_mul(a, b):   # Quat4f.mul ; (x, y, z, w)
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    w = f32(f32(f32(f32(aw * bw) - f32(ax * bx)) - f32(ay * by)) - f32(az * bz))
    x = f32(f32(f32(f32(aw * bx) + f32(bw * ax)) + f32(ay * bz)) - f32(az * by))
    y = f32(f32(f32(f32(aw * by) + f32(bw * ay)) - f32(ax * bz)) + f32(az * bx))
    z = f32(f32(f32(f32(aw * bz) + f32(bw * az)) + f32(ax * by)) - f32(ay * bx))
    return [x, y, z, w]


def rotation_rows(orientation):   # RTStatics.java:345-354 + Matrix4f.set(Quat4f)
    one, two = f32(1.0), f32(2.0)
    q = quat_mul(quat_mul(quat_axis_angle([f32(0), f32(1), f32(0)], -f32(orientation[0])),
                          quat_axis_angle([f32(1), f32(0), f32(0)], f32(orientation[1]))),
                 quat_axis_angle([f32(0), f32(0), f32(1)], -f32(orientation[2])))
    x, y, z, w = q
    m00 = f32(f32(one - f32(f32(two * y) * y)) - f32(f32(two * z) * z))
    m10 = f32(two * f32(f32(x * y) + f32(w * z)))
    m20 = f32(two * f32(f32(x * z) - f32(w * y)))
    m01 = f32(two * f32(f32(x * y) - f32(w * z)))
    m11 = f32(f32(one - f32(f32(two * x) * x)) - f32(f32(two * z) * z))
    m21 = f32(two * f32(f32(y * z) + f32(w * x)))
    m02 = f32(two * f32(f32(x * z) + f32(w * y)))
    m12 = f32(two * f32(f32(y * z) - f32(w * x)))
    m22 = f32(f32(one - f32(f32(two * x) * x)) - f32(f32(two * y) * y))
    return [m00, m01, m02, m10, m11, m12, m20, m21, m22]


def camera(minmax, orientation, fov, near, W, H):   # Camera.java:50-81
    m = rotation_rows(orientation)
    mn, mx = [f32(x) for x in minmax[:3]], [f32(x) for x in minmax[3:]]
    width, height, depth = f32(mx[0] - mn[0]), f32(mx[1] - mn[1]), f32(mx[2] - mn[2])
    cx, cy, cz = f32(f32(width / f32(2)) + mn[0]), f32(f32(height / f32(2)) + mn[1]), f32(f32(depth / f32(2)) + mn[2])
    distance = f32(f32(width / f32(2)) / f32(math.tan(np.float64(f32(fov)) / 180.0 * math.pi)))
    o = [cx, cy, f32(cz + distance)]
    origin = [f32(f32(f32(m[3 * r] * o[0]) + f32(m[3 * r + 1] * o[1])) + f32(m[3 * r + 2] * o[2])) for r in range(3)]
    vw = f32((f32(W) / f32(H) if W >= H else f32(1.0)) * f32(near))
    vh = f32((f32(1.0) if W >= H else f32(H) / f32(W)) * f32(near))
    return origin, m, vw, vh


# ---------------------------------------------------------------- analytic objects

def least_positive(i, j):   # RTStatics.java:82-100
    if i < 0 and j < 0:
        return f32(-1)
    if i < 0 and j > 0:
        return j
    if i > 0 and j < 0:
        return i
    return i if i < j else j


def _sphere_w(o, d, c, radius, first_threshold):   # Sphere.java:38-99 / BoundingSphere.java:34-95
    f64 = np.float64
    with np.errstate(all="ignore"):
        a = f32(f64(d[0]) ** 2 + f64(d[1]) ** 2 + f64(d[2]) ** 2)
        e = [f32(o[k] - c[k]) for k in range(3)]
        b = f32(f32(2) * f32(f32(f32(d[0] * e[0]) + f32(d[1] * e[1])) + f32(d[2] * e[2])))
        cc = f32(f64(e[0]) ** 2 + f64(e[1]) ** 2 + f64(e[2]) ** 2 - f64(radius) ** 2)
        b24c = f32(f32(f64(b) ** 2) - f32(f32(4) * cc))
        wplus = f32(f32(f64(-b) + np.sqrt(f64(b24c))) / f32(f32(2) * a))
        wminus = f32(f32(f64(-b) - np.sqrt(f64(b24c))) / f32(f32(2) * a))
        if b24c < 0:
            return f32(-1)
        w = least_positive(wplus, wminus)
        if w <= 0:
            return f32(-1)

        def dist(t):
            q = [f32(o[k] + f32(d[k] * t)) for k in range(3)]
            x = [f32(o[k] - q[k]) for k in range(3)]
            return f32(np.sqrt(f64(f32(f32(f32(x[0] * x[0]) + f32(x[1] * x[1])) + f32(x[2] * x[2])))))
        if w == wplus:
            nd = dist(wplus)
            if f64(nd) < first_threshold and wminus > 0:
                w = wminus
            elif f64(nd) < first_threshold and wminus < 0:
                return f32(-1)
        elif w == wminus:
            md = dist(wminus)
            if f64(md) < 0.01 and wplus > 0:
                w = wplus
            elif f64(md) < 0.01 and wplus < 0:
                return f32(-1)
        return w if w > 0 else f32(-1)


def bounding_sphere_intersects(o, d, c, radius, eps):   # BoundingSphere.java:29-102
    return bool(_sphere_w(o, d, c, radius, np.float64(f32(eps))) > 0)


def sphere_intersection(o, d, c, radius, eps):   # Sphere.java:29-112 -> (point, normal, w) or None
    s = f32(f32(eps) * f32(2))
    origin = [f32(o[k] + f32(d[k] * s)) for k in range(3)]
    w = _sphere_w(origin, d, c, radius, 0.01)
    if not w > 0:
        return None
    p = [f32(f32(d[k] * w) + origin[k]) for k in range(3)]
    return p, vnormalize(vsub(p, c)), w


def plane_setup(corners):   # Plane.java:28-76 -> (A, B, C, D, normal, minmax6)
    c = [[f32(x) for x in corners[i if i < len(corners) else 2]] for i in range(4)]   # :60-63

    def m(a, b):
        return f32(a * b)

    def s3(a, b, cc):
        return f32(f32(a + b) + cc)
    A = s3(m(c[0][1], f32(c[1][2] - c[2][2])), m(c[1][1], f32(c[2][2] - c[0][2])), m(c[2][1], f32(c[0][2] - c[1][2])))
    B = s3(m(c[0][2], f32(c[1][0] - c[2][0])), m(c[1][2], f32(c[2][0] - c[0][0])), m(c[2][2], f32(c[0][0] - c[1][0])))
    C = s3(m(c[0][0], f32(c[1][1] - c[2][1])), m(c[1][0], f32(c[2][1] - c[0][1])), m(c[2][0], f32(c[0][1] - c[1][1])))
    D = f32(-s3(m(c[0][0], f32(m(c[1][1], c[2][2]) - m(c[2][1], c[1][2]))), m(c[1][0], f32(m(c[2][1], c[0][2]) - m(c[0][1], c[2][2]))),
                m(c[2][0], f32(m(c[0][1], c[1][2]) - m(c[1][1], c[0][2])))))
    mm = [min(c[i][k] for i in range(4)) for k in range(3)] + [max(c[i][k] for i in range(4)) for k in range(3)]
    return A, B, C, D, vnormalize([A, B, C]), mm


def plane_intersection(o, d, corners, eps):   # Plane.java:87-154 -> (point, normal, w) or None
    A, B, C, D, pn, mm = plane_setup(corners)
    eps = f32(eps)
    with np.errstate(all="ignore"):
        s = f32(eps * f32(2))
        origin = [f32(o[k] + f32(d[k] * s)) for k in range(3)]
        vD = vdot(pn, d)
        if vD > eps:
            pn = [f32(-x) for x in pn]
            vD = vdot(pn, d)
        if not (vD <= -eps or vD >= eps):
            return None
        num = f32(f32(f32(f32(A * origin[0]) + f32(B * origin[1])) + f32(C * origin[2])) + D)
        den = f32(f32(f32(A * d[0]) + f32(B * d[1])) + f32(C * d[2]))
        t = f32(f32(-num) / den)
        if not t > -eps:
            return None
        rD = list(d)
        if vD > -eps:
            rD = [f32(-x) for x in rD]
        pI = [f32(origin[k] + f32(rD[k] * t)) for k in range(3)]
        for k in range(3):
            if not (pI[k] >= f32(mm[k] - eps) and pI[k] <= f32(mm[3 + k] + eps)):
                return None
        tv = vsub(pI, origin)
        w = f32(np.sqrt(np.float64(f32(f32(f32(tv[0] * tv[0]) + f32(tv[1] * tv[1])) + f32(tv[2] * tv[2])))))
        return pI, pn, w


def refraction_direction(n, d, index):   # RefractionMaterial.java:61-95
    f64 = np.float64
    aether = f32(1.00029)
    sn = list(n)
    cos_i = f64(vdot(sn, d))
    if cos_i <= 0:
        nn = f64(f32(f32(index) / aether))
        cos_i = -cos_i
    else:
        nn = f64(f32(aether / f32(index)))
        sn = [f32(-x) for x in sn]
    root = 1.0 - nn * nn * (1.0 - cos_i * cos_i)
    if root < 0:
        return reflection_direction(sn, d)
    k1 = f32(nn)
    k2 = f32(nn * cos_i - math.sqrt(root))
    out = [f32(f32(d[k] * k1) + f32(sn[k] * k2)) for k in range(3)]
    return vnormalize(out)


def checker_color(mm, p, c1, c2, cell):   # CheckerboardMaterial.java:21-49
    span = [f32(f32(mm[3 + k]) - f32(mm[k])) for k in range(3)]
    dist = [f32(f32(p[k]) - f32(mm[k])) for k in range(3)]
    if any(dist[k] < 0 or dist[k] > span[k] for k in range(3)):
        return [f32(0)] * 3
    x, y, z = (int(math.floor(float(f32(dist[k] / f32(cell[k]))))) for k in range(3))
    one = (x % 2 == 0 and y % 2 == 0 and z % 2 == 0) or (x % 2 == 1 and y % 2 == 0 and z % 2 == 1) or \
        (x % 2 == 0 and y % 2 == 1 and z % 2 == 1)
    return [f32(v) for v in (c1 if one else c2)]


# ---- the hierarchy itself, as the reference's recursion (small meshes only: pure Python) -------------------------
def kd_build(vertices, faces, minmax, axis, depth, max_depth, max_children, use_kdtree):   # KDNode ctor, KDNode.java:29-90
    """faces: list of (face id, (i0, i1, i2)); minmax: [lo3, hi3] as f32 lists.  Returns a dict node."""
    node = {"minmax": [f32(x) for x in minmax[0]] + [f32(x) for x in minmax[1]], "faces": faces, "left": None, "right": None}
    if not (use_kdtree and depth < max_depth and len(faces) > max_children):
        return node
    lo, hi = f32(minmax[0][axis]), f32(minmax[1][axis])
    median = f32(f32(f32(hi - lo) / f32(2.0)) + lo)
    left, right = [], []
    for fid, tri in faces:
        vals = [f32(vertices[i][axis]) for i in tri]
        fmin, fmax = min(vals), max(vals)
        if fmax <= median:
            left.append((fid, tri))
        elif fmin >= median:
            right.append((fid, tri))
        else:
            left.append((fid, tri))
            right.append((fid, tri))

    def bounds(fs):   # RTStatics.getMinMax(vertices, faces, out), RTStatics.java:244-264
        pts = [vertices[i] for _, tri in fs for i in tri]
        return [[min(f32(p[c]) for p in pts) for c in range(3)], [max(f32(p[c]) for p in pts) for c in range(3)]]

    nxt = (axis + 1) % 3
    if left:
        node["left"] = kd_build(vertices, left, bounds(left), nxt, depth + 1, max_depth, max_children, use_kdtree)
    if right:
        node["right"] = kd_build(vertices, right, bounds(right), nxt, depth + 1, max_depth, max_children, use_kdtree)
    return node


def kd_intersect(node, o, d, vertices, normals, eps, shadow_distance):   # KDNode.getChildIntersection, KDNode.java:93-123
    """Returns (face id, [point3, normal3, w]) or None."""
    if node["left"] is not None or node["right"] is not None:
        tl = tr = None
        if node["left"] is not None and aabb_intersection(o, d, node["left"]["minmax"]):
            tl = kd_intersect(node["left"], o, d, vertices, normals, eps, shadow_distance)
        if node["right"] is not None and aabb_intersection(o, d, node["right"]["minmax"]):
            tr = kd_intersect(node["right"], o, d, vertices, normals, eps, shadow_distance)
        if tl is None and tr is None:
            return None
        if tl is None:
            return tr
        if tr is None:
            return tl
        return tl if tl[1][6] < tr[1][6] else tr
    closest = None
    for fid, tri in node["faces"]:
        if aabb_intersection(o, d, node["minmax"]):
            t = intersects_triangle(o, d, [vertices[i] for i in tri], [normals[i] for i in tri], eps)
            if t is not None and t[6] > shadow_distance and (closest is None or t[6] < closest[1][6]):
                closest = (fid, t)
    return closest


def mesh_closest(vertices, normals, indices, o, d, eps, max_depth=20, max_children=5, use_kdtree=True, tree=None):
    """KDTree ctor + KDTree.getChildIntersection + Camera.getClosestIntersection's acceptance for ONE mesh
    (KDTree.java:24-65, Camera.java:203-211, Ray.java:9-13).  Returns (tree, hit or None)."""
    if tree is None:
        lo = [min(f32(v[c]) for v in vertices) for c in range(3)]     # over ALL vertices (KDTree.java:35-44)
        hi = [max(f32(v[c]) for v in vertices) for c in range(3)]
        span = max(f32(hi[0] - lo[0]), max(f32(hi[1] - lo[1]), f32(hi[2] - lo[2])))
        tree = {"root": kd_build(vertices, [(k, tuple(int(x) for x in f)) for k, f in enumerate(indices)], [lo, hi], 0, 0,
                                 max_depth, max_children, use_kdtree),
                "shadow_distance": f32(span / f32(2000.0))}
    o = [f32(x) for x in o]
    d = vnormalize([f32(x) for x in d])
    root = tree["root"]
    hit = None
    if aabb_intersection(o, d, root["minmax"]):          # Camera.java:205 object.intersects(ray)
        if aabb_intersection(o, d, root["minmax"]):      # KDTree.getChildIntersection tests it again (KDTree.java:60)
            hit = kd_intersect(root, o, d, vertices, normals, eps, tree["shadow_distance"])
    if hit is not None and not (hit[1][6] > eps):         # Camera.java:208
        hit = None
    return tree, hit
