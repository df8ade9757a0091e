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
