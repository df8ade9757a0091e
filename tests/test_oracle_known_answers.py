"""Hand-derived known answers for the oracle (the reference has no tests or golden vectors: SURVEY.md 4).
Each expected value below was worked out from the Java text, not from running the oracle."""
import ctypes as C

import numpy as np

f32 = np.float32


def _aabb(orc, o, d, mm, near=0.01, far=3000.0):
    return orc.lib().orc_aabb_intersection(orc.vec(*o), orc.vec(*d), orc.vec(*mm), near, far)


def test_aabb_basic_and_planes(orc):
    box = (-1, -1, -1, 1, 1, 1)
    assert _aabb(orc, (0, 0, 5), (0, 0, -1), box) == 1          # enters at t=4, leaves at t=6
    assert _aabb(orc, (0, 0, 5), (0, 0, 1), box) == 0           # box behind: txmax = -4 < NEAR_PLANE
    assert _aabb(orc, (3, 0, 5), (0, 0, -1), box) == 0          # misses in x (x slab = [-inf..] with 0 direction)
    # NEAR_PLANE: a box whose exit distance is <= 0.01 is culled (RTStatics.java:157)
    assert _aabb(orc, (0, 0, 1.005), (0, 0, -1), (-1, -1, 1.0, 1, 1, 1.0)) == 0    # exit at t = 0.005
    assert _aabb(orc, (0, 0, 1.02), (0, 0, -1), (-1, -1, 1.0, 1, 1, 1.0)) == 1     # exit at t = 0.02
    # FAR_PLANE: entry distance must be < 3000
    assert _aabb(orc, (0, 0, 3002), (0, 0, -1), box) == 0       # enters at 3001
    assert _aabb(orc, (0, 0, 3000.5), (0, 0, -1), box) == 1     # enters at 2999.5
    # origin inside the box: txmin < 0 < txmax
    assert _aabb(orc, (0, 0, 0), (0.6, 0.0, 0.8), box) == 1


def test_aabb_signed_zero_direction(orc):
    """div = 1/(-0.0) = -inf takes the `else` branch (RTStatics.java:109-115): for an origin strictly inside
    the slab both products are +-inf with the right signs; on the slab boundary 0 * inf = NaN falls through."""
    box = (-1, -1, -1, 1, 1, 1)
    assert _aabb(orc, (0.5, 0, 5), (0.0, 0, -1), box) == 1
    assert _aabb(orc, (0.5, 0, 5), (-0.0, 0, -1), box) == 1
    assert _aabb(orc, (2.0, 0, 5), (-0.0, 0, -1), box) == 0
    # x = 1 exactly: txmax = (1 - 1) * inf = NaN; no comparison with NaN is true, so NaN survives both
    # "if (tymax < txmax)" and "if (tzmax < txmax)" and the final "txmax > NEAR_PLANE" is false -> miss
    assert _aabb(orc, (1.0, 0, 5), (0.0, 0, -1), box) == 0
    # x = -1 exactly puts the NaN in txmin instead; "tzmin > txmin" is false too, and "txmin < FAR_PLANE" fails
    assert _aabb(orc, (-1.0, 0, 5), (0.0, 0, -1), box) == 0


def _tri(orc, o, d, p, n=None, eps=1e-15):
    n = n if n is not None else [[0, 0, 1]] * 3
    out = (C.c_float * 7)()
    ok = orc.lib().orc_intersects_triangle(orc.vec(*o), orc.vec(*d), orc.vec(*np.ravel(p)), orc.vec(*np.ravel(n)), eps, out)
    return ok, np.array(out[:], dtype=np.float32)


def test_triangle_unit_right_triangle(orc):
    p = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    ok, r = _tri(orc, (0.25, 0.25, 1.0), (0, 0, -1), p)
    # e1=(1,0,0) e2=(0,1,0) p=d x e2=(1,0,0) det=1; T=(.25,.25,1) u=.25; q=T x e1=(0,1,-.25) v=q.d=.25
    assert ok == 1
    np.testing.assert_array_equal(r, np.array([0.25, 0.25, 0.0, 0, 0, 1, 1.0], dtype=np.float32))
    assert _tri(orc, (0.75, 0.75, 1.0), (0, 0, -1), p)[0] == 0      # u + v > 1
    assert _tri(orc, (-0.1, 0.2, 1.0), (0, 0, -1), p)[0] == 0       # u < 0
    assert _tri(orc, (0.25, 0.25, 1.0), (1, 0, 0), p)[0] == 0       # parallel: divisor = 0 inside (-eps, eps)


def test_triangle_has_no_t_positive_test(orc):
    """t = (q.e2)/det is never computed (Triangle.java:124): a triangle BEHIND the origin is a hit, with the
    positive Euclidean distance."""
    p = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    ok, r = _tri(orc, (0.25, 0.25, -2.0), (0, 0, -1), p)
    assert ok == 1 and r[6] == f32(2.0) and r[2] == 0


def test_triangle_two_sided_and_interpolated_normal_not_renormalised(orc):
    p = [[0, 0, 0], [0, 1, 0], [1, 0, 0]]    # flipped winding: det = -1, still a hit
    n = [[0, 0, 1], [0, 0, 3], [2, 0, 0]]
    ok, r = _tri(orc, (0.5, 0.25, 1.0), (0, 0, -1), p, n)
    assert ok == 1
    # u = .25 (towards p1 = (0,1,0)), v = .5, w = .25 : normal = .25*n0 + .25*n1 + .5*n2
    np.testing.assert_allclose(r[3:6], [1.0, 0.0, 1.0], rtol=0, atol=0)
    np.testing.assert_allclose(r[:3], [0.5, 0.25, 0.0], rtol=0, atol=0)


def test_reflection_direction(orc):
    out = (C.c_float * 3)()
    orc.lib().orc_reflection_direction(orc.vec(0, 1, 0), orc.vec(0.6, -0.8, 0.0), out)
    np.testing.assert_allclose(out[:], [0.6, 0.8, 0.0], rtol=2e-7)
    # un-normalised normal: R = L - 2 (L.N) N with |N| = 2 -> (0.6, -0.8 + 2*1.6*2, 0) normalised
    orc.lib().orc_reflection_direction(orc.vec(0, 2, 0), orc.vec(0.6, -0.8, 0.0), out)
    ref = np.array([0.6, 5.6, 0.0]) / np.hypot(0.6, 5.6)
    np.testing.assert_allclose(out[:], ref, rtol=3e-7)


def test_normalize_uses_double_reciprocal_sqrt(orc):
    v = (C.c_float * 3)(3.0, 4.0, 12.0)
    orc.lib().orc_normalize(v)
    s = f32(1.0 / np.sqrt(np.float64(169.0)))
    np.testing.assert_array_equal(np.array(v[:], dtype=np.float32), np.array([3, 4, 12], dtype=np.float32) * s)


def test_hsv_pure_colours_and_grey(orc):
    def to_hsv(rgb):
        out = (C.c_float * 3)()
        orc.lib().orc_rgb_to_hsv(orc.vec(*rgb), out)
        return list(out)

    def to_rgb(hsv):
        out = (C.c_float * 3)()
        orc.lib().orc_hsv_to_rgb(orc.vec(*hsv), out)
        return list(out)
    # red: x = 0, v = 1; R != x, G == x -> f = B - R = -1, i = 5 -> h = 5 - (-1)/1 = 6
    assert to_hsv((1, 0, 0)) == [6.0, 1.0, 1.0]
    # green: R == x -> f = G - B = 1, i = 3 -> h = 2;  blue: R == x -> f = G - B = -1, i = 3 -> h = 4
    assert to_hsv((0, 1, 0)) == [2.0, 1.0, 1.0]
    assert to_hsv((0, 0, 1)) == [4.0, 1.0, 1.0]
    # h = 6 -> case 6: f = 0, i even -> f = 1, m = n = 0 -> (v, n, m)
    assert to_rgb((6.0, 1.0, 1.0)) == [1.0, 0.0, 0.0]
    assert to_hsv((0.5, 0.5, 0.5)) == [-1.0, 0.0, 0.5]
    assert to_rgb((-1.0, 0.0, 0.3)) == [f32(0.3)] * 3
    # round trip of a saturated colour
    h = to_hsv((0.2, 0.8, 0.3))
    np.testing.assert_allclose(to_rgb(h), [0.2, 0.8, 0.3], rtol=1e-6)
    # value replaced by a luminance > 1 is clamped per channel (RTStatics.java:507-509)
    assert to_rgb((h[0], h[1], 6.0))[1] == 1.0


def test_phong_negative_base_gives_positive_highlight(orc):
    """Math.pow(V.R, 20) keeps no sign for an even exponent (PhongLightingModel.java:44); LdotN is not clamped."""
    out = (C.c_float * 3)()
    one = orc.vec(1, 1, 1)
    # light straight above the point, normal +y, eye looking UP along +y (from below): V = -dir = (0,-1,0)
    # L = (0,1,0), R = L - 2(L.N)N = (0,-1,0), V.R = 1 -> pow = 1; unshadowed:
    orc.lib().orc_phong(one, one, one, 20.0, orc.vec(0.3, 0.3, 0.3), orc.vec(0.5, 0.5, 0.9), orc.vec(0, 10, 0), orc.vec(0, 0, 0),
                        orc.vec(0, 1, 0), orc.vec(0, 1, 0), 0, out)
    exp = f32(1.0 * (1.0 * (1 * 1.0 * f32(0.3) + 1.0 * 1 * 1.0 * f32(0.3) + np.float64(f32(0.4) * f32(0.5)))))
    assert out[0] == exp
    # eye on the other side: V.R = -1 -> pow(-1, 20) = +1: same highlight
    out2 = (C.c_float * 3)()
    orc.lib().orc_phong(one, one, one, 20.0, orc.vec(0.3, 0.3, 0.3), orc.vec(0.5, 0.5, 0.9), orc.vec(0, 10, 0), orc.vec(0, 0, 0),
                        orc.vec(0, 1, 0), orc.vec(0, -1, 0), 0, out2)
    assert out2[0] == out[0]
    # shadowed: shade = 0.6f, no specular
    orc.lib().orc_phong(one, one, one, 20.0, orc.vec(0.3, 0.3, 0.3), orc.vec(0.5, 0.5, 0.9), orc.vec(0, 10, 0), orc.vec(0, 0, 0),
                        orc.vec(0, 1, 0), orc.vec(0, 1, 0), 1, out2)
    assert out2[2] == f32(np.float64(f32(0.6)) * (np.float64(f32(0.3)) + np.float64(f32(0.4) * f32(0.9))))


def test_brdf_remap_phi_zero_row(orc):
    """Row with phi_v = 0: normalizePhi adds 2*pi to any phi < pi/2 (PBRTMath.java:99-101)."""
    out = (C.c_float * 3)()
    th_v, th_i, phi_i = 0.5, 0.25, 2.0
    orc.lib().orc_remap_angles(orc.vec(th_v, 0.0), orc.vec(th_i, phi_i), out)
    F_PI = f32(np.pi)
    phii = f32(f32(0.0) + f32(2.0) * F_PI)          # 0 < pi/2 -> + 2pi
    phio = f32(phi_i)                               # 2.0 >= pi/2 -> unchanged
    dphi = f32(phii - phio)                          # 4.28 > pi -> 2pi - dphi
    dphi = f32(f32(2.0) * F_PI - dphi)
    exp = [f32(f32(np.sin(th_v)) * f32(np.sin(th_i))), f32(dphi / F_PI), f32(f32(np.cos(th_v)) * f32(np.cos(th_i)))]
    np.testing.assert_array_equal(np.array(out[:], dtype=np.float32), np.array(exp, dtype=np.float32))


def test_java_util_random_lcg(orc):
    """new java.util.Random(42).nextInt() == -1170105035 (well known); nextFloat() is its top 24 bits / 2^24."""
    state = C.c_uint64((42 ^ 0x5DEECE66D) & ((1 << 48) - 1))
    x = orc.lib().orc_rng_next_float(C.byref(state))
    assert x == f32(((-1170105035) & 0xFFFFFFFF) >> 8) / f32(1 << 24)
    assert 0.7275 < x < 0.7276


def test_epsilon_rule_and_shadow_distance(orc):
    import ray_tracer_amd as rt
    v, i = rt.meshgen.blob(200, span=9.5)
    s = orc.OracleScene([(v, i, None, orc.material())])
    assert f32(s.epsilon()) == f32(1e-15)
    info = s.object_info(0)
    span = (info["minmax"][3:] - info["minmax"][:3]).max()
    assert info["shadow_distance"] == f32(span / f32(2000.0))
    s.close()
    v, i = rt.meshgen.blob(200, span=10.5)
    s = orc.OracleScene([(v, i, None, orc.material())])
    assert f32(s.epsilon()) == f32(1e-3)
    s.close()


def test_tie_rule_last_leaf_first_slot(orc):
    """Two coincident triangles (same vertices, different face ids): equal w.  In ONE leaf the earlier face
    wins (KDNode.java:114, strict <).  Across leaves the LATER leaf in left-to-right order wins (:105)."""
    quad = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    verts = np.concatenate([quad, quad])          # faces 0 and 1 coincide
    idx = np.array([[0, 1, 2], [3, 4, 5]], dtype=np.int32)
    st = orc.settings(epsilon=1e-15)
    s = orc.OracleScene([(verts, idx, None, orc.material())])       # 2 faces <= MAX_CHILDREN: one leaf
    r = s.trace(st, [[0.25, 0.25, 1.0]], [[0, 0, -1]])
    assert r["face"][0] == 0
    s.close()
    # force a split between two coincident copies placed in different leaves: 6 filler faces on the left,
    # the duplicate pair straddling nothing -> both copies fall in the same child; instead use max_children=0
    s = orc.OracleScene([(verts, idx, None, orc.material())], max_children=0, max_depth=1)
    # depth-0 node splits at x = 0.5: both faces straddle -> both in LEFT and RIGHT leaves; tie => right leaf,
    # within it the first face
    r = s.trace(st, [[0.25, 0.25, 1.0]], [[0, 0, -1]])
    assert r["face"][0] == 0
    s.close()


def test_image_packing_and_normalisation(orc):
    """Camera.image: Color(bound(c)).getRGB() = (int)(c*255+0.5) per channel, alpha 255 (Camera.java:103-107);
    Camera.updateNormalizedImage stretches the HSV value to [0,1] over the frame (Camera.java:217-263)."""
    import numpy as np
    rgb = np.array([[[0.0, 0.5, 1.0], [-0.3, 1.7, 0.25]],
                    [[0.2, 0.2, 0.2], [0.6, 0.3, 0.1]]], dtype=np.float32)
    out = np.zeros((2, 2), dtype=np.uint32)
    orc.lib().orc_image_argb8(orc.P(rgb), 2, 2, 0, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert out[0, 0] == 0xFF0080FF          # 0.5*255+0.5 = 128
    assert out[0, 1] == 0xFF00FF40          # clamped; 0.25*255+0.5 = 64.25 -> 64
    assert out[1, 0] == 0xFF333333
    # normalised: values (max channel) are 1.0, 1.7, 0.2, 0.6 -> min 0.2, max 1.7; grey 0.2 -> 0, the 1.7 pixel -> value 1
    orc.lib().orc_image_argb8(orc.P(rgb), 2, 2, 1, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert out[1, 0] == 0xFF000000
    assert (out[0, 1] >> 8) & 0xFF == 0xFF
    g = (np.float32(0.6) - np.float32(0.2)) / (np.float32(1.7) - np.float32(0.2))      # red channel of the last pixel = its value
    assert (out[1, 1] >> 16) & 0xFF == int(np.float64(np.float32(g * np.float32(255.0))) + 0.5)


# ---------------------------------------------------------------- analytic objects, hand-derived

def _sphere(orc, o, d, c, radius, eps=1e-3, bounding=0):
    out = np.zeros(7, dtype=np.float32)
    rc = orc.lib().orc_sphere_intersection(orc.P(orc.f32(o)), orc.P(orc.f32(d)), orc.P(orc.f32(c)), radius, eps, bounding, orc.P(out))
    return rc, out


def test_sphere_unit_at_origin(orc):
    # Sphere.java:30-33 advances the origin by 2*EPSILON: z = 4.998; roots of t^2 - 9.996 t + 23.980004: 3.998, 5.998
    rc, out = _sphere(orc, (0, 0, 5), (0, 0, -1), (0, 0, 0), 1.0)
    assert rc == 1
    assert np.allclose(out[:3], (0, 0, 1), atol=2e-6) and np.allclose(out[3:6], (0, 0, 1), atol=1e-6)
    assert abs(out[6] - 3.998) < 2e-6                      # w is measured from the advanced origin
    assert _sphere(orc, (0, 2, 5), (0, 0, -1), (0, 0, 0), 1.0)[0] == 0     # b^2 - 4c < 0
    assert _sphere(orc, (0, 0, 5), (0, 0, 1), (0, 0, 0), 1.0)[0] == 0      # both roots negative (leastPositive -> -1)
    rc, out = _sphere(orc, (0, 0, 0), (0, 0, -1), (0, 0, 0), 1.0)          # from inside: the positive root
    assert rc == 1 and np.allclose(out[:3], (0, 0, -1), atol=2e-6) and abs(out[6] - 0.998) < 2e-6
    # BoundingSphere.intersects does not advance the origin and only answers yes / no
    assert _sphere(orc, (0, 0, 5), (0, 0, -1), (0, 0, 0), 1.0, bounding=1)[0] == 1
    assert _sphere(orc, (0, 2, 5), (0, 0, -1), (0, 0, 0), 1.0, bounding=1)[0] == 0


def test_sphere_surface_origin_skips_near_root(orc):
    # a ray leaving the surface inwards: the near root (distance < 0.01 from the origin) is replaced by the far one
    # (Sphere.java:82-98): origin (0,0,1) on the unit sphere, direction -z -> exits at z = -1, w ~ 2 - 2*eps
    rc, out = _sphere(orc, (0, 0, 1), (0, 0, -1), (0, 0, 0), 1.0, eps=1e-3)
    assert rc == 1 and np.allclose(out[:3], (0, 0, -1), atol=3e-6) and abs(out[6] - 1.998) < 3e-6
    # leaving outwards: near root within 0.01 and the other root negative -> null (Sphere.java:94-96)
    assert _sphere(orc, (0, 0, 1.001), (0, 0, 1), (0, 0, 0), 1.0, eps=1e-15)[0] == 0


def _plane(orc, o, d, corners, eps=1e-3):
    out = np.zeros(7, dtype=np.float32)
    c = orc.f32(corners).reshape(-1)
    rc = orc.lib().orc_plane_intersection(orc.P(orc.f32(o)), orc.P(orc.f32(d)), orc.P(c), len(corners), eps, orc.P(out))
    return rc, out


def test_plane_default_unit_floor(orc):
    floor = [(-1, 0, -1), (-1, 0, 1), (1, 0, -1), (1, 0, 1)]   # Plane() (Plane.java:78-80): A = C = D = 0, B = 4
    rc, out = _plane(orc, (0, 2, 0), (0, -1, 0), floor)
    assert rc == 1 and np.allclose(out[:3], (0, 0, 0), atol=1e-6) and tuple(out[3:6]) == (0, 1, 0) and abs(out[6] - 1.998) < 1e-6
    rc, out = _plane(orc, (0, -2, 0), (0, 1, 0), floor)          # from below: the normal is flipped towards the ray (:98-101)
    assert rc == 1 and tuple(out[3:6]) == (0, -1, 0) and abs(out[6] - 1.998) < 1e-6
    assert _plane(orc, (3, 2, 0), (0, -1, 0), floor)[0] == 0     # outside the corners' box (pointInPolygon)
    assert _plane(orc, (0, 2, 0), (1, 0, 0), floor)[0] == 0      # parallel
    assert _plane(orc, (0, 2, 0), (0, 1, 0), floor)[0] == 0      # behind the origin: t < -EPSILON
    assert _plane(orc, (0, 2, 0), (0, -1, 0), [(0, 0, 0), (1, 0, 0), (2, 0, 0)])[0] == -1   # colinear (Plane.java:56-58)
    rc, out = _plane(orc, (0.9, 2, 0.9), (0, -1, 0), floor[:3])  # three corners: the box of the triangle, not the triangle
    assert rc == 1


def test_refraction_direction_known(orc):
    out = np.zeros(3, dtype=np.float32)
    # head-on: n = 1.52 / 1.00029, cosI = 1, root = 1 -> dir * n + N * (n - 1) = (0, 0, -1)
    orc.lib().orc_refraction_direction(orc.P(orc.f32((0, 0, 1))), orc.P(orc.f32((0, 0, -1))), 1.52, orc.P(out))
    assert np.allclose(out, (0, 0, -1), atol=1e-6)
    # the "entering" branch uses n = index / air > 1, so grazing rays are totally reflected (RefractionMaterial.java:81-82)
    d = orc.f32((1, 0, -0.5))
    d /= np.linalg.norm(d)
    orc.lib().orc_refraction_direction(orc.P(orc.f32((0, 0, 1))), orc.P(d), 1.52, orc.P(out))
    assert np.allclose(out, (d[0], 0, -d[2]), atol=1e-6)
    # 45 degrees from behind the surface (cosI > 0): n = air / index, the normal is flipped; Snell in double as a cross-check
    d = orc.f32((np.sqrt(0.5), 0, np.sqrt(0.5)))
    orc.lib().orc_refraction_direction(orc.P(orc.f32((0, 0, 1))), orc.P(d), 1.52, orc.P(out))
    n = 1.00029 / 1.52
    sin_t = n * np.sqrt(0.5)
    assert np.allclose(out, (sin_t, 0, np.sqrt(1 - sin_t * sin_t)), atol=1e-6)


def test_checkerboard_cells(orc):
    import ctypes as C
    mm = orc.f32((0, 0, 0, 10, 0, 10))
    ext = orc.material_ext(0.0, (1, 1, 0), (1, 0, 0), (1, 1, 1))
    out = np.zeros(3, dtype=np.float32)

    def col(p):
        orc.lib().orc_checker_color(orc.P(mm), orc.P(orc.f32(p)), C.byref(ext), orc.P(out))
        return tuple(out)
    assert col((0.5, 0, 0.5)) == (1, 1, 0)     # (even, even, even) -> color1
    assert col((1.5, 0, 0.5)) == (1, 0, 0)     # (odd, even, even)  -> color2
    assert col((1.5, 0, 1.5)) == (1, 1, 0)     # (odd, even, odd)   -> color1
    assert col((0.5, 0, 1.5)) == (1, 0, 0)     # (even, even, odd)  -> color2
    assert col((-0.1, 0, 0.5)) == (0, 0, 0)    # outside the object's bounds -> black (CheckerboardMaterial.java:34-36)
    assert col((0.5, 0.1, 0.5)) == (0, 0, 0)


def _analytic_scene(orc, glass_index=1.52):
    floor = orc.plane([(-50, 0, -100), (-50, -40, 25), (50, -40, 25), (50, 0, -100)],
                      orc.material(4, (0, 0, 0), 20.0), orc.material_ext(0, (1, 1, 0), (1, 0, 0), (10, 10, 10)))
    glass = orc.sphere((0, -12, 0), 5, orc.material(3, (0.5, 0.5, 0.5), 100.0), orc.material_ext(glass_index))
    mirror = orc.sphere((5, -15, -10), 3, orc.material(2, (0, 0, 1), 100.0))
    return orc.OracleScene([floor, glass, mirror])


def test_whitted_scene_structure(orc):
    """RayTracer.getWhittedObjects (RayTracer.java:155-167): bounds, EPSILON, and what single rays see."""
    sc = _analytic_scene(orc)
    assert sc.epsilon() == np.float32(1e-3)
    assert tuple(sc.object_info(0)["minmax"]) == (-50, -40, -100, 50, 0, 25)
    assert tuple(sc.object_info(1)["minmax"]) == (-5, -17, -5, 5, -7, 5)       # BoundingSphere.getMinMax
    st = orc.settings(epsilon=1e-3, lighting=0, light_ambient=(0.25, 0.5, 0.75))
    # straight down the z axis at the glass sphere's centre: object 1, w = distance to its near side
    r = sc.trace(st, [(0, -12, 50)], [(0, 0, -1)])
    assert r["object"][0] == 1 and abs(r["w"][0] - (45 - 0.002)) < 1e-4 and np.allclose(r["normal"][0], (0, 0, 1), atol=1e-6)
    # past everything: miss
    assert sc.trace(st, [(0, 30, 50)], [(0, 0, -1)])["object"][0] == -1
    # down onto the sloping floor (y = -0.32 (z + 100) => y(-50) = -16): object 0, normal facing up
    r = sc.trace(st, [(-40, 20, -50)], [(0, -1, 0)])
    assert r["object"][0] == 0 and abs(r["point"][0][1] + 16.0) < 1e-3 and r["normal"][0][1] > 0.9


def test_recursion_cutoff_returns_diffuse(orc):
    """SceneObject.getColor (SceneObject.java:64-67): at depth >= MAX_RECURSION_DEPTH an analytic object answers with
    its diffuse colour.  Two facing mirror planes: with depth 1 the second mirror answers (0.2, 0.4, 0.6)."""
    a = orc.plane([(-10, -10, 0), (10, -10, 0), (10, 10, 0), (-10, 10, 0)], orc.material(2, (0.9, 0.1, 0.1), 100.0))
    b = orc.plane([(-10, -10, 4), (10, -10, 4), (10, 10, 4), (-10, 10, 4)], orc.material(2, (0.2, 0.4, 0.6), 100.0))
    sc = orc.OracleScene([a, b])
    cam = sc.camera((0, 0, 0), 30.0, 0.01, 1, 1)
    cam.origin[:] = (0, 0, 2)          # between the mirrors; the one pixel looks along (-0.5, 0.5, -1) at mirror a
    for depth, expect in ((1, (0.2, 0.4, 0.6)), (2, (0.9, 0.1, 0.1)), (0, (0.9, 0.1, 0.1))):
        st = orc.settings(epsilon=1e-15, lighting=0, max_recursion_depth=depth)
        rgb = sc.render(cam, st, n_threads=1)["rgb"][0, 0]
        assert np.allclose(rgb, expect, atol=1e-7), (depth, rgb)
