"""ctypes bindings of oracle/liborc.so (the CPU restatement of the reference; the CHECKER).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liborc.so")

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("emission", C.c_float * 3), ("ambient", C.c_float * 3),
                ("diffuse", C.c_float * 3), ("specular", C.c_float * 3), ("shininess", C.c_float),
                ("brdf_id", C.c_int32)]


class MeshDesc(C.Structure):
    _fields_ = [("vertices", fp), ("normals", fp), ("indices", ip), ("n_vertices", C.c_uint32),
                ("n_faces", C.c_uint32), ("material", Material), ("compute_kdtree", C.c_int32)]


class MaterialExt(C.Structure):
    _fields_ = [("refraction_index", C.c_float), ("checker_color1", C.c_float * 3), ("checker_color2", C.c_float * 3),
                ("checker_cell", C.c_float * 3)]


class ObjectDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("mesh", MeshDesc), ("ext", MaterialExt), ("center", C.c_float * 3),
                ("radius", C.c_float), ("corners", C.c_float * 12), ("n_corners", C.c_uint32)]


class BuildOpts(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("max_children", C.c_int32), ("use_kdtree", C.c_int32), ("device", C.c_int32)]


class CameraDesc(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("rotation", C.c_float * 9), ("viewport_w", C.c_float),
                ("viewport_h", C.c_float), ("near_plane", C.c_float), ("width", C.c_int32), ("height", C.c_int32)]


class Settings(C.Structure):
    _fields_ = [("epsilon", C.c_float), ("near_plane", C.c_float), ("far_plane", C.c_float),
                ("max_recursion_depth", C.c_int32), ("multi_samples", C.c_int32), ("brdf_samples", C.c_int32),
                ("lighting_model", C.c_int32), ("max_mirror_bounces", C.c_int32), ("rng_seed", C.c_uint64),
                ("light_origin", C.c_float * 3), ("light_emission", C.c_float * 3), ("light_ambient", C.c_float * 3),
                ("tile_size", C.c_int32), ("tile_rank", C.c_int32), ("tile_world", C.c_int32), ("collect_stats", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("secondary_rays", C.c_uint64),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("hits", C.c_uint64)]


_lib = None


def build():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_scene_create.argtypes = [C.POINTER(MeshDesc), C.c_uint32, C.POINTER(BuildOpts), C.POINTER(C.c_void_p)]
        L.orc_scene_create_objects.argtypes = [C.POINTER(ObjectDesc), C.c_uint32, C.POINTER(BuildOpts), C.POINTER(C.c_void_p)]
        L.orc_sphere_intersection.argtypes = [fp, fp, fp, C.c_float, C.c_float, C.c_int, fp]
        L.orc_plane_intersection.argtypes = [fp, fp, fp, C.c_uint32, C.c_float, fp]
        L.orc_refraction_direction.argtypes = [fp, fp, C.c_float, fp]
        L.orc_refraction_direction.restype = None
        L.orc_checker_color.argtypes = [fp, fp, C.POINTER(MaterialExt), fp]
        L.orc_checker_color.restype = None
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_destroy.restype = None
        L.orc_scene_set_brdf.argtypes = [C.c_void_p, fp, C.c_int32, fp, C.c_int32]
        L.orc_scene_object_info.argtypes = [C.c_void_p, C.c_uint32, fp, fp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64), ip]
        L.orc_scene_object_normals.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_scene_object_normals.restype = fp
        L.orc_scene_epsilon.argtypes = [C.c_void_p]
        L.orc_scene_epsilon.restype = C.c_float
        L.orc_camera_setup.argtypes = [C.c_void_p, fp, C.c_float, C.c_float, C.c_int32, C.c_int32, C.POINTER(CameraDesc)]
        L.orc_camera_setup.restype = None
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(Settings), C.c_int32, C.c_int32, fp, ip, ip, fp,
                                 C.POINTER(Stats)]
        L.orc_trace_rays.argtypes = [C.c_void_p, C.POINTER(Settings), C.c_uint64, fp, fp, ip, ip, fp, fp, fp]
        L.orc_shade_rays.argtypes = [C.c_void_p, C.POINTER(Settings), C.c_uint64, fp, fp, ip, fp, fp, fp, fp]
        L.orc_aabb_intersection.argtypes = [fp, fp, fp, C.c_float, C.c_float]
        L.orc_intersects_triangle.argtypes = [fp, fp, fp, fp, C.c_float, fp]
        L.orc_reflection_direction.argtypes = [fp, fp, fp]
        L.orc_reflection_direction.restype = None
        L.orc_normalize.argtypes = [fp]
        L.orc_normalize.restype = None
        L.orc_rgb_to_hsv.argtypes = [fp, fp]
        L.orc_rgb_to_hsv.restype = None
        L.orc_hsv_to_rgb.argtypes = [fp, fp]
        L.orc_hsv_to_rgb.restype = None
        L.orc_remap_angles.argtypes = [fp, fp, fp]
        L.orc_remap_angles.restype = None
        L.orc_remap_vectors.argtypes = [fp, fp, fp, fp]
        L.orc_remap_vectors.restype = None
        L.orc_brdf_luminocity.argtypes = [C.c_void_p, C.c_int32, C.c_int32, fp, fp, fp, fp]
        L.orc_brdf_luminocity.restype = C.c_float
        L.orc_phong.argtypes = [fp, fp, fp, C.c_float, fp, fp, fp, fp, fp, fp, C.c_int32, fp]
        L.orc_phong.restype = None
        L.orc_quat_rotation.argtypes = [fp, fp]
        L.orc_quat_rotation.restype = None
        L.orc_rng_init.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_rng_init.restype = C.c_uint64
        L.orc_rng_next_float.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_rng_next_float.restype = C.c_float
        L.orc_compute_normals.argtypes = [fp, C.c_uint32, ip, C.c_uint32, fp]
        L.orc_compute_normals.restype = None
        L.orc_image_argb8.argtypes = [fp, C.c_int32, C.c_int32, C.c_int, C.POINTER(C.c_uint32)]
        L.orc_image_argb8.restype = None
        L.orc_ply_load.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(fp), C.POINTER(fp),
                                   C.POINTER(ip), ip]
        L.orc_ply_free.argtypes = [C.c_void_p]
        L.orc_ply_free.restype = None
        _lib = L
    return _lib


def f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def P(a):
    return a.ctypes.data_as(fp)


def PI(a):
    return a.ctypes.data_as(ip)


def vec(*v):
    return (C.c_float * len(v))(*v)


def material(kind=0, diffuse=(1, 1, 1), shininess=20.0, brdf_id=0, specular=None, ambient=None, emission=(0, 0, 0)):
    m = Material()
    m.kind = kind
    m.diffuse[:] = diffuse
    m.specular[:] = specular if specular is not None else diffuse
    m.ambient[:] = ambient if ambient is not None else diffuse
    m.emission[:] = emission
    m.shininess = shininess
    m.brdf_id = brdf_id
    return m


def material_ext(refraction_index=0.0, color1=(0, 0, 0), color2=(0, 0, 0), cell=(1, 1, 1)):
    e = MaterialExt()
    e.refraction_index = refraction_index
    e.checker_color1[:] = color1
    e.checker_color2[:] = color2
    e.checker_cell[:] = cell
    return e


def sphere(center, radius, mat, ext=None):
    """A Sphere(radius, origin, colorInfo) entry for OracleScene."""
    return ("sphere", tuple(center), float(radius), mat, ext)


def plane(corners, mat, ext=None):
    """A Plane(corners, colorInfo) entry for OracleScene."""
    return ("plane", [tuple(c) for c in corners], None, mat, ext)


class OracleScene:
    """objects: list of (vertices, indices, normals-or-None, Material) meshes and sphere(...) / plane(...) entries,
    in Scene.objects order."""

    def __init__(self, meshes, max_depth=20, max_children=5, use_kdtree=True):
        L = lib()
        self._keep = []
        descs = (ObjectDesc * max(len(meshes), 1))()
        for k, entry in enumerate(meshes):
            if isinstance(entry[0], str):
                kind, geo, radius, m, ext = entry
                descs[k].kind = 1 if kind == "sphere" else 2
                if kind == "sphere":
                    descs[k].center[:] = geo
                    descs[k].radius = radius
                else:
                    flat = [float(x) for c in geo for x in c]
                    descs[k].corners[: len(flat)] = flat
                    descs[k].n_corners = len(geo)
                descs[k].mesh.material = m
                if ext is not None:
                    descs[k].ext = ext
                continue
            v, i, n, m = entry[:4]
            v = f32(v).reshape(-1, 3)
            i = np.ascontiguousarray(np.asarray(i, dtype=np.int32)).reshape(-1, 3)
            n = None if n is None else f32(n).reshape(-1, 3)
            self._keep += [v, i, n]
            descs[k].kind = 0
            descs[k].mesh.vertices = P(v)
            descs[k].mesh.normals = P(n) if n is not None else None
            descs[k].mesh.indices = PI(i)
            descs[k].mesh.n_vertices = v.shape[0]
            descs[k].mesh.n_faces = i.shape[0]
            descs[k].mesh.material = m
            if len(entry) > 4 and entry[4] is not None:
                descs[k].ext = entry[4]
            if len(entry) > 5 and entry[5] is not None:   # this mesh's own computeKDTree (TriangleMesh.java:89)
                descs[k].mesh.compute_kdtree = 1 if entry[5] else -1
        self.n_vertices = [d.mesh.n_vertices for d in descs[: len(meshes)]]
        opts = BuildOpts(max_depth, max_children, 1 if use_kdtree else 0, -1)
        self.h = C.c_void_p()
        rc = L.orc_scene_create_objects(descs, len(meshes), C.byref(opts), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("orc_scene_create_objects failed: %d" % rc)
        self.n_objs = len(meshes)

    def set_brdf(self, angles, weights):
        a, w = f32(angles), f32(weights)
        assert lib().orc_scene_set_brdf(self.h, P(a), a.shape[0], P(w), w.shape[0]) == 0

    def epsilon(self):
        return float(lib().orc_scene_epsilon(self.h))

    def object_info(self, k):
        mm = np.zeros(6, dtype=np.float32)
        sd = C.c_float()
        nn, nl, ns = C.c_uint64(), C.c_uint64(), C.c_uint64()
        md = C.c_int32()
        assert lib().orc_scene_object_info(self.h, k, P(mm), C.byref(sd), C.byref(nn), C.byref(nl), C.byref(ns), C.byref(md)) == 0
        return {"minmax": mm, "shadow_distance": np.float32(sd.value), "n_nodes": nn.value, "n_leaves": nl.value,
                "n_leaf_slots": ns.value, "max_depth": md.value}

    def normals(self, k):
        p = lib().orc_scene_object_normals(self.h, k)
        return np.ctypeslib.as_array(p, shape=(self.n_vertices[k] * 3,)).copy().reshape(-1, 3)

    def camera(self, orientation, fov, near, W, H):
        cam = CameraDesc()
        lib().orc_camera_setup(self.h, vec(*orientation), fov, near, W, H, C.byref(cam))
        return cam

    def render(self, cam, st, n_threads=8, pixel_stride=1, want_stats=False):
        W, H = cam.width, cam.height
        rgb = np.zeros((H, W, 3), dtype=np.float32)
        hid = np.full((H, W), -2, dtype=np.int32)
        hobj = np.full((H, W), -2, dtype=np.int32)
        hw = np.zeros((H, W), dtype=np.float32)
        stats = Stats()
        rc = lib().orc_render(self.h, C.byref(cam), C.byref(st), n_threads, pixel_stride, P(rgb), PI(hid), PI(hobj), P(hw),
                              C.byref(stats))
        assert rc == 0
        out = {"rgb": rgb, "hit_id": hid, "hit_obj": hobj, "hit_w": hw}
        if want_stats:
            out["stats"] = {k: getattr(stats, k) for k, _ in Stats._fields_}
        return out

    def trace(self, st, origins, dirs):
        o, d = f32(origins).reshape(-1, 3), f32(dirs).reshape(-1, 3)
        n = o.shape[0]
        hid = np.empty(n, dtype=np.int32)
        hobj = np.empty(n, dtype=np.int32)
        hw = np.empty(n, dtype=np.float32)
        hp = np.empty((n, 3), dtype=np.float32)
        hn = np.empty((n, 3), dtype=np.float32)
        assert lib().orc_trace_rays(self.h, C.byref(st), n, P(o), P(d), PI(hid), PI(hobj), P(hw), P(hp), P(hn)) == 0
        return {"face": hid, "object": hobj, "w": hw, "point": hp, "normal": hn}

    def shade(self, st, origins, dirs):
        """Closest hit and lightingModel.getPixelColor(info, 0) of each ray (orc_shade_rays)."""
        o, d = f32(origins).reshape(-1, 3), f32(dirs).reshape(-1, 3)
        n = o.shape[0]
        hobj = np.empty(n, dtype=np.int32)
        hw = np.empty(n, dtype=np.float32)
        hp, hn, rgb = (np.empty((n, 3), dtype=np.float32) for _ in range(3))
        assert lib().orc_shade_rays(self.h, C.byref(st), n, P(o), P(d), PI(hobj), P(hw), P(hp), P(hn), P(rgb)) == 0
        return {"object": hobj, "w": hw, "point": hp, "normal": hn, "rgb": rgb}

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def settings(epsilon=1e-15, near=0.01, far=3000.0, ms=1, brdf_samples=1, lighting=1, seed=0,
             light_origin=(0, 100, 100), light_emission=(0.3, 0.3, 0.3), light_ambient=(0.5, 0.5, 0.9), bounces=64, max_recursion_depth=5):
    st = Settings()
    st.epsilon = epsilon
    st.near_plane = near
    st.far_plane = far
    st.max_recursion_depth = max_recursion_depth
    st.multi_samples = ms
    st.brdf_samples = brdf_samples
    st.lighting_model = lighting
    st.max_mirror_bounces = bounces
    st.rng_seed = seed
    st.light_origin[:] = light_origin
    st.light_emission[:] = light_emission
    st.light_ambient[:] = light_ambient
    st.tile_size = 0
    st.tile_rank = 0
    st.tile_world = 1
    st.collect_stats = 0
    return st


def ply_load(path):
    L = lib()
    nv, nf, hn = C.c_uint32(), C.c_uint32(), C.c_int32()
    pv, pn, pi = fp(), fp(), ip()
    rc = L.orc_ply_load(path.encode(), C.byref(nv), C.byref(nf), C.byref(pv), C.byref(pn), C.byref(pi), C.byref(hn))
    if rc != 0:
        raise RuntimeError("orc_ply_load failed")
    v = np.ctypeslib.as_array(pv, shape=(max(nv.value, 1) * 3,))[: nv.value * 3].copy().reshape(-1, 3)
    n = np.ctypeslib.as_array(pn, shape=(max(nv.value, 1) * 3,))[: nv.value * 3].copy().reshape(-1, 3)
    i = np.ctypeslib.as_array(pi, shape=(max(nf.value, 1) * 3,))[: nf.value * 3].copy().reshape(-1, 3)
    L.orc_ply_free(pv)
    L.orc_ply_free(pn)
    L.orc_ply_free(pi)
    return v, i, (n if hn.value else None)
