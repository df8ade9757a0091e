"""Host-side mirror of the reference's renderer interface, over the C ABI of libmi355rt.so.

The reference is Java and no JDK exists in this pipeline, so this mirror (same class names,
argument order and meaning as the Java classes it stands in for) is what the parity tests and the
benchmark drive.  It holds no arithmetic of the hot path: scene build, camera constants, tracing
and shading all happen behind include/mi355rt.h.

    reference (src/stephen/ranger/ar/...)              here
    RTStatics (settings)              RTStatics.java:18-37      RTStatics
    lighting/Light                    Light.java:7-16           Light
    lighting/LightingModel            LightingModel.java        LightingModel
    lighting/PhongLightingModel       PhongLightingModel.java   PhongLightingModel
    materials/ColorInformation        ColorInformation.java     ColorInformation
    materials/BRDFMaterial            BRDFMaterial.java         BRDFMaterial
    materials/ReflectionMaterial      ReflectionMaterial.java   ReflectionMaterial
    materials/RefractionMaterial      RefractionMaterial.java   RefractionMaterial
    materials/CheckerboardMaterial    CheckerboardMaterial.java CheckerboardMaterial
    sceneObjects/TriangleMesh         TriangleMesh.java         TriangleMesh
    sceneObjects/Sphere               Sphere.java               Sphere
    sceneObjects/Plane                Plane.java                Plane
    Scene                             Scene.java                Scene
    Camera                            Camera.java               Camera
"""
import ctypes as C

import time

import numpy as np

from . import _abi


class RTStatics:
    """RTStatics.java:18-25 -- mutable global settings, as in the reference."""
    EPSILON = 1e-15
    NEAR_PLANE = 0.01
    FAR_PLANE = 3000.0
    MAX_RECURSION_DEPTH = 5
    MAX_CHILDREN = 5
    MAX_DEPTH = 20
    # not in the reference: cap on mirror-to-mirror recursion, where KDTree.getColor has no cut-off
    MAX_MIRROR_BOUNCES = 64


def _f32(a, shape=None):
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        arr = arr.reshape(shape)
    return arr


def _fp(arr):
    return arr.ctypes.data_as(_abi.c_float_p)


def _ip(arr):
    return arr.ctypes.data_as(_abi.c_int32_p)


class Light:
    """Light(origin, emission, ambient) -- colours as float RGB triples (Color.getColorComponents)."""

    def __init__(self, origin=(100.0, 100.0, 100.0), emission=(1.0, 1.0, 1.0), ambient=(1.0, 1.0, 1.0)):
        self.origin = tuple(float(v) for v in origin)
        self.emission = tuple(float(v) for v in emission)
        self.ambient = tuple(float(v) for v in ambient)


class LightingModel:
    """LightingModel: pixel colour = material colour."""
    kind = _abi.LIGHTING_PLAIN

    def __init__(self):
        self.camera = None

    def setCamera(self, camera):
        self.camera = camera


class PhongLightingModel(LightingModel):
    """PhongLightingModel(light, objects): one point light, one shadow ray."""
    kind = _abi.LIGHTING_PHONG

    def __init__(self, light, objects=None):
        super().__init__()
        self.light = light
        self.objects = objects


class ColorInformation:
    """ColorInformation(diffuse[, shininess]) or (emission, ambient, diffuse, specular, shininess)."""
    kind = _abi.MAT_COLOR
    brdf_id = 0

    def __init__(self, *args):
        black = (0.0, 0.0, 0.0)
        if len(args) == 0:
            args = (black,)
        if len(args) == 1:
            args = (args[0], 20.0)
        if len(args) == 2:
            diffuse, shininess = args
            args = (black, diffuse, diffuse, diffuse, shininess)
        if len(args) != 5:
            raise TypeError("ColorInformation takes 0, 1, 2 or 5 arguments")
        self.emission, self.ambient, self.diffuse, self.specular = (tuple(float(v) for v in c) for c in args[:4])
        self.shininess = float(args[4])

    def _abi_material(self):
        m = _abi.Material()
        m.kind = self.kind
        m.emission[:] = self.emission
        m.ambient[:] = self.ambient
        m.diffuse[:] = self.diffuse
        m.specular[:] = self.specular
        m.shininess = self.shininess
        m.brdf_id = int(self.brdf_id)
        return m

    def _abi_material_ext(self):
        return _abi.MaterialExt()


class BRDFMaterial(ColorInformation):
    """BRDFMaterial(materialID, color).  The parsed CUReT tables are handed over once through
    BRDFMaterial.setTables(angles[n,4], weights[m,n]) (the Java statics brdfDirections / brdfWeights)."""
    kind = _abi.MAT_BRDF
    tables = None

    def __init__(self, materialID, color):
        super().__init__(color)
        self.brdf_id = int(materialID)

    @classmethod
    def setTables(cls, angles, weights):
        angles = _f32(angles)
        weights = _f32(weights)
        if angles.ndim != 2 or angles.shape[1] != 4 or weights.ndim != 2 or weights.shape[1] != angles.shape[0]:
            raise ValueError("need angles[n,4] and weights[m,n]")
        cls.tables = (angles, weights)

    @staticmethod
    def parseBRDFFile(path):
        """The token rule of BRDFMaterial.parseBRDFFile (BRDFMaterial.java:109-116): split each line on
        single spaces, drop token 0, keep tokens that contain a digit."""
        rows = []
        with open(path, "r") as f:
            for line in f.read().splitlines():
                toks = line.split(" ")
                while toks and toks[-1] == "":
                    toks.pop()
                rows.append([np.float32(t) for t in toks[1:] if any(ch.isdigit() for ch in t)])
        return rows


class ReflectionMaterial(ColorInformation):
    """ReflectionMaterial(diffuse): ColorInformation(diffuse, 100) whose colour is a mirror bounce."""
    kind = _abi.MAT_REFLECTION

    def __init__(self, diffuse):
        super().__init__(diffuse, 100.0)


class RefractionMaterial(ColorInformation):
    """RefractionMaterial(diffuse, refractionIndex): ColorInformation(diffuse, 100) whose colour is what the
    refracted ray sees plus an absorption term (RefractionMaterial.java:29-59)."""
    kind = _abi.MAT_REFRACTION
    INDEX_OF_AIR = 1.00029
    INDEX_OF_GLASS = 1.52
    INDEX_OF_ICE = 1.31
    INDEX_OF_WATER = 1.33
    INDEX_OF_QUARTZ = 1.46
    INDEX_OF_SAPPHIRE = 1.77
    INDEX_OF_DIAMOND = 2.417

    def __init__(self, diffuse, refractionIndex):
        super().__init__(diffuse, 100.0)
        self.refractionIndex = float(refractionIndex)

    def _abi_material_ext(self):
        e = _abi.MaterialExt()
        e.refraction_index = self.refractionIndex
        return e


class CheckerboardMaterial(ColorInformation):
    """CheckerboardMaterial(c1, c2, cellWidth, cellHeight, cellDepth): ColorInformation() (all black, shininess
    20) whose colour alternates between c1 and c2 over the object's bounds (CheckerboardMaterial.java:21-49)."""
    kind = _abi.MAT_CHECKERBOARD

    def __init__(self, c1, c2, cellWidth, cellHeight, cellDepth):
        super().__init__()
        self.color1 = tuple(float(v) for v in c1)
        self.color2 = tuple(float(v) for v in c2)
        self.cell = (float(cellWidth), float(cellHeight), float(cellDepth))

    def _abi_material_ext(self):
        e = _abi.MaterialExt()
        e.checker_color1[:] = self.color1
        e.checker_color2[:] = self.color2
        e.checker_cell[:] = self.cell
        return e


class Sphere:
    """Sphere(radius, origin, colorInfo) (Sphere.java:15-22); intersected analytically behind its BoundingSphere."""

    def __init__(self, radius=1.0, origin=(0.0, 0.0, 0.0), colorInfo=None):
        self.radius = float(radius)
        self.origin = tuple(float(v) for v in origin)
        self.colorInfo = colorInfo if colorInfo is not None else ColorInformation()
        self.computeKDTree = True

    def getBoundingVolume(self):
        return self

    def _fill(self, desc, keep):
        desc.kind = _abi.OBJ_SPHERE
        desc.center[:] = self.origin
        desc.radius = self.radius


class Plane:
    """Plane(corners, colorInfo) (Plane.java:28-76): three or four corners; the hit region is the corners'
    axis-aligned box on the plane, as Plane.pointInPolygon has it."""

    def __init__(self, corners=None, colorInfo=None):
        if corners is None:   # Plane() (Plane.java:78-80)
            corners = [(-1, 0, -1), (-1, 0, 1), (1, 0, -1), (1, 0, 1)]
            colorInfo = CheckerboardMaterial((1.0, 1.0, 0.0), (0.0, 0.0, 1.0), 1.0, 1.0, 1.0)
        c = _f32(corners).reshape(-1, 3)
        if c.shape[0] < 3:
            raise ValueError("The array of corner values must contain three or more vertices")   # Plane.java:33-35
        self.corners = c[:4]
        self.colorInfo = colorInfo if colorInfo is not None else ColorInformation()
        self.computeKDTree = True

    def getBoundingVolume(self):
        return self

    def _fill(self, desc, keep):
        desc.kind = _abi.OBJ_PLANE
        flat = self.corners.reshape(-1)
        desc.corners[: flat.size] = [float(v) for v in flat]
        desc.n_corners = self.corners.shape[0]


class TriangleMesh:
    """TriangleMesh(modelLocation, colorInfo, computeKDTree).

    modelLocation is a PLY path (parsed by the library's reader with the reference loader's rules) or
    a (vertices[n,3], indices[f,3]) / (vertices, indices, normals) tuple of arrays."""

    def __init__(self, modelLocation, colorInfo, computeKDTree=True):
        self.colorInfo = colorInfo
        self.computeKDTree = bool(computeKDTree)
        if isinstance(modelLocation, (str, bytes)):
            self.vertices, self.indices, self.normals = load_ply(modelLocation)
        else:
            verts, idx = modelLocation[0], modelLocation[1]
            nrm = modelLocation[2] if len(modelLocation) > 2 else None
            self.vertices = _f32(verts).reshape(-1, 3)
            self.indices = np.ascontiguousarray(np.asarray(idx, dtype=np.int32)).reshape(-1, 3)
            self.normals = None if nrm is None else _f32(nrm).reshape(-1, 3)

    def getBoundingVolume(self):
        return self

    def _fill(self, desc, keep):
        v, i, n = self.vertices, self.indices, self.normals
        keep += [v, i, n]
        desc.kind = _abi.OBJ_MESH
        desc.mesh.vertices = _fp(v)
        desc.mesh.normals = _fp(n) if n is not None else None
        desc.mesh.indices = _ip(i)
        desc.mesh.n_vertices = v.shape[0]
        desc.mesh.n_faces = i.shape[0]


def debug_switch(name=None, value=None):
    """mi355rt_debug_switch: set one of the library's tuning switches (it reads the MI355RT_* environment once, at first
    use); value None = the built-in default, name None = every switch back to what the environment said."""
    lib = _abi.load()
    _abi.check(lib.mi355rt_debug_switch(name.encode() if name is not None else None, str(value).encode() if value is not None else None))


def load_ply(path):
    """mi355rt_ply_load: returns (vertices, indices, normals-or-None)."""
    lib = _abi.load()
    nv, nf, hn = C.c_uint32(), C.c_uint32(), C.c_int32()
    pv, pn, pi = _abi.c_float_p(), _abi.c_float_p(), _abi.c_int32_p()
    if isinstance(path, str):
        path = path.encode()
    t0 = time.perf_counter()
    _abi.check(lib.mi355rt_ply_load(path, C.byref(nv), C.byref(nf), C.byref(pv), C.byref(pn), C.byref(pi), C.byref(hn)))
    load_ply.last_parse_s = time.perf_counter() - t0   # the library's share (the copies into numpy arrays below are this wrapper's)
    try:
        v = np.ctypeslib.as_array(pv, shape=(max(nv.value, 1) * 3,))[: nv.value * 3].copy().reshape(-1, 3)
        n = np.ctypeslib.as_array(pn, shape=(max(nv.value, 1) * 3,))[: nv.value * 3].copy().reshape(-1, 3) if hn.value else None
        i = np.ctypeslib.as_array(pi, shape=(max(nf.value, 1) * 3,))[: nf.value * 3].copy().reshape(-1, 3)
    finally:
        lib.mi355rt_free(pv)
        lib.mi355rt_free(pn)
        lib.mi355rt_free(pi)
    return v, i, n


def compute_normals(vertices, indices):
    """mi355rt_compute_normals (TriangleMesh.computeNormals)."""
    lib = _abi.load()
    v = _f32(vertices).reshape(-1, 3)
    i = np.ascontiguousarray(np.asarray(indices, dtype=np.int32)).reshape(-1, 3)
    out = np.empty_like(v)
    _abi.check(lib.mi355rt_compute_normals(_fp(v), v.shape[0], _ip(i), i.shape[0], _fp(out)))
    return out


class Scene:
    """Scene(objects, light, cameraOrientation, lightingModel, fov).

    Like the reference's constructor it sets the global RTStatics.EPSILON from the scene's span
    (Scene.java:35-36); unlike it, it also hands the meshes to the library, which builds the
    reference's hierarchy and uploads it to the GPU (`device`: HIP ordinal, -1 = current)."""

    def __init__(self, objects, light, cameraOrientation, lightingModel, fov, device=-1):
        self.objects = [o.getBoundingVolume() for o in objects]
        self.light = light
        self.cameraOrientation = tuple(float(v) for v in cameraOrientation)
        self.lightingModel = lightingModel
        self.fov = float(fov)
        self._lib = _abi.load()
        self._handle = C.c_void_p()

        descs = (_abi.ObjectDesc * max(len(self.objects), 1))()
        keep = []
        for k, obj in enumerate(self.objects):   # TriangleMesh, Sphere or Plane, in Scene.objects order
            obj._fill(descs[k], keep)
            descs[k].mesh.material = obj.colorInfo._abi_material()
            descs[k].ext = obj.colorInfo._abi_material_ext()
            descs[k].mesh.compute_kdtree = 1 if obj.computeKDTree else -1   # per mesh, as TriangleMesh.java:89 passes it on
        opts = _abi.BuildOpts(RTStatics.MAX_DEPTH, RTStatics.MAX_CHILDREN, 1, int(device))
        _abi.check(self._lib.mi355rt_scene_create_objects(descs, len(self.objects), C.byref(opts), C.byref(self._handle)))
        if any(isinstance(m.colorInfo, BRDFMaterial) for m in self.objects):
            if BRDFMaterial.tables is None:
                raise ValueError("BRDFMaterial.setTables(angles, weights) must be called before building a BRDF scene")
            ang, wts = BRDFMaterial.tables
            _abi.check(self._lib.mi355rt_brdf_tables(self._handle, _fp(ang), ang.shape[0], _fp(wts), wts.shape[0]))
        eps = C.c_float()
        _abi.check(self._lib.mi355rt_scene_epsilon(self._handle, C.byref(eps)))
        RTStatics.EPSILON = float(eps.value)

    def setDevices(self, devices):
        """mi355rt_scene_set_devices: every later frame of this scene is spread over these HIP devices (devices[0] = the
        scene's own, where the frame is gathered); the reference's createImage() fan-out (Camera.java:129-182)."""
        arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        _abi.check(self._lib.mi355rt_scene_set_devices(self._handle, arr, len(devices)))

    def multi_info(self):
        """mi355rt_scene_multi_info: ranks, transport of the exchange step, devices, each rank's kernel time of the last frame."""
        mi = _abi.MultiInfo()
        _abi.check(self._lib.mi355rt_scene_multi_info(self._handle, C.byref(mi)))
        n = mi.n_ranks
        return {"n_ranks": n, "transport": _abi.TRANSPORT_NAMES[mi.transport], "devices": list(mi.devices[:n]),
                "render_ms": [float(x) for x in mi.render_ms[:n]]}

    def view(self):
        """mi355rt_scene_view: a second frame context on this scene (shared arrays in HBM, own scratch), so that frames
        rendered through different views on different streams are in flight together.  Cameras are made on a view
        exactly as on a scene."""
        v = object.__new__(Scene)
        v.objects, v.light, v.cameraOrientation, v.lightingModel, v.fov = self.objects, self.light, self.cameraOrientation, self.lightingModel, self.fov
        v._lib = self._lib
        v._handle = C.c_void_p()
        _abi.check(self._lib.mi355rt_scene_view(self._handle, C.byref(v._handle)))
        return v

    def object_info(self, k):
        mm = (C.c_float * 6)()
        sd = C.c_float()
        nn, nl, ns = C.c_uint64(), C.c_uint64(), C.c_uint64()
        md = C.c_int32()
        _abi.check(self._lib.mi355rt_scene_object_info(self._handle, k, mm, C.byref(sd), C.byref(nn), C.byref(nl),
                                                       C.byref(ns), C.byref(md)))
        return {"minmax": np.array(mm[:], dtype=np.float32), "shadow_distance": np.float32(sd.value),
                "n_nodes": nn.value, "n_leaves": nl.value, "n_leaf_slots": ns.value, "max_depth": md.value}

    def layout(self):
        """mi355rt_debug_scene_layout: sizes of the device arrays."""
        out = (C.c_uint64 * 8)()
        _abi.check(self._lib.mi355rt_debug_scene_layout(self._handle, out))
        return dict(zip(("node_records", "leaf_descriptors", "group_faces", "leaf_slots", "groups", "shared_groups", "big_leaves", "device_bytes"),
                        [int(x) for x in out]))

    def digest(self):
        """mi355rt_debug_scene_digest: (nodes, leaf slots, normals) digests and whether any mesh was built on the GPU."""
        out = (C.c_uint64 * 4)()
        _abi.check(self._lib.mi355rt_debug_scene_digest(self._handle, out))
        return (out[0], out[1], out[2]), bool(out[3])

    def build_seconds(self):
        """mi355rt_scene_build_seconds: where scene creation spent its time."""
        out = (C.c_double * 7)()
        _abi.check(self._lib.mi355rt_scene_build_seconds(self._handle, out))
        return dict(zip(("host_validate_bounds", "gpu_upload_normals", "gpu_hierarchy", "gpu_leaf_groups", "gpu_records", "host_records_groups",
                         "copy_to_scene"), [round(float(x), 4) for x in out]))

    def device_bytes(self):
        b = C.c_uint64()
        _abi.check(self._lib.mi355rt_scene_device_bytes(self._handle, C.byref(b)))
        return b.value

    def close(self):
        if getattr(self, "_handle", None) and self._handle.value:
            self._lib.mi355rt_scene_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Camera:
    """Camera(scene, multiSamples, brdfSamples, nearPlane, screenWidth, screenHeight).

    createImage() is the drop-in seam (Camera.java:119): here it blocks until the frame is done and
    fills `pixels` (float32 [W][H][3], as Camera.pixels[x][y]), `hit_id`, `hit_obj` and `hit_w` ([W][H])."""

    def __init__(self, scene, multiSamples, brdfSamples, nearPlane, screenWidth, screenHeight, rng_seed=0):
        self.scene = scene
        self.objects = scene.objects
        self.lightingModel = scene.lightingModel
        self.light = scene.light
        self.multiSamples = int(multiSamples)
        self.brdfSamples = int(brdfSamples)
        self.orientation = scene.cameraOrientation
        self.nearPlaneDistance = float(nearPlane)
        self.screenWidth = int(screenWidth)
        self.screenHeight = int(screenHeight)
        self.rng_seed = int(rng_seed)
        self._pinned = None
        self._pinned_ptr = None
        self.listeners = []
        self._lib = _abi.load()
        self._cam = _abi.CameraDesc()
        orient = (C.c_float * 3)(*self.orientation)
        _abi.check(self._lib.mi355rt_camera_setup(scene._handle, orient, scene.fov, self.nearPlaneDistance,
                                                  self.screenWidth, self.screenHeight, C.byref(self._cam)))
        self.origin = np.array(self._cam.origin[:], dtype=np.float32)
        self.rotation = np.array(self._cam.rotation[:], dtype=np.float32).reshape(3, 3)
        self.viewportWidth = float(self._cam.viewport_w)
        self.viewportHeight = float(self._cam.viewport_h)
        self.pixels = None
        self.hit_id = None
        self.hit_obj = None
        self.hit_w = None
        self.lightingModel.setCamera(self)

    def settings(self, tile_rank=0, tile_world=1, tile_size=0, collect_stats=False, collect_times=False, collect_diag=False, tile_compact=False):
        st = _abi.Settings()
        st.epsilon = RTStatics.EPSILON
        st.near_plane = RTStatics.NEAR_PLANE
        st.far_plane = RTStatics.FAR_PLANE
        st.max_recursion_depth = RTStatics.MAX_RECURSION_DEPTH
        st.multi_samples = self.multiSamples
        st.brdf_samples = self.brdfSamples
        st.lighting_model = self.lightingModel.kind
        st.max_mirror_bounces = RTStatics.MAX_MIRROR_BOUNCES
        st.rng_seed = self.rng_seed
        st.light_origin[:] = self.light.origin
        st.light_emission[:] = self.light.emission
        st.light_ambient[:] = self.light.ambient
        st.tile_size = tile_size
        st.tile_rank = tile_rank
        st.tile_world = tile_world
        st.tile_compact = 1 if tile_compact else 0
        st.collect_stats = (_abi.COLLECT_COUNTS if collect_stats else 0) | (_abi.COLLECT_TIMES if collect_times else 0) | \
            (_abi.COLLECT_DIAG if collect_diag else 0)
        return st

    def addActionListener(self, listener):
        self.listeners.append(listener)

    def pinOutput(self, on=True):
        """Keep ONE page-locked set of output arrays for this camera, as Camera.pixels is one array (Camera.java:38):
        createImage() then overwrites them in place, and the copies back from the device run at PCIe speed."""
        if self._pinned_ptr:
            # arrays handed out earlier are views of this memory: after this call they must not be used
            self._pinned = None
            _abi.check(self._lib.mi355rt_host_free(self._pinned_ptr))
            self._pinned_ptr = None
        if on:
            W, H = self.screenWidth, self.screenHeight
            n = W * H
            ptr = C.c_void_p()
            _abi.check(self._lib.mi355rt_host_alloc(n * 28, C.byref(ptr)))
            self._pinned_ptr = ptr
            base = (C.c_uint8 * (n * 28)).from_address(ptr.value)
            self._pinned = (np.frombuffer(base, dtype=np.float32, count=n * 3).reshape(H, W, 3),
                            np.frombuffer(base, dtype=np.int32, count=n, offset=n * 12).reshape(H, W),
                            np.frombuffer(base, dtype=np.int32, count=n, offset=n * 16).reshape(H, W),
                            np.frombuffer(base, dtype=np.float32, count=n, offset=n * 20).reshape(H, W),
                            np.frombuffer(base, dtype=np.uint32, count=n, offset=n * 24).reshape(H, W))

    def __del__(self):
        try:
            if getattr(self, "_pinned_ptr", None):
                self.pinOutput(False)
        except Exception:
            pass

    def createImage(self, outputs=("pixels", "hit_id", "hit_obj", "hit_w")):
        """Camera.createImage().  `outputs`: which of the frame's arrays come back to the host -- "pixels" (Camera.pixels),
        "image" (Camera.image, 8-bit ARGB written by the frame's own kernels), "hit_id", "hit_obj", "hit_w"; the default is what
        the parity tests compare.  Only what is asked for crosses PCIe (mi355rt_render_host)."""
        W, H = self.screenWidth, self.screenHeight
        if self._pinned:
            rgb, hid, hobj, hw, img = self._pinned
        else:
            rgb = np.empty((H, W, 3), dtype=np.float32) if "pixels" in outputs else None
            hid = np.empty((H, W), dtype=np.int32) if "hit_id" in outputs else None
            hobj = np.empty((H, W), dtype=np.int32) if "hit_obj" in outputs else None
            hw = np.empty((H, W), dtype=np.float32) if "hit_w" in outputs else None
            img = np.empty((H, W), dtype=np.uint32) if "image" in outputs else None
        st = self.settings()
        hf = _abi.HostFrame(_fp(rgb) if "pixels" in outputs else None, img.ctypes.data_as(_abi.c_uint32_p) if "image" in outputs else None,
                            _ip(hid) if "hit_id" in outputs else None, _ip(hobj) if "hit_obj" in outputs else None,
                            _fp(hw) if "hit_w" in outputs else None)
        _abi.check(self._lib.mi355rt_render_host(self.scene._handle, C.byref(self._cam), C.byref(st), C.byref(hf)))
        if "pixels" in outputs:
            self._rgb_rows = rgb
            self.pixels = rgb.transpose(1, 0, 2)   # [x][y][c] like Camera.pixels
        if "image" in outputs:
            self.image = img                       # [H, W] uint32 ARGB, Camera.image
        if "hit_id" in outputs:
            self.hit_id = hid.T
        if "hit_obj" in outputs:
            self.hit_obj = hobj.T                  # index into Scene.objects (IntersectionInformation.intersectionObject), -1 = miss
        if "hit_w" in outputs:
            self.hit_w = hw.T
        for listener in self.listeners:   # ActionEvent id 1, "finished" (Camera.java:170-172)
            listener(self, 1, "finished")
        self.listeners = []

    def getImage(self, normalized=False):
        """Camera.image (normalized=False) / Camera.getNormalizedImage() as uint32 ARGB [H, W] from the last frame."""
        H, W = self.screenHeight, self.screenWidth
        out = np.empty((H, W), dtype=np.uint32)
        _abi.check(self._lib.mi355rt_image_argb8(_fp(self._rgb_rows), W, H, 1 if normalized else 0,
                                                 out.ctypes.data_as(_abi.c_uint32_p), -1))
        return out

    def getNormalizedImage(self):
        return self.getImage(normalized=True)

    def render_device(self, frame, stream=None, **kw):
        """mi355rt_render_device: `frame` is an _abi.Frame of device pointers; asynchronous."""
        st = self.settings(**kw)
        _abi.check(self._lib.mi355rt_render_device(self.scene._handle, C.byref(self._cam), C.byref(st), C.byref(frame),
                                                   C.c_void_p(stream) if stream else None))

    def render_device_noalloc(self, frame, stream=None, **kw):
        """mi355rt_render_device_noalloc: the same frame out of the library's static device scratch (no allocation)."""
        st = self.settings(**kw)
        _abi.check(self._lib.mi355rt_render_device_noalloc(self.scene._handle, C.byref(self._cam), C.byref(st), C.byref(frame),
                                                           C.c_void_p(stream) if stream else None))

    def frame_stats(self):
        s = _abi.Stats()
        _abi.check(self._lib.mi355rt_frame_stats(self.scene._handle, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _abi.Stats._fields_}

    def kernel_profile(self, reset=False):
        """mi355rt_kernel_profile: {kernel name: {ms, launches, rays, box_tests, tri_tests}}."""
        kp = _abi.KernelProfile()
        _abi.check(self._lib.mi355rt_kernel_profile(self.scene._handle, C.byref(kp), 1 if reset else 0))
        return {name: {f: getattr(kp, f)[k] for f, _ in _abi.KernelProfile._fields_} for k, name in enumerate(_abi.KERNEL_NAMES)}

    def debug_counters(self):
        out = (C.c_uint64 * 64)()
        _abi.check(self._lib.mi355rt_debug_counters(self.scene._handle, out))
        return list(out)

    def getClosestIntersection(self, origins, directions):
        """Camera.getClosestIntersection(null, origin, direction, null, 0) for arrays of rays."""
        o = _f32(origins).reshape(-1, 3)
        d = _f32(directions).reshape(-1, 3)
        n = o.shape[0]
        hid = np.empty(n, dtype=np.int32)
        hobj = np.empty(n, dtype=np.int32)
        hw = np.empty(n, dtype=np.float32)
        hp = np.empty((n, 3), dtype=np.float32)
        hn = np.empty((n, 3), dtype=np.float32)
        st = self.settings()
        _abi.check(self._lib.mi355rt_trace_rays(self.scene._handle, C.byref(st), n, _fp(o), _fp(d), _ip(hid), _ip(hobj),
                                                _fp(hw), _fp(hp), _fp(hn)))
        return {"face": hid, "object": hobj, "w": hw, "point": hp, "normal": hn}
