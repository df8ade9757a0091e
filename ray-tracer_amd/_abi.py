"""ctypes view of include/mi355rt.h (the C ABI of libmi355rt.so).

Nothing here computes: it declares the structs and prototypes, loads the library and turns a
non-zero status into an exception carrying mi355rt_last_error().  There is no fallback: if the
library is missing, or has no HIP device to run on, calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355RT_LIB: tuning builds only (tools/variants.sh); the product is the in-tree libmi355rt.so
LIB_PATH = os.environ.get("MI355RT_LIB") or os.path.join(_HERE, "libmi355rt.so")

MAT_COLOR, MAT_BRDF, MAT_REFLECTION, MAT_REFRACTION, MAT_CHECKERBOARD = 0, 1, 2, 3, 4
OBJ_MESH, OBJ_SPHERE, OBJ_PLANE = 0, 1, 2
MAX_REFRACTION_TERMS = 8
LIGHTING_PLAIN, LIGHTING_PHONG = 0, 1
E_ARG, E_DEVICE, E_MEMORY, E_LIMIT, E_IO = -1, -2, -3, -4, -5
NOALLOC_SCRATCH_BYTES = 128 << 20

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint32_p = C.POINTER(C.c_uint32)


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("emission", C.c_float * 3), ("ambient", C.c_float * 3),
                ("diffuse", C.c_float * 3), ("specular", C.c_float * 3), ("shininess", C.c_float),
                ("brdf_id", C.c_int32)]


class MeshDesc(C.Structure):
    _fields_ = [("vertices", c_float_p), ("normals", c_float_p), ("indices", c_int32_p),
                ("n_vertices", C.c_uint32), ("n_faces", C.c_uint32), ("material", Material), ("compute_kdtree", C.c_int32)]


class MaterialExt(C.Structure):
    _fields_ = [("refraction_index", C.c_float), ("checker_color1", C.c_float * 3), ("checker_color2", C.c_float * 3),
                ("checker_cell", C.c_float * 3)]


class ObjectDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("mesh", MeshDesc), ("ext", MaterialExt), ("center", C.c_float * 3),
                ("radius", C.c_float), ("corners", C.c_float * 12), ("n_corners", C.c_uint32)]


class BuildOpts(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("max_children", C.c_int32), ("use_kdtree", C.c_int32),
                ("device", C.c_int32)]


class CameraDesc(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("rotation", C.c_float * 9), ("viewport_w", C.c_float),
                ("viewport_h", C.c_float), ("near_plane", C.c_float), ("width", C.c_int32),
                ("height", C.c_int32)]


class Settings(C.Structure):
    _fields_ = [("epsilon", C.c_float), ("near_plane", C.c_float), ("far_plane", C.c_float),
                ("max_recursion_depth", C.c_int32), ("multi_samples", C.c_int32),
                ("brdf_samples", C.c_int32), ("lighting_model", C.c_int32),
                ("max_mirror_bounces", C.c_int32), ("rng_seed", C.c_uint64),
                ("light_origin", C.c_float * 3), ("light_emission", C.c_float * 3),
                ("light_ambient", C.c_float * 3), ("tile_size", C.c_int32), ("tile_rank", C.c_int32),
                ("tile_world", C.c_int32), ("collect_stats", C.c_int32), ("tile_compact", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("secondary_rays", C.c_uint64),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("hits", C.c_uint64)]


COLLECT_COUNTS, COLLECT_TIMES, COLLECT_DIAG = 1, 2, 4
K_GENERATE, K_WALK_PRIMARY, K_SHADE, K_WALK_SECONDARY, K_WALK_SHADOW, K_PHONG, K_RESOLVE, K_ANALYTIC, N_KERNELS = 0, 1, 2, 3, 4, 5, 6, 7, 8
KERNEL_NAMES = ("generate", "walk_primary", "shade", "walk_secondary", "walk_shadow", "phong", "resolve", "analytic")


class KernelProfile(C.Structure):
    _fields_ = [("ms", C.c_double * N_KERNELS), ("launches", C.c_uint64 * N_KERNELS), ("rays", C.c_uint64 * N_KERNELS),
                ("box_tests", C.c_uint64 * N_KERNELS), ("tri_tests", C.c_uint64 * N_KERNELS)]


class HostFrame(C.Structure):
    _fields_ = [("rgb", c_float_p), ("argb8", c_uint32_p), ("hit_id", c_int32_p), ("hit_obj", c_int32_p), ("hit_w", c_float_p)]


MAX_RANKS = 16
TRANSPORT_NAMES = ("none", "peer", "rccl")


class MultiInfo(C.Structure):
    _fields_ = [("n_ranks", C.c_int32), ("transport", C.c_int32), ("devices", C.c_int32 * MAX_RANKS), ("render_ms", C.c_float * MAX_RANKS)]


class Frame(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("argb8", C.c_void_p), ("hit_id", C.c_void_p),
                ("hit_obj", C.c_void_p), ("hit_w", C.c_void_p)]


# name -> (restype, argtypes); exactly the entry points include/mi355rt.h declares
PROTOTYPES = {
    "mi355rt_scene_create": (C.c_int, [C.POINTER(MeshDesc), C.c_uint32, C.POINTER(BuildOpts), C.POINTER(C.c_void_p)]),
    "mi355rt_scene_create_objects": (C.c_int, [C.POINTER(ObjectDesc), C.c_uint32, C.POINTER(BuildOpts), C.POINTER(C.c_void_p)]),
    "mi355rt_scene_destroy": (None, [C.c_void_p]),
    "mi355rt_scene_view": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mi355rt_scene_set_devices": (C.c_int, [C.c_void_p, c_int32_p, C.c_int32]),
    "mi355rt_scene_multi_info": (C.c_int, [C.c_void_p, C.POINTER(MultiInfo)]),
    "mi355rt_debug_switch": (C.c_int, [C.c_char_p, C.c_char_p]),
    "mi355rt_render_host": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(Settings), C.POINTER(HostFrame)]),
    "mi355rt_debug_scene_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mi355rt_brdf_tables": (C.c_int, [C.c_void_p, c_float_p, C.c_int32, c_float_p, C.c_int32]),
    "mi355rt_scene_epsilon": (C.c_int, [C.c_void_p, c_float_p]),
    "mi355rt_scene_object_info": (C.c_int, [C.c_void_p, C.c_uint32, c_float_p, c_float_p, C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), c_int32_p]),
    "mi355rt_scene_build_seconds": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "mi355rt_scene_device_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mi355rt_camera_setup": (C.c_int, [C.c_void_p, c_float_p, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                       C.POINTER(CameraDesc)]),
    "mi355rt_render": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(Settings), c_float_p, c_int32_p, c_int32_p, c_float_p]),
    "mi355rt_host_alloc": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p)]),
    "mi355rt_host_free": (C.c_int, [C.c_void_p]),
    "mi355rt_render_device": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(Settings), C.POINTER(Frame), C.c_void_p]),
    "mi355rt_render_device_noalloc": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(Settings), C.POINTER(Frame), C.c_void_p]),
    "mi355rt_tiles_unpack_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Frame), C.c_void_p]),
    "mi355rt_frame_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "mi355rt_kernel_profile": (C.c_int, [C.c_void_p, C.POINTER(KernelProfile), C.c_int]),
    "mi355rt_image_argb8": (C.c_int, [c_float_p, C.c_int32, C.c_int32, C.c_int, c_uint32_p, C.c_int32]),
    "mi355rt_image_argb8_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355rt_trace_rays": (C.c_int, [C.c_void_p, C.POINTER(Settings), C.c_uint64, c_float_p, c_float_p, c_int32_p,
                                     c_int32_p, c_float_p, c_float_p, c_float_p]),
    "mi355rt_debug_arith": (C.c_int, [C.c_uint64, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]),
    "mi355rt_debug_scene_digest": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mi355rt_debug_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mi355rt_ply_load": (C.c_int, [C.c_char_p, c_uint32_p, c_uint32_p, C.POINTER(c_float_p), C.POINTER(c_float_p),
                                   C.POINTER(c_int32_p), c_int32_p]),
    "mi355rt_compute_normals": (C.c_int, [c_float_p, C.c_uint32, c_int32_p, C.c_uint32, c_float_p]),
    "mi355rt_free": (None, [C.c_void_p]),
    "mi355rt_last_error": (C.c_char_p, []),
    "mi355rt_abi_version": (C.c_int, []),
    "mi355rt_device_count": (C.c_int, []),
}


class Mi355rtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mi355rt error %d: %s" % (code, message))
        self.code = code
        self.message = message


_lib = None


def load():
    """Loads libmi355rt.so (built by ray-tracer_amd/csrc/Makefile); raises if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `make -C ray-tracer_amd/csrc` "
                              "(or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise Mi355rtError(rc, load().mi355rt_last_error().decode("utf-8", "replace"))
