"""ray-tracer_amd: MI355X-native per-pixel trace loop of sranger/Ray-Tracer.

The product is libmi355rt.so (HIP kernels + C ABI, include/mi355rt.h, sources in csrc/).  The
Python modules are the host-side mirror of the reference's renderer interface (host.py), the ctypes
view of the ABI (_abi.py), procedural input meshes (meshgen.py) and image-tile sharding over
torch.distributed (tiles.py).  The directory name is not a Python identifier; import it through the
`ray_tracer_amd` shim at the repository root.
"""
from . import _abi, meshgen  # noqa: F401
from .host import (BRDFMaterial, Camera, CheckerboardMaterial, ColorInformation, Light, LightingModel,  # noqa: F401
                   PhongLightingModel, Plane, ReflectionMaterial, RefractionMaterial, RTStatics, Scene, Sphere,
                   TriangleMesh, compute_normals, debug_switch, load_ply)

__all__ = ["BRDFMaterial", "Camera", "CheckerboardMaterial", "ColorInformation", "Light", "LightingModel",
           "PhongLightingModel", "Plane", "ReflectionMaterial", "RefractionMaterial", "RTStatics", "Scene", "Sphere",
           "TriangleMesh", "compute_normals", "debug_switch", "load_ply", "meshgen"]
