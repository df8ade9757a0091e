"""PCIe-inclusive frame rate: mi355rt_render with HOST output buffers on the bench workload (profiling helper)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (one HIP runtime per process: torch's first)
import ray_tracer_amd as rt

cfg = rt.meshgen.STANFORD["dragon"]
v, i = rt.meshgen.stanford_standin("dragon")
light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
mesh = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
scene = rt.Scene([mesh], light, cfg["orientation"], rt.PhongLightingModel(light, [mesh]), cfg["fov"])
cam = rt.Camera(scene, 4, 1, rt.RTStatics.NEAR_PLANE, 1920, 1080, rng_seed=2024)
for pinned in (False, True):
    cam.pinOutput(pinned)
    for _ in range(3):
        cam.createImage()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        cam.createImage()
    dt = (time.perf_counter() - t0) / n
    print("mi355rt_render (host buffers: 24.9 MB rgb + 8.3 MB ids + 8.3 MB w back over PCIe, %s): %.2f ms per frame"
          % ("page-locked, from mi355rt_host_alloc" if pinned else "pageable memory, fresh arrays per frame", dt * 1e3))
cam.pinOutput(False)
