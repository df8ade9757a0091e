"""Per-kernel times of BASELINE configurations other than the bench frame (profiling helper, not a test).
usage: config_times.py [buddha_brdf] [brdf_samples]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ray_tracer_amd as rt
from helpers import curet_tables

brdf_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = rt.meshgen.STANFORD["buddha"]
v, i = rt.meshgen.stanford_standin("buddha")
rt.BRDFMaterial.setTables(*curet_tables())   # the reference's own tables (tests/golden/curet_tables.npz)
light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
mesh = rt.TriangleMesh((v, i), rt.BRDFMaterial(15, (0.0, 1.0, 0.0)), True)
scene = rt.Scene([mesh], light, cfg["orientation"], rt.PhongLightingModel(light, [mesh]), cfg["fov"])
for ms in (1, 4):
    cam = rt.Camera(scene, ms, brdf_samples, rt.RTStatics.NEAR_PLANE, 1920, 1080, rng_seed=5)
    rgb = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
    frame = rt._abi.Frame(rgb.data_ptr(), None, None, None, None)
    for _ in range(3):
        cam.render_device(frame)
    torch.cuda.synchronize()
    cam.kernel_profile(reset=True)
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        cam.render_device(frame, collect_times=True)
    e1.record()
    torch.cuda.synchronize()
    kp = cam.kernel_profile(reset=True)
    print("buddha BRDF 1080p, %d spp, brdf_samples %d: %.2f ms per frame; " % (ms, brdf_samples, e0.elapsed_time(e1) / n)
          + ", ".join("%s %.3f" % (k, x["ms"] / n) for k, x in kp.items() if x["launches"]))
