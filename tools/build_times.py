"""Scene() times on the stand-in models, GPU builder (twice: the first scene of a process pays HIP's start-up) and host
builder, with the phases mi355rt_scene_build_seconds reports (profiling helper, not a test).
usage: build_times.py [model ...] [--no-host]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (HIP runtime load order, see tests/conftest.py)
import ray_tracer_amd as rt

models = [a for a in sys.argv[1:] if not a.startswith("--")] or ["dragon"]
for model in models:
    cfg = rt.meshgen.STANFORD[model]
    v, i = rt.meshgen.stanford_standin(model)
    mesh = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
    light = rt.Light()
    res = {}
    for builder in ("gpu", "gpu", "host"):
        if builder == "host" and ("--no-host" in sys.argv or i.shape[0] > 12_000_000):
            continue
        rt.debug_switch("BUILDER", "host" if builder == "host" else None)   # (the library reads the environment only once)
        t0 = time.time()
        scene = rt.Scene([mesh], light, cfg["orientation"], rt.LightingModel(), cfg["fov"])
        dt = time.time() - t0
        info = scene.object_info(0)
        dig = scene.digest()[0] if i.shape[0] < 12_000_000 else None
        res[builder] = (dt, dig)
        print(model, i.shape[0], "faces", builder, "Scene() %.3f s" % dt, {k: round(x, 4) for k, x in scene.build_seconds().items()},
              {k: info[k] for k in ("n_nodes", "n_leaf_slots", "max_depth")}, flush=True)
        scene.close()
    rt.debug_switch("BUILDER", None)
    if "host" in res and res["gpu"][1] is not None:
        print(model, "digests equal:", res["gpu"][1] == res["host"][1], flush=True)
