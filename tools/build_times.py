"""Host vs GPU hierarchy build times on the stand-in models (profiling helper, not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (HIP runtime load order, see tests/conftest.py)
import ray_tracer_amd as rt

for model in sys.argv[1:] or ["dragon"]:
    cfg = rt.meshgen.STANFORD[model]
    v, i = rt.meshgen.stanford_standin(model)
    mesh = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
    light = rt.Light()
    res = {}
    for builder in ("gpu", "host", "gpu"):
        if builder == "host":
            os.environ["MI355RT_BUILDER"] = "host"
        else:
            os.environ.pop("MI355RT_BUILDER", None)
        t0 = time.time()
        scene = rt.Scene([mesh], light, cfg["orientation"], rt.LightingModel(), cfg["fov"])
        dt = time.time() - t0
        info = scene.object_info(0)
        dig = scene.digest()[0] if i.shape[0] < 12_000_000 else None
        res[builder] = (dt, dig)
        print(model, i.shape[0], "faces", builder, "Scene() %.3f s" % dt, {k: info[k] for k in ("n_nodes", "n_leaf_slots", "max_depth")}, flush=True)
        scene.close()
    if res["gpu"][1] is not None:
        print(model, "digests equal:", res["gpu"][1] == res["host"][1], flush=True)
