import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
for model in ("dragon", "buddha", "thai", "lucy"):
    v, i = rt.meshgen.stanford_standin(model)
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
    mesh = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
    import time; t=time.time()
    sc = rt.Scene([mesh], light, rt.meshgen.STANFORD[model]["orientation"], rt.PhongLightingModel(light, [mesh]), rt.meshgen.STANFORD[model]["fov"])
    print(model, i.shape[0], sc.layout(), "build %.2f s" % (time.time()-t), flush=True)
    sc.close()
