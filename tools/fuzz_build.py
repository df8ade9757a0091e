"""Randomised GPU-builder-vs-host-builder comparison (validation helper): random meshes, depth and leaf-size limits;
the scene digest (node records, leaf slots, normals) must be identical.  usage: fuzz_build.py [first_seed] [n_cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import ray_tracer_amd as rt
from test_gpu_build import _scene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = on_gpu = 0
t0 = time.time()
for seed in range(first, first + count):
    rs = np.random.RandomState(9000 + seed)
    scale = float(rs.choice([0.2, 3.0, 400.0]))
    meshes = []
    for k in range(int(rs.randint(1, 3))):
        if rs.rand() < 0.6:
            v, i = rt.meshgen.blob(int(rs.randint(12, 40000)), span=scale * rs.uniform(0.3, 1.0), seed=int(rs.randint(1 << 20)),
                                   stretch=tuple(rs.uniform(0.3, 1.0, 3)))
        else:
            v, i = rt.meshgen.soup(int(rs.randint(10, 30000)), span=scale, tri_size=rs.uniform(0.005, 0.2), seed=int(rs.randint(1 << 20)))
        t = rs.rand()
        if t < 0.15:
            i = np.ascontiguousarray(np.concatenate([i, i[rs.permutation(i.shape[0])]]), dtype=np.int32)
        elif t < 0.3:
            i = i.copy()
            sel = rs.choice(i.shape[0], size=max(1, i.shape[0] // 20), replace=False)
            i[sel, 2] = i[sel, 1]
        elif t < 0.4:      # many faces in one plane (a split that separates nothing on one axis)
            v = v.copy()
            v[:, int(rs.randint(0, 3))] = np.float32(rs.uniform(-1, 1) * scale)
        meshes.append((v, i))
    kw = dict(max_depth=int(rs.choice([1, 3, 8, 20, 20, 20])), max_children=int(rs.choice([1, 2, 5, 5, 5, 16])))
    host = _scene(rt, meshes, "host", **kw)
    gpu = _scene(rt, meshes, "gpu", **kw)
    dh, _ = host.digest()
    dg, g = gpu.digest()
    on_gpu += 1 if g else 0
    keys = ("n_nodes", "n_leaves", "n_leaf_slots", "max_depth")
    ok = dh == dg and all(host.object_info(k)[q] == gpu.object_info(k)[q] for k in range(len(meshes)) for q in keys)
    bad += 0 if ok else 1
    if not ok or seed % 50 == 0:
        print("seed %d scale %g meshes %s %s gpu-built %s: %s" % (seed, scale, [m[1].shape[0] for m in meshes], kw, g, "ok" if ok else "MISMATCH"), flush=True)
    host.close()
    gpu.close()
print("%d cases (%d built on the GPU), %d mismatches, %.0f s" % (count, on_gpu, bad, time.time() - t0))
sys.exit(1 if bad else 0)
