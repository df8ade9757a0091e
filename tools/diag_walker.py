"""Walker diagnostics on the bench workload (profiling helper, not a test): per-phase lane utilisation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ray_tracer_amd as rt

model = sys.argv[1] if len(sys.argv) > 1 else "dragon"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # > 1: rank 0's share of the frame only
W, H, spp = 1920, 1080, 4
cfg = rt.meshgen.STANFORD[model]
v, i = rt.meshgen.stanford_standin(model)
light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
mesh = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
for lm, name in ((rt.LightingModel(), "primary only"), (rt.PhongLightingModel(light, [mesh]), "primary+shadow")):
    scene = rt.Scene([mesh], light, cfg["orientation"], lm, cfg["fov"])
    cam = rt.Camera(scene, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024)
    rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    frame = rt._abi.Frame(rgb.data_ptr(), None, None, None, None)
    for _ in range(3):   # the third frame runs with the longest-first order from the second
        cam.render_device(frame, collect_stats=True, tile_rank=0, tile_world=world, tile_size=16)   # light: same occupancy as the production kernel
    torch.cuda.synchronize()
    light_d = cam.debug_counters()
    cam.render_device(frame, collect_diag=True, tile_rank=0, tile_world=world, tile_size=16)
    torch.cuda.synchronize()
    d = cam.debug_counters()
    names = ["primary", "shadow", "secondary", "box", "tri", "hits", "node_lane", "node_wave", "leaf_lane", "leaf_wave", "outer", "refill"]
    print(name, dict(zip(names, d)))
    print("  steals %d; longest wave %.2f ms, mean wave %.2f ms, mean time after queue dry %.2f ms (waves of LAST walker launch sets; sums over launches)" % (d[12], d[13] / 1e5, d[14] / 1e5 / (6 * 256 * 4), d[15] / 1e5 / (6 * 256 * 4)))
    print("  triangle tests that repeat one of the ray's last 8 distinct faces (owner lanes and helpers, each with its own list): %d of %d" % (d[63], d[4]))
    print("  node-phase lane util %.3f  leaf-phase lane util %.3f  node trips/outer %.1f leaf trips/outer %.1f" % (
        d[6] / max(64.0 * d[7], 1), d[8] / max(64.0 * d[9], 1), d[7] / max(d[10], 1), d[9] / max(d[10], 1)))
    M = (1 << 64) - 1
    d = light_d[:43] + d[43:59] + light_d[59:]   # timeline from the light run
    print('  tail trips per wave {max,sum}: primary', d[59:61], 'shadow', d[61:63])
    for kind, kname in ((0, "primary"), (2, "shadow")):
        t = d[25 + 6 * kind: 31 + 6 * kind]
        if not t[3]:
            continue
        t0 = (~t[0]) & M
        print("  %s walker timeline (ms from first wave start): first wave finds queue dry %.3f, last wave finds it dry %.3f, last wave ends %.3f; "
              "longest ray %d trips; %d of %d waves got a ray" % (kname, (((~t[1]) & M) - t0) / 1e5, (t[2] - t0) / 1e5, (t[3] - t0) / 1e5, t[4], t[5], 6 * 256 * 4))
    print("  per kind rays/box/tri:", d[16:25])
    print("  rays by owner trips (<16,32,64,128,256,512,1024,more): primary", d[43:51], "shadow", d[51:59])
    scene.close()
