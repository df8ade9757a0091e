"""Work-list walker diagnostics on the bench workload (profiling helper, not a test; run with MI355RT_WALKER=lds):
entries and chunks per list, rounds, timeline."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ray_tracer_amd as rt

model = sys.argv[1] if len(sys.argv) > 1 else "dragon"
fuse = os.environ.get("MI355RT_FUSE_SHADOW", "")
W, H, spp = 1920, 1080, 4
cfg = rt.meshgen.STANFORD[model]
v, i = rt.meshgen.stanford_standin(model)
light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
mesh = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
scene = rt.Scene([mesh], light, cfg["orientation"], rt.PhongLightingModel(light, [mesh]), cfg["fov"])
cam = rt.Camera(scene, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024)
rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
frame = rt._abi.Frame(rgb.data_ptr(), None, None, None, None)
for _ in range(3):
    cam.render_device(frame, collect_stats=True)
torch.cuda.synchronize()
d = cam.debug_counters()
M = (1 << 64) - 1
print("rays", d[16:19], "box", d[19:22], "tri", d[22:25])
waves = max(d[10], 1)
print("rounds (sum over waves) %d; node: %d entries in %d chunks (fill %.3f); leaf: %d in %d (%.3f); resolve: %d in %d (%.3f); admin: %d in %d (%.3f)" % (
    d[10], d[6], d[7], d[6] / max(64.0 * d[7], 1), d[8], d[9], d[8] / max(64.0 * d[9], 1), d[63], d[62], d[63] / max(64.0 * d[62], 1),
    d[11], d[12], d[11] / max(64.0 * d[12], 1)))
print("chunks per round and wave: node %.2f leaf %.2f resolve %.2f admin %.2f" % (d[7] / waves, d[9] / waves, d[62] / waves, d[12] / waves))
print("longest wave %.3f ms; sums: lifetime %.1f ms, after dry %.1f ms" % (d[13] / 1e5, d[14] / 1e5, d[15] / 1e5))
for kind, kname in ((0, "primary"), (2, "shadow")):
    t = d[25 + 6 * kind: 31 + 6 * kind]
    if not t[3]:
        continue
    t0 = (~t[0]) & M
    print("  %s timeline (ms from first wave start): first wave sees the queue dry %.3f, last %.3f, last wave ends %.3f" % (
        kname, (((~t[1]) & M) - t0) / 1e5, (t[2] - t0) / 1e5, (t[3] - t0) / 1e5))

if any(d[43:50]):
    tot = float(sum(d[43:50]))
    print("  round time by part (share of wave time): barrier wait %.3f, LDS state read %.3f, node/leaf data arrival %.3f, node/leaf compute %.3f, append %.3f, resolve chunks %.3f, admin chunks %.3f" % tuple(x / tot for x in d[43:50]))
    print("  ticks per node/leaf chunk: state read %.0f, data %.0f, compute %.0f; per chunk of any kind: append %.0f; per round: barrier %.0f; per resolve chunk %.0f; per admin chunk %.0f" % (
        d[44] / max(d[7] + d[9], 1), d[45] / max(d[7] + d[9], 1), d[46] / max(d[7] + d[9], 1), d[47] / max(d[7] + d[9] + d[62] + d[12], 1), d[43] / max(d[10], 1), d[48] / max(d[62], 1), d[49] / max(d[12], 1)))
scene.close()
