"""Throughput with D frames in flight (D scene handles, D streams), profiling helper: does the next frame's bulk hide a frame's tails?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ray_tracer_amd as rt

W, H, spp = 1920, 1080, 4
cfg = rt.meshgen.STANFORD["dragon"]
v, i = rt.meshgen.stanford_standin("dragon")
light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))
mesh = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
for D in (1, 2, 3):
    scenes = [rt.Scene([mesh], light, cfg["orientation"], rt.PhongLightingModel(light, [mesh]), cfg["fov"]) for _ in range(D)]
    cams = [rt.Camera(s, spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024) for s in scenes]
    streams = [torch.cuda.Stream() for _ in range(D)]
    bufs = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(D)]
    frames = [rt._abi.Frame(b.data_ptr(), None, None, None, None) for b in bufs]
    def run(n):
        for k in range(n):
            d = k % D
            cams[d].render_device(frames[d], stream=streams[d].cuda_stream)
    run(2 * D)
    torch.cuda.synchronize()
    N = 60
    t0 = time.perf_counter()
    run(N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print("frames in flight %d: %.3f ms per frame, %.1f Mrays/s" % (D, dt * 1e3, 10292566 / dt / 1e6), flush=True)
    same = all(torch.equal(bufs[0], b) for b in bufs[1:])
    print("  frames identical:", same)
    for s in scenes:
        s.close()
