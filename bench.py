#!/usr/bin/env python3
"""bench.py -- Mrays/s (primary + shadow) on the Dragon-class workload, 1920x1080, 4 spp, Phong.

A "step" is one frame: every primary sample of the 1080p image, one shadow ray per hit sample,
shaded, averaged and written.  The scene (flattened hierarchy) is resident in HBM before the timed
region; the framebuffer stays in HBM.  With N > 1 GPUs (one process per GPU, launched by
torch.distributed.run) the frame's 16x16-pixel tiles are interleaved over the ranks, the scene is
replicated, and the step ends with ONE gather of the tiles to rank 0 over RCCL (strong scaling:
the frame is fixed).  Rays are counted by the kernel, not assumed.

Prints ONE JSON line on rank 0 (see the repository's bench contract) with two extra objects:
  roofline     -- the dominant kernel (the primary-ray walker): its algorithmic bytes per launch / its
                  average launch duration, from HIP events the library records around every kernel on
                  the stream it launches on, over the timed region, vs 8 TB/s
  cpu_baseline -- the oracle (C restatement of the reference's algorithm, reference visit order) timed
                  on this host's cores on a bounded sample of the same frame.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_BOX_TEST = 32          # 24 B box + 8 B link per child box tested
BYTES_PER_TRI_TEST = 48          # 3 x float4 positions
BYTES_PER_HIT = 48               # 3 x float4 normals per final hit
BYTES_PER_PIXEL = 16             # f32 RGB + i32 id written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="dragon", help="stand-in from meshgen.STANFORD")
    ap.add_argument("--faces", type=int, default=0, help="override the face count (0 = the model's own)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="tuning aid (single process): render only rank 0's tiles of an N-rank run; the line is not a bench result")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for the baseline sample")
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="extra 'pipelined' figure (N = 1 only): frames rendered with this many in flight; 1 = skip")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: libmi355rt has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import ray_tracer_amd as rt
    from ray_tracer_amd import tiles

    cfg = rt.meshgen.STANFORD[args.model]
    verts, idx = rt.meshgen.stanford_standin(args.model, args.faces or None)
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))          # RayTracer.java:106
    mesh = rt.TriangleMesh((verts, idx), rt.ColorInformation((0.9, 0.9, 0.9)), True)  # RayTracer.java:122-123
    t0 = time.time()
    scene = rt.Scene([mesh], light, cfg["orientation"], rt.PhongLightingModel(light, [mesh]), cfg["fov"], device=local_rank)
    build_s = time.time() - t0
    W, H = args.width, args.height
    cam = rt.Camera(scene, args.spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024)

    rgb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    hit_id = torch.zeros((H, W), dtype=torch.int32, device=dev)
    hit_w = torch.zeros((H, W), dtype=torch.float32, device=dev)
    frame = rt._abi.Frame(rgb.data_ptr(), None, hit_id.data_ptr(), None, hit_w.data_ptr())
    sharder = tiles.TileSharder(W, H, rank, world, tiles.DEFAULT_TILE, dev)
    stream = torch.cuda.current_stream().cuda_stream   # the kernel is launched on torch's current stream

    tile_world = args.emulate_world if (args.emulate_world > 1 and world == 1) else world

    def step(stats=False, times=False):
        cam.render_device(frame, stream=stream, tile_rank=rank, tile_world=tile_world, tile_size=tiles.DEFAULT_TILE,
                          collect_stats=stats, collect_times=times)

    # counted, not assumed: one instrumented frame
    step(stats=True)
    torch.cuda.synchronize()
    st = cam.frame_stats()
    kcounts = cam.kernel_profile(reset=True)   # per-walker counts of that frame
    counts = torch.tensor([st["primary_rays"], st["shadow_rays"], st["secondary_rays"], st["box_tests"], st["tri_tests"],
                           st["hits"]], dtype=torch.float64, device=dev)
    local_counts = counts.clone()
    if world > 1:
        dist.all_reduce(counts)
    primary, shadow, secondary, boxes, tris, hits = [float(x) for x in counts.tolist()]
    rays_per_frame = primary + shadow + secondary

    for _ in range(args.warmup):
        step(times=True)
        sharder.gather(rgb, hit_w, hit_id)
    torch.cuda.synchronize()
    cam.kernel_profile(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step(times=True)
        ev[k][1].record()
        sharder.gather(rgb, hit_w, hit_id)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    frame_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))   # all kernels of one frame, this rank
    kprof = cam.kernel_profile()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    mrays = rays_per_frame / (elapsed / args.steps) / 1e6

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # roofline of the dominant kernel, this rank's launches over the timed region
    kernels = {name: {"ms_per_launch": round(v["ms"] / v["launches"], 4), "launches_per_frame": v["launches"] / args.steps,
                      "ms_per_frame": round(v["ms"] / args.steps, 4)}
               for name, v in kprof.items() if v["launches"]}
    for name, v in kernels.items():
        if kcounts[name]["rays"]:
            v.update(rays=kcounts[name]["rays"], box_tests=kcounts[name]["box_tests"], tri_tests=kcounts[name]["tri_tests"])
    dom = max(kernels, key=lambda n: kernels[n]["ms_per_frame"])
    lp, ls, lsec, lb, lt, lh = [float(x) for x in local_counts.tolist()]
    own_pixels = float(sharder.my_pix.ge(0).sum().item())
    frame_bytes = BYTES_PER_BOX_TEST * lb + BYTES_PER_TRI_TEST * lt + BYTES_PER_HIT * lh + BYTES_PER_PIXEL * own_pixels
    kc = dict(kcounts[dom])
    # MODE_FUSED: the primary launch walked the shadow rays too (no shadow launch); their tests belong to it
    fused = dom == "walk_primary" and "walk_shadow" not in kernels and kcounts["walk_shadow"]["rays"] > 0
    if fused:
        for k in ("rays", "box_tests", "tri_tests"):
            kc[k] += kcounts["walk_shadow"][k]
        kernels[dom].update(rays=kc["rays"], box_tests=kc["box_tests"], tri_tests=kc["tri_tests"])
    n_launch = max(kernels[dom]["launches_per_frame"], 1.0)
    # SURVEY.md 8(d) per-unit figures x the units one launch of that kernel processes (counted by the kernel itself
    # in the instrumented frame): child boxes tested, triangle slots tested, final hits whose normals are fetched
    dom_hits = lh if dom == "walk_primary" else 0.0
    algo_bytes = (BYTES_PER_BOX_TEST * kc["box_tests"] + BYTES_PER_TRI_TEST * kc["tri_tests"] + BYTES_PER_HIT * dom_hits) / n_launch
    dom_ms = kernels[dom]["ms_per_launch"]
    achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            key = "%s_%dx%d_%dspp_n%d" % (args.model, W, H, args.spp, world)
            traffic = tj.get(key, {}).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    dom_rays = max(kc["rays"] / n_launch, 1.0)
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "walk_kernel<0,MODE_FUSED> (primary rays, each hit's shadow ray walked by the same lane)" if fused else
                          {"walk_primary": "walk_kernel<0,MODE_CLOSEST> (primary rays)",
                           "walk_shadow": "walk_kernel<0,MODE_SHADOW_ANY> (shadow rays)"}.get(dom, dom),
                "kernel_ms": dom_ms, "algorithmic_bytes_per_launch": algo_bytes,
                "rays_per_launch": dom_rays, "bytes_per_ray": round(algo_bytes / dom_rays, 1),
                "box_tests_per_ray": round(kc["box_tests"] / n_launch / dom_rays, 2),
                "tri_tests_per_ray": round(kc["tri_tests"] / n_launch / dom_rays, 2),
                "frame": {"gpu_ms": round(frame_ms, 4), "algorithmic_bytes": frame_bytes,
                          "achieved_GBs": round(frame_bytes / (frame_ms * 1e-3) / 1e9, 2), "kernels": kernels}}

    # Not the headline: the same frames with two in flight (a second scene handle and stream).  Every launch of the
    # pipeline ends on its longest rays with the chip draining; the next frame's bulk fills that in.  `value` above stays
    # the one-frame-at-a-time number the roofline figures belong to.
    pipelined = None
    if world == 1 and tile_world == 1 and args.frames_in_flight > 1:
        pipelined = frames_in_flight(rt, args, mesh, light, cfg, dev, W, H, rays_per_frame)

    # N > 1: the frame rank 0 gathered from all ranks' tiles must be the frame one GPU renders alone, bit for bit
    frame_verified = None
    if world > 1:
        rgb1, id1, w1 = torch.zeros_like(rgb), torch.zeros_like(hit_id), torch.zeros_like(hit_w)
        whole = rt._abi.Frame(rgb1.data_ptr(), None, id1.data_ptr(), None, w1.data_ptr())
        cam.render_device(whole, stream=stream, tile_rank=0, tile_world=1, tile_size=tiles.DEFAULT_TILE)
        torch.cuda.synchronize()
        frame_verified = bool(torch.equal(rgb1, rgb) and torch.equal(id1, hit_id) and torch.equal(w1, hit_w))

    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(args, verts, idx, cfg, light)

    out = {
        "metric": "Mrays/sec (primary+shadow) on Dragon 1080p",
        "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s stand-in (seeded procedural blob, %d faces) %dx%d, %d spp, Phong, primary+shadow"
                               % ("configs[1]: Dragon" if args.model == "dragon" and not args.faces else args.model, idx.shape[0], W, H, args.spp),
                   "rays_per_frame": int(rays_per_frame), "primary": int(primary), "shadow": int(shadow),
                   "tile": tiles.DEFAULT_TILE, "parallelism": "tiles%d" % world,
                   "hierarchy": {k: scene.object_info(0)[k] for k in ("n_nodes", "n_leaves", "n_leaf_slots", "max_depth")},
                   "scene_hbm_bytes": scene.device_bytes(), "scene_build_s": round(build_s, 2)},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    if pipelined:
        out["pipelined"] = pipelined
    if frame_verified is not None:
        out["config"]["gathered_frame_equals_single_gpu_frame"] = frame_verified
    if cpu:
        out["speedup_vs_cpu_baseline"] = round(mrays / cpu["value"], 1)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def frames_in_flight(rt, args, mesh, light, cfg, dev, W, H, rays_per_frame):
    """K frames with D in flight: D scene handles (each owns its scratch) on D streams, round robin, no host sync between."""
    D = args.frames_in_flight
    scenes = [rt.Scene([mesh], light, cfg["orientation"], rt.PhongLightingModel(light, [mesh]), cfg["fov"], device=dev.index) for _ in range(D)]
    cams = [rt.Camera(s, args.spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024) for s in scenes]
    streams = [torch.cuda.Stream(device=dev) for _ in range(D)]
    bufs = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(D)]
    frames = [rt._abi.Frame(b.data_ptr(), None, None, None, None) for b in bufs]

    def run(n):
        for k in range(n):
            cams[k % D].render_device(frames[k % D], stream=streams[k % D].cuda_stream)

    run(2 * D)
    torch.cuda.synchronize()
    n = max(args.steps, 2 * D)
    t0 = time.perf_counter()
    run(n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    same = all(bool(torch.equal(bufs[0], b)) for b in bufs[1:])
    for s in scenes:
        s.close()
    return {"frames_in_flight": D, "steps": n, "ms_per_step": round(dt * 1e3, 4), "value": round(rays_per_frame / dt / 1e6, 2),
            "unit": "Mrays/s", "frames_identical": same,
            "note": "not the headline value: frames overlap, so a frame's launch tails are filled by the next frame's bulk"}


def cpu_baseline(args, verts, idx, cfg, light):
    """The oracle in reference order (exhaustive both-children traversal), multithreaded, on a bounded
    sample of the same frame: every k-th pixel in row-major order, all of its samples and shadow rays."""
    import oracle
    import ray_tracer_amd as rt
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16"))))
    osc = oracle.OracleScene([(verts, idx, None, oracle.material(diffuse=(0.9, 0.9, 0.9)))])
    ocam = osc.camera(cfg["orientation"], cfg["fov"], rt.RTStatics.NEAR_PLANE, args.width, args.height)
    ost = oracle.settings(epsilon=osc.epsilon(), ms=args.spp, lighting=1, seed=2024, light_origin=light.origin,
                          light_emission=light.emission, light_ambient=light.ambient)
    # probe at a sparse stride, then size the timed sample to about --cpu-seconds of wall time on these cores:
    # a strided subset of the frame when the whole frame would take longer, else the whole frame, repeated
    stride = 257
    t0 = time.perf_counter()
    o = osc.render(ocam, ost, n_threads=cores, pixel_stride=stride, want_stats=True)
    dt = max(time.perf_counter() - t0, 1e-4)
    full_s = dt * stride
    stride = max(1, int(np.ceil(full_s / args.cpu_seconds)))
    if stride > 1 and stride % 2 == 0:
        stride += 1   # odd strides walk every column of an even-width image
    rays = 0
    repeats = 0
    t0 = time.perf_counter()
    while True:   # at least one pass over the sample, then more until about --cpu-seconds have gone
        o = osc.render(ocam, ost, n_threads=cores, pixel_stride=stride, want_stats=True)
        s = o["stats"]
        rays += s["primary_rays"] + s["shadow_rays"] + s["secondary_rays"]
        repeats += 1
        dt = time.perf_counter() - t0
        if dt * (repeats + 1) / repeats > args.cpu_seconds or repeats >= 64:
            break
    osc.close()
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "every %d-th pixel of the same %dx%d frame, %d spp + shadow rays, rendered %d time(s) (%d rays, %.1f s); "
                      "C restatement of the Java algorithm in reference visit order, %d pthreads"
                      % (stride, args.width, args.height, args.spp, repeats, rays, dt, cores),
            "java_present": bool(any(os.path.exists(os.path.join(p, "java")) for p in os.environ.get("PATH", "").split(":")))}


if __name__ == "__main__":
    main()
