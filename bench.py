#!/usr/bin/env python3
"""bench.py -- Mrays/s (primary + shadow) on the Dragon-class workload, 1920x1080, 4 spp, Phong.

A "step" is one frame: every primary sample of the 1080p image, one shadow ray per hit sample,
shaded, averaged and written.  The scene (flattened hierarchy) is resident in HBM before the timed
region; the framebuffer stays in HBM.  Rays are counted by the kernel, not assumed.

N > 1 GPUs, two ways, the same tiles and the same kernels either way (the frame's 16x16-pixel tiles interleaved over the
ranks, the scene replicated, every rank renders its tiles straight into one compact block, ONE exchange of the blocks
to rank 0 over RCCL plus one unpack kernel there; strong scaling: the frame is fixed):
  * `python bench.py --gpus N` (WORLD_SIZE unset): ONE process.  The scene gets the device list 0..N-1
    (mi355rt_scene_set_devices) and every frame is one library call, as the reference's fan-out is one
    Camera.createImage() call (Camera.java:129-182): per-device host threads, streams and the RCCL exchange are inside
    libmi355rt.so.  This is the drop-in's own path.
  * under torch.distributed.run (WORLD_SIZE set; the driver's launch), or with --spawn-ranks: one process per GPU over
    torch.distributed (backend nccl = RCCL), the blocks gathered with dist.gather.  Rank 0 then also times the
    one-process path over the same GPUs in a child process and reports it as `in_library`.
`--dry-run-gloo` rehearses exactly that launch and exchange on CPUs (gloo, library-shaped blocks, no
rendering): it is how the N-rank path is tested where there is no GPU.

Prints ONE JSON line on rank 0 (see the repository's bench contract) with extra objects:
  roofline        the dominant kernel: its algorithmic bytes per launch / its average launch duration,
                  from HIP events the library records around every kernel on the stream it launches on,
                  over the timed region, vs 8 TB/s
  cpu_baseline    the oracle (C restatement of the reference's algorithm, reference visit order) timed
                  on this host's cores -- all the process may use -- on a bounded sample of the same frame
  pipelined       the same frames with two in flight per rank (exchange of frame k under frame k + 1)
  blocking_render the drop-in seam itself: wall time of the blocking mi355rt_render into page-locked
                  host buffers (N = 1)
  soup            SURVEY 8(d)'s adversarial second input: a uniform triangle soup of the same face count
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
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

# <base dir> layout of the reference's README.txt:10-15 (RayTracer.java:117-140)
MODEL_FILES = {"bunny": "bunny/reconstruction/bun_zipper.ply", "dragon": "dragon_recon/dragon_vrip.ply",
               "buddha": "happy_recon/happy_vrip.ply", "lucy": "lucy.ply", "thai": "xyzrgb_statuette.ply/data"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="dragon", help="stand-in from meshgen.STANFORD")
    ap.add_argument("--faces", type=int, default=0, help="override the face count (0 = the model's own)")
    ap.add_argument("--models-dir", default="", help="base dir laid out as the reference's README.txt:10-15: render the real PLY instead of the stand-in")
    ap.add_argument("--via-ply", action="store_true",
                    help="write the stand-in as a big-endian binary PLY (what TriangleMesh.readBinary reads) and load it back through "
                         "mi355rt_ply_load: config.scene_create then has the parse time too")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--material", default="color", choices=["color", "brdf", "mirror-floor"],
                    help="color: ColorInformation(0.9) as RayTracer.java:122-123; brdf: BRDFMaterial(15) on the reference's CUReT tables "
                         "(configs[2]); mirror-floor: the mesh over a ReflectionMaterial floor (configs[3]: secondary rays)")
    ap.add_argument("--brdf-samples", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the blocking_render and soup objects (profiling runs)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="tuning aid (one GPU): render ONE rank's tiles of an N-rank run at a time, every rank in turn; the line carries the "
                         "per-rank times and their max (what an N-GPU frame would take before the exchange); not a bench result")
    ap.add_argument("--emulate-rank", type=int, default=-1, help="with --emulate-world: only this rank (-1: every rank in turn)")
    ap.add_argument("--tile", type=int, default=0, help="tile size of the sharding (0: tiles.DEFAULT_TILE)")
    ap.add_argument("--spawn-ranks", action="store_true", help="with --gpus N: one process per GPU over torch.distributed (the cross-check of the one-process path)")
    ap.add_argument("--in-library-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for the baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = every core this process may use)")
    ap.add_argument("--frames-in-flight", type=int, default=2, help="the 'pipelined' figure: frames in flight per rank; 1 = skip")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="CPU rehearsal of the N-rank launch and exchange: gloo, library-shaped blocks, no GPU, no rendering")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="with --gpus N: all N ranks on GPU 0 -- the one-process path with the device list [0] * N (peer copies), or with "
                         "--spawn-ranks N processes whose blocks travel over gloo through host memory: the whole N-rank code path where there "
                         "is one GPU; not a bench result")
    return ap.parse_args()


def launch_ranks(args):
    """The parent of an N-rank run.  It makes no torch.cuda / HIP call that initialises a GPU (device_count does not, on
    this image): a process that has touched the GPU must not be replaced or forked into rank processes."""
    n = args.gpus
    if not args.dry_run_gloo and not args.rehearse_one_gpu:
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d asked for, %d HIP device(s) visible: refusing to report a %d-GPU number from fewer GPUs\n"
                             % (n, have, n))
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    rank0_out = tempfile.TemporaryFile(mode="w+")   # a file, not a pipe: nothing can block on a full pipe while the parent polls
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if args.rehearse_one_gpu else str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=rank0_out if r == 0 else sys.stderr))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
        time.sleep(0.05)
    if failed is None:
        for r, p in enumerate(procs):
            if p.returncode != 0:
                failed = (r, p.returncode)
    if failed is not None:
        for p in procs:   # the exact processes started above, by handle
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write("bench.py: rank %d exited with status %d; no result\n" % failed)
        return 1
    rank0_out.seek(0)
    out = rank0_out.read()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        sys.stderr.write("bench.py: rank 0 printed %d JSON lines\n%s\n" % (len(lines), out))
        return 1
    print(lines[0])
    return 0


def dry_run_gloo(args, rank, world):
    """The N-rank path without GPUs: every rank fills its compact block (the layout the library's kernels write with
    settings.tile_compact) from one seeded synthetic frame, the blocks are gathered with gloo, rank 0 unpacks them and
    must hold the whole frame.  Frames in flight use the same double-buffered async path as the GPU run."""
    from ray_tracer_amd import tiles
    if world == 1 and "MASTER_PORT" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, T = args.width, args.height, tiles.DEFAULT_TILE
    g = torch.Generator().manual_seed(2024)
    full = (torch.rand(H, W, 3, generator=g), torch.randint(-1, 871414, (H, W), generator=g, dtype=torch.int32),
            torch.randint(-1, 2, (H, W), generator=g, dtype=torch.int32), torch.rand(H, W, generator=g))
    D = max(1, args.frames_in_flight)
    sh = tiles.TileSharder(W, H, rank, world, T, "cpu", n_buffers=D)
    outs = [[torch.full_like(t, -7) for t in full] for _ in range(D)]
    pending = [None] * D
    t0 = time.perf_counter()
    for k in range(max(args.steps, D)):
        b = k % D
        if pending[b]:
            pending[b]()
        tiles.tiles_pack_reference(full[0], full[1], full[2], full[3], rank, world, T, sh.blocks[b])   # stands in for the render
        pending[b] = sh.gather(*outs[b], buffer=b, async_op=True)
    for f in pending:
        if f:
            f()
    dist.barrier()
    dt = time.perf_counter() - t0
    ranks = dist.get_world_size()
    if rank == 0:
        same = all(torch.equal(a, b) for o in outs for a, b in zip(o, full))
        print(json.dumps({"metric": "Mrays/sec (primary+shadow) on Dragon 1080p", "value": None, "unit": "Mrays/s", "dry_run": True,
                          "backend": "gloo", "n_gpus": args.gpus, "ranks": ranks, "steps": args.steps, "frames_in_flight": D,
                          "gathered_frame_equals_single_gpu_frame": bool(same), "block_bytes_per_rank": 4 * tiles.PLANES * sh.n,
                          "exchange_s": round(dt, 3), "note": "launch + exchange rehearsal on CPUs: nothing was rendered or timed"}))
    dist.destroy_process_group()
    return 0


def load_model(rt, args):
    cfg = rt.meshgen.STANFORD[args.model]
    if args.models_dir:
        path = os.path.join(args.models_dir, MODEL_FILES[args.model])
        if not os.path.exists(path):
            raise SystemExit("bench.py: %s not found (--models-dir must be laid out as the reference's README.txt:10-15)" % path)
        v, i, n = rt.load_ply(path)
        return cfg, v, i, n, "%s (%s)" % (args.model, path)
    v, i = rt.meshgen.stanford_standin(args.model, args.faces or None)
    label = "configs[1]: Dragon" if args.model == "dragon" and not args.faces else args.model
    if args.via_ply:   # the loader's half of the path: TriangleMesh.readBinary's file format, parsed by the library's reader
        path = os.path.join(tempfile.gettempdir(), "mi355rt_bench_%s_%d.ply" % (args.model, os.getpid()))
        rt.meshgen.write_ply(path, v, i, binary=True)
        nbytes = os.path.getsize(path)
        t0 = time.time()
        v, i, _ = rt.load_ply(path)
        PLY_TIMES["ply_parse_s"] = round(rt.host.load_ply.last_parse_s, 3)          # mi355rt_ply_load itself
        PLY_TIMES["ply_load_with_numpy_copies_s"] = round(time.time() - t0, 3)   # + this wrapper's copies out of the library's arrays
        PLY_TIMES["ply_bytes"] = nbytes
        os.remove(path)
    return cfg, v, i, None, "%s stand-in (seeded procedural blob, %d faces)" % (label, i.shape[0])


PLY_TIMES = {}


def stats_of(ms):
    a = np.asarray(ms, dtype=np.float64)
    return {"mean": round(float(a.mean()), 4), "median": round(float(np.median(a)), 4), "min": round(float(a.min()), 4),
            "max": round(float(a.max()), 4)}


def build_scene(rt, args, device):
    """The workload's scene: (scene, cfg, verts, idx, normals, label, light, seconds)."""
    cfg, verts, idx, normals, label = load_model(rt, args)
    light = rt.Light((0.0, 100.0, 100.0), (0.3, 0.3, 0.3), (0.5, 0.5, 0.9))          # RayTracer.java:106
    if args.material == "brdf":
        from helpers import curet_tables   # tests/golden/curet_tables.npz: the reference's resources/table.txt + abrdf.dat, parsed
        rt.BRDFMaterial.setTables(*curet_tables())
        color = rt.BRDFMaterial(15, (0.0, 1.0, 0.0))               # RayTracer.java:162
    else:
        color = rt.ColorInformation((0.9, 0.9, 0.9))               # RayTracer.java:122-123
    mesh = rt.TriangleMesh((verts, idx) if normals is None else (verts, idx, normals), color, True)
    objects = [mesh]
    if args.material == "mirror-floor":   # a lone mirrored mesh reflects nothing (ReflectionMaterial.java:14-24 skips the object itself)
        fv, fi = rt.meshgen.quad(float(verts[:, 1].min()) - 0.01 * cfg["span"], 0.8 * cfg["span"])
        objects.append(rt.TriangleMesh((fv, fi), rt.ReflectionMaterial((0.8, 0.8, 0.8)), True))
    t0 = time.time()
    scene = rt.Scene(objects, light, cfg["orientation"], rt.PhongLightingModel(light, objects), cfg["fov"], device=device)
    return scene, cfg, verts, idx, normals, label, light, time.time() - t0


def main():
    args = parse_args()
    tile = args.tile or 16
    spawn = args.spawn_ranks or args.dry_run_gloo
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and spawn:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))      # torch.distributed ranks (processes)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    lib_ranks = args.gpus if world == 1 else 1           # ranks inside the library (ONE process: mi355rt_scene_set_devices)
    if args.in_library_child:
        world, rank, local_rank, lib_ranks = 1, 0, 0, args.gpus
    if world > 1 and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` or with --nproc-per-node N" % (args.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_run_gloo:
        sys.exit(dry_run_gloo(args, rank, world))
    if lib_ranks > 1 and not args.rehearse_one_gpu and torch.cuda.device_count() < lib_ranks:
        raise SystemExit("bench.py: --gpus %d asked for, %d HIP device(s) visible: refusing to report a %d-GPU number from fewer GPUs"
                         % (lib_ranks, torch.cuda.device_count(), lib_ranks))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: libmi355rt has no CPU path")
    if args.rehearse_one_gpu and world > 1:
        local_rank = 0   # (every rank of the rehearsal on the one GPU, also when torch.distributed.run numbered them)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rehearse = args.rehearse_one_gpu and world > 1
    cdev = torch.device("cpu") if rehearse else dev   # where the few scalars that travel between ranks live
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: RCCL sees %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    import ray_tracer_amd as rt
    from ray_tracer_amd import tiles

    scene, cfg, verts, idx, normals, label, light, build_s = build_scene(rt, args, local_rank)
    W, H = args.width, args.height
    cam = rt.Camera(scene, args.spp, args.brdf_samples, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024)

    rgb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    hit_id = torch.zeros((H, W), dtype=torch.int32, device=dev)
    hit_obj = torch.zeros((H, W), dtype=torch.int32, device=dev)
    hit_w = torch.zeros((H, W), dtype=torch.float32, device=dev)
    frame = rt._abi.Frame(rgb.data_ptr(), None, hit_id.data_ptr(), hit_obj.data_ptr(), hit_w.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream   # the kernel is launched on torch's current stream

    if args.emulate_world > 1 and world == 1 and lib_ranks == 1:
        print(json.dumps(emulate_shares(rt, tiles, args, scene, cam, frame, stream, dev, label, tile)))
        return

    # the frame ONE GPU renders alone: what every N-GPU frame below must equal bit for bit
    single = None
    if lib_ranks > 1 or (world > 1 and rank == 0):
        single = [torch.zeros_like(t) for t in (rgb, hit_id, hit_obj, hit_w)]
        whole = rt._abi.Frame(single[0].data_ptr(), None, single[1].data_ptr(), single[2].data_ptr(), single[3].data_ptr())
        cam.render_device(whole, stream=stream, tile_size=tile)
        torch.cuda.synchronize()
    multi = None
    if lib_ranks > 1:
        scene.setDevices([0] * lib_ranks if args.rehearse_one_gpu else list(range(lib_ranks)))
        multi = scene.multi_info()

    sharder = tiles.TileSharder(W, H, rank, world, tile, dev, via_host=rehearse)
    target = frame if world == 1 else sharder.blocks[0].frame()   # processes: this rank's tiles, compact, ready to send

    def step(stats=False, times=False):
        if world > 1:
            cam.render_device(target, stream=stream, tile_rank=rank, tile_world=world, tile_size=tile, collect_stats=stats, collect_times=times,
                              tile_compact=True)
        else:   # one GPU, or the library's own ranks behind the one call
            cam.render_device(target, stream=stream, tile_size=tile, collect_stats=stats, collect_times=times)

    def exchange():
        sharder.gather(rgb, hit_id, hit_obj, hit_w)

    # counted, not assumed: one instrumented frame
    step(stats=True)
    torch.cuda.synchronize()
    st = cam.frame_stats()
    kcounts = cam.kernel_profile(reset=True)   # per-walker counts of that frame (the library's own ranks: rank 0's)
    counts = torch.tensor([st["primary_rays"], st["shadow_rays"], st["secondary_rays"], st["box_tests"], st["tri_tests"],
                           st["hits"]], dtype=torch.float64, device=cdev)
    local_counts = counts.clone()
    if world > 1:
        dist.all_reduce(counts)
    primary, shadow, secondary, boxes, tris, hits = [float(x) for x in counts.tolist()]
    rays_per_frame = primary + shadow + secondary

    for _ in range(args.warmup):
        step(times=True)
        exchange()
    torch.cuda.synchronize()
    cam.kernel_profile(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    lib_rank_ms = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step(times=True)
        ev[k][1].record()
        exchange()
        ev[k][2].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    render_ms = [a.elapsed_time(b) for a, b, _ in ev]     # all kernels of one frame, this rank (the library's ranks: the whole frame, exchange and unpack included)
    step_ms = [a.elapsed_time(c) for a, _, c in ev]       # with this rank's part of the exchange
    frame_ms = float(np.mean(render_ms))
    kprof = cam.kernel_profile()
    rank_ms = [frame_ms]
    if lib_ranks > 1:
        rank_ms = scene.multi_info()["render_ms"]          # each rank's kernels of the last frame (HIP events on the rank's stream)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine = torch.tensor([frame_ms], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t.item()) for t in every]
    ms_per_step = elapsed / args.steps * 1e3
    mrays = rays_per_frame / (elapsed / args.steps) / 1e6

    # N > 1: the frame gathered from all ranks' tiles must be the frame one GPU renders alone, bit for bit
    frame_verified = None
    if single is not None:
        frame_verified = all(bool(torch.equal(a.view(torch.int32), b.view(torch.int32))) for a, b in zip(single, (rgb, hit_id, hit_obj, hit_w)))

    # Not the headline: the same frames with two in flight per rank (views of the scene on their own streams); the
    # exchange of frame k runs under the render of frame k + 1.  Every launch of the pipeline ends on its longest rays
    # with the chip draining; the next frame's bulk fills that in.  All ranks take part (the gathers are collectives).
    pipelined = None
    if args.frames_in_flight > 1:
        try:
            pipelined = frames_in_flight(rt, tiles, args, scene, dev, rank, world, world, W, H, rays_per_frame, rehearse, tile)
        except Exception as e:   # the extra figure must not cost the run its headline line
            pipelined = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.synchronize()
    if pipelined and "_frames_equal" in pipelined:
        eq = pipelined.pop("_frames_equal")
        if single is not None:
            frame_verified = frame_verified and eq(single)

    if rank != 0:   # every collective is behind us: the rest is rank 0's bookkeeping
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
    own_pixels = float((tiles.tile_pixels(W, H, tiles.owned_tiles(W, H, rank, world, tile), tile) >= 0).sum().item())
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
    dom_hits = (lh / lib_ranks) if dom == "walk_primary" else 0.0
    algo_bytes = (BYTES_PER_BOX_TEST * kc["box_tests"] + BYTES_PER_TRI_TEST * kc["tri_tests"] + BYTES_PER_HIT * dom_hits) / n_launch
    dom_ms = kernels[dom]["ms_per_launch"]
    achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
    measured = measured_counters(args, W, H, world * lib_ranks, dom)
    traffic = measured.get("hbm_bytes_per_launch")
    dom_rays = max(kc["rays"] / n_launch, 1.0)
    exact_shadow = dom == "walk_shadow" and not shadow_any_hit_proven(scene, light)
    frac = achieved / HBM_PEAK_GBS
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(frac, 4), "traffic": traffic,
                # What bounds the kernel in fact (profiles/, PMC): the small-leaf walker issues vector instructions 90 % of the
                # launch at 39 % lane use; `frac` above is SURVEY 8(d)'s ALGORITHMIC bytes over time, which L2 hits let exceed
                # what HBM moves (big-leaf scenes: even 1.0) -- the memory side's own figure is hbm_measured_frac.
                "limiter": measured.get("limiter"), "valu_busy": measured.get("valu_busy"), "lane_util": measured.get("lane_util"),
                # (the counters' word; round 3's A/B runs qualify it for the small-leaf walker -- a tenth fewer vector
                # instructions in its leaf step, a fifth fewer L1 accesses in its node step, its two fetches overlapped: no
                # gain from any, so no single unit is the limit; L1 0.58 accesses per cycle and CU + 0.34 stalled)
                "limiter_note": "DESIGN.md section 6; profiles/r03_dragon_l1_counters.txt" if measured.get("limiter") == "valu-issue" and fused else None,
                "traffic_source": measured.get("source"),
                # the memory side's own figure, from the PMC counters of profiles/ (per launch, when a profile of this workload
                # and of these kernel sources is there): what the kernel really moves from HBM -- the rest are L2 / MALL hits
                "hbm_measured_GBs": round(traffic / (dom_ms * 1e-3) / 1e9, 2) if traffic else None,
                "hbm_measured_frac": round(traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                "kernel": "walk_kernel<0,MODE_FUSED> (primary rays, each hit's shadow ray walked by the same lane)" if fused else
                          {"walk_primary": "walk_kernel<0,MODE_CLOSEST> (primary rays)",
                           "walk_shadow": "walk_kernel<0,MODE_SHADOW> (shadow rays, the reference's closest-hit form)" if exact_shadow
                                          else "walk_kernel<0,MODE_SHADOW_ANY> (shadow rays)"}.get(dom, dom),
                "kernel_ms": dom_ms, "algorithmic_bytes_per_launch": algo_bytes,
                "rays_per_launch": dom_rays, "bytes_per_ray": round(algo_bytes / dom_rays, 1),
                "box_tests_per_ray": round(kc["box_tests"] / n_launch / dom_rays, 2),
                "tri_tests_per_ray": round(kc["tri_tests"] / n_launch / dom_rays, 2),
                "frame": {"gpu_ms": round(frame_ms, 4), "algorithmic_bytes": frame_bytes,
                          "achieved_GBs": round(frame_bytes / (frame_ms * 1e-3) / 1e9, 2), "kernels": kernels}}
    if frac > 1.0:
        roofline["note"] = ("algorithmic bytes per second exceed the HBM peak: SURVEY 8(d)'s byte model counts every box and triangle test as a "
                            "fetch, and this kernel's neighbouring rays share them in L2; not a fraction of the roofline -- see hbm_measured_frac")

    blocking = soup = None
    if world == 1 and not args.no_extras:
        blocking = blocking_render(rt, cam, rays_per_frame)
        if lib_ranks == 1 and args.model == "dragon" and not args.models_dir and args.material == "color":
            soup = soup_line(rt, args, light, cfg, dev, W, H)

    cpu = None
    if not args.no_cpu_baseline and args.material == "color" and not args.in_library_child and world == 1 and lib_ranks == 1:
        cpu = cpu_baseline(args, verts, idx, normals, cfg, light)   # (N = 1 only: a reported baseline, timed once per round)

    in_library = None
    if world > 1 and not rehearse:
        in_library = in_library_child(args)

    n_total = world * lib_ranks
    out = {
        "metric": "Mrays/sec (primary+shadow) on Dragon 1080p",
        "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": n_total, "ranks": dist.get_world_size() if world > 1 else lib_ranks,
        "processes": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" if not args.models_dir else "PLY file",
        "config": {"workload": "%s %dx%d, %d spp, Phong, primary+shadow%s" % (label, W, H, args.spp, {"color": "", "brdf": ", BRDFMaterial(15) on the CUReT tables, brdf_samples %d" % args.brdf_samples, "mirror-floor": ", over a ReflectionMaterial floor (secondary rays)"}[args.material]),
                   "rays_per_frame": int(rays_per_frame), "primary": int(primary), "shadow": int(shadow), "secondary": int(secondary),
                   "tile": tile, "parallelism": "tiles%d" % n_total,
                   "multi_gpu_path": ("one process: mi355rt_scene_set_devices, exchange inside the library (%s)" % multi["transport"]) if lib_ranks > 1
                                     else ("one process per GPU: torch.distributed gather (%s)" % ("gloo" if rehearse else "nccl = RCCL") if world > 1 else None),
                   "hierarchy": {k: scene.object_info(0)[k] for k in ("n_nodes", "n_leaves", "n_leaf_slots", "max_depth")},
                   "scene_hbm_bytes": scene.device_bytes(), "scene_build_s": round(build_s, 2),
                   # where Scene() spent it (mi355rt_scene_build_seconds; TriangleMesh.java:84-85,123-124 and KDTree.java:49-55 print the reference's own)
                   "scene_create": dict(scene.build_seconds(), total_s=round(build_s, 3), **PLY_TIMES)},
        # per step, from HIP events on this rank's stream: the frame's kernels alone, and with this rank's part of the exchange
        "step_gpu_ms": {"render": stats_of(render_ms), "render_and_exchange": stats_of(step_ms)},
        "rank_render_ms": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4), "all": [round(x, 4) for x in rank_ms]},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    if multi:
        out["devices"] = multi["devices"]
        out["transport"] = multi["transport"]
    if pipelined:
        out["pipelined"] = pipelined
    if blocking:
        out["blocking_render"] = blocking
    if soup:
        out["soup"] = soup
    if in_library:
        out["in_library"] = in_library
    if frame_verified is not None:
        out["gathered_frame_equals_single_gpu_frame"] = frame_verified
    if args.rehearse_one_gpu and n_total > 1:
        out["rehearsal"] = "all %d ranks on one GPU (%s): the N-rank code path, not a bench result" % (
            n_total, "blocks exchanged over gloo through host memory" if rehearse else "device list [0] * N, peer copies")
    if cpu:
        out["speedup_vs_cpu_baseline"] = round(mrays / cpu["value"], 1)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def in_library_child(args, limit_s=240):
    """Under torch.distributed.run (one process per GPU) rank 0 also times the drop-in's own path -- ONE process, the device
    list inside the library -- over the same GPUs, in a child process with a time limit (a fault or hang there must not
    cost the run its line).  The other ranks have finished their frames; their processes only wait to exit."""
    cmd = [sys.executable, "-u", os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--model", args.model, "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp), "--material", args.material,
           "--no-cpu-baseline", "--no-extras", "--in-library-child"]
    if args.faces:
        cmd += ["--faces", str(args.faces)]
    if args.tile:
        cmd += ["--tile", str(args.tile)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                            "ROLE_RANK", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit_s, text=True)
    except subprocess.TimeoutExpired:
        return {"error": "no result within %d s" % limit_s}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        return {"error": "exit status %d" % r.returncode, "stderr_tail": r.stderr[-600:]}
    j = json.loads(lines[0])
    keep = ("value", "unit", "ms_per_step", "n_gpus", "ranks", "processes", "devices", "transport", "rank_render_ms", "step_gpu_ms",
            "gathered_frame_equals_single_gpu_frame", "pipelined")
    res = {k: j[k] for k in keep if k in j}
    res["path"] = j["config"]["multi_gpu_path"]
    return res


def emulate_shares(rt, tiles, args, scene, cam, frame, stream, dev, label, tile):
    """One GPU renders each rank's tiles of an N-rank frame in turn: per-rank times, their max (what the N-GPU frame takes
    before its exchange step: the frame is done when its slowest rank is), and the same with two frames in flight."""
    N = args.emulate_world
    W, H = args.width, args.height
    ranks = [args.emulate_rank] if args.emulate_rank >= 0 else list(range(N))
    cam.render_device(frame, stream=stream, tile_size=tile, collect_stats=True)
    torch.cuda.synchronize()
    st = cam.frame_stats()
    rays = st["primary_rays"] + st["shadow_rays"] + st["secondary_rays"]
    one, rays_rank, sustained, kernels = [], [], [], []
    pipe_streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.frames_in_flight))]   # the same streams for every rank's pipeline
    for r in ranks:
        cam.render_device(frame, stream=stream, tile_rank=r, tile_world=N, tile_size=tile, collect_stats=True)
        torch.cuda.synchronize()
        sr = cam.frame_stats()
        rays_rank.append(int(sr["primary_rays"] + sr["shadow_rays"] + sr["secondary_rays"]))
        for _ in range(max(2, args.warmup)):
            cam.render_device(frame, stream=stream, tile_rank=r, tile_world=N, tile_size=tile)
        torch.cuda.synchronize()
        ms = []
        for _ in range(args.steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            cam.render_device(frame, stream=stream, tile_rank=r, tile_world=N, tile_size=tile)
            b.record()
            torch.cuda.synchronize()   # one frame at a time
            ms.append(a.elapsed_time(b))
        one.append(float(np.median(ms)))
        cam.kernel_profile(reset=True)   # where a share's time goes: its kernels (HIP events around each launch), the rest is launch gaps
        for _ in range(3):
            cam.render_device(frame, stream=stream, tile_rank=r, tile_world=N, tile_size=tile, collect_times=True)
        torch.cuda.synchronize()
        kp = cam.kernel_profile(reset=True)
        kernels.append({k: round(v["ms"] / 3.0, 4) for k, v in kp.items() if v["launches"]})
        if args.frames_in_flight > 1:
            D = args.frames_in_flight
            pipe = tiles.FramePipeline(scene, lambda sc: rt.Camera(sc, args.spp, args.brdf_samples, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024),
                                       W, H, r, 1, D, tile, dev, tile_world=N, streams=pipe_streams)
            for _ in range(2 * D):
                pipe.submit()
            pipe.drain()
            n = max(args.steps, 2 * D)
            t0 = time.perf_counter()
            for _ in range(n):
                pipe.submit()
            pipe.drain()
            sustained.append((time.perf_counter() - t0) / n * 1e3)
            pipe.close()
    res = {"emulated": True, "note": "one GPU rendering each rank's tiles of an N-rank frame in turn, before the exchange step: not a bench result",
           "workload": "%s %dx%d, %d spp" % (label, W, H, args.spp), "world": N, "tile": tile, "ranks_measured": ranks, "rays_per_frame": int(rays),
           "rank_rays": rays_rank, "rank_ms": [round(x, 4) for x in one], "max_ms": round(max(one), 4), "mean_ms": round(float(np.mean(one)), 4),
           "rank_kernel_ms": kernels}
    if sustained:
        res.update(frames_in_flight=args.frames_in_flight, rank_sustained_ms=[round(x, 4) for x in sustained], max_sustained_ms=round(max(sustained), 4))
    return res


def measured_counters(args, W, H, n, kernel):
    """The PMC figures of profiles/hbm_traffic.json for this workload and kernel -- HBM bytes per launch, VALU busy, lane use --
    when that profile was taken from the kernel sources this library was built from (their hash is in the file); else
    nothing: the line must not carry another build's counters as this run's."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            tj = json.load(f)
        entry = tj.get("%s_%dx%d_%dspp_n%d" % (args.model, W, H, args.spp, n), {})
        if args.material != "color" or args.faces or args.models_dir:
            return {}
        if entry.get("kernel_sources_sha1") != kernel_sources_hash():
            return {"source": "none: profiles/hbm_traffic.json was taken from other kernel sources (%s)" % str(entry.get("kernel_sources_sha1"))[:12]}
        k = dict(entry.get(kernel, {}))
        if k:
            k["source"] = "profiles/hbm_traffic.json (%s; rocprofv3 --pmc passes of tools/profile_round.sh on another box, same kernel sources %s): not measured in this run" % (
                entry.get("taken", "?"), kernel_sources_hash()[:12])
        return k
    except Exception:
        return {}


def kernel_sources_hash():
    import hashlib
    h = hashlib.sha1()
    for name in ("trace_kernels.hip", "device_util.h", "layout.h", "walk_args.h"):
        with open(os.path.join(ROOT, "ray-tracer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def shadow_any_hit_proven(scene, light):
    """Whether the frame's shadow rays ran the any-hit form (far light) or the reference's closest-hit form: mirrors the
    library's own test only for the label in the JSON line."""
    mm = scene.object_info(0)["minmax"]
    lo, hi = mm[:3], mm[3:]
    diag = float(np.linalg.norm(hi - lo))
    d = np.maximum(np.maximum(lo - np.array(light.origin), np.array(light.origin) - hi), 0.0)
    return float(np.linalg.norm(d)) > 1.01 * diag


def frames_in_flight(rt, tiles, args, scene, dev, rank, world, tile_world, W, H, rays_per_frame, via_host=False, tile=16):
    """K frames with D in flight on every rank (tiles.FramePipeline): D views of the scene (mi355rt_scene_view: shared
    arrays, own scratch) on D streams, round robin, no host sync between; with N > 1 each frame's block is gathered
    asynchronously and unpacked on rank 0 when its slot comes round again, so the exchange of frame k runs under the
    render of frame k + 1."""
    D = args.frames_in_flight
    pipe = tiles.FramePipeline(scene, lambda sc: rt.Camera(sc, args.spp, args.brdf_samples, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024),
                               W, H, rank, world, D, tile, dev, tile_world=tile_world, via_host=via_host)
    full = rank == 0
    for _ in range(2 * D):
        pipe.submit()
    pipe.drain()
    if world > 1:
        dist.barrier()
    n = max(args.steps, 2 * D)
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.submit()
    pipe.drain()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=torch.device("cpu") if via_host else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    dt /= n
    outs = pipe.outs
    same = all(bool(torch.equal(a, b)) for o in outs[1:] for a, b in zip(o, outs[0])) if (full or world == 1) else None
    res = {"frames_in_flight": D, "steps": n, "ms_per_step": round(dt * 1e3, 4), "value": round(rays_per_frame / dt / 1e6, 2),
           "unit": "Mrays/s", "frames_identical": same, "views_of_one_scene": True,
           "note": "not the headline value: frames overlap, so a frame's launch tails (and, with N > 1, its exchange) are filled by the next frame's bulk"}
    if full:
        keep = [t.clone() for t in outs[0]]
        res["_frames_equal"] = lambda single: all(bool(torch.equal(a.view(torch.int32), b.view(torch.int32))) for a, b in zip(single, keep))
    pipe.close()
    return res


def blocking_render(rt, cam, rays_per_frame, n=10):
    """Wall time of the drop-in call itself, Camera.createImage() -> mi355rt_render_host: the frame's kernels, the
    device-to-host copies into ONE page-locked set of arrays (as Camera.pixels is one array), the host mirror's listener
    event.  Three sets of outputs: everything the parity tests compare (rgb + hit_id + hit_obj + hit_w), Camera.pixels alone
    (rgb), and Camera.image alone (the 8-bit ARGB image, written by the frame's own kernels: a third of rgb's bytes)."""
    cam.pinOutput(True)
    res = {}
    try:
        for key, outputs, bpp in (("all", ("pixels", "hit_id", "hit_obj", "hit_w"), 24), ("rgb_only", ("pixels",), 12), ("argb8_only", ("image",), 4)):
            cam.createImage(outputs=outputs)
            ms = []
            for _ in range(n):
                t0 = time.perf_counter()
                cam.createImage(outputs=outputs)
                ms.append((time.perf_counter() - t0) * 1e3)
            med = float(np.median(ms))
            res[key] = {"ms": stats_of(ms), "value": round(rays_per_frame / (med * 1e-3) / 1e6, 2), "bytes_to_host": cam.screenWidth * cam.screenHeight * bpp}
    finally:
        cam.pinOutput(False)
    out = {"call": "mi355rt_render_host (blocking, host outputs page-locked)", "frames": n, "unit": "Mrays/s",
           "note": "PCIe-inclusive: never the headline value"}
    out.update(res["all"])   # (ms / value / bytes_to_host of the full set at the top level, as in round 2's line)
    out["rgb_only"] = res["rgb_only"]
    out["argb8_only"] = res["argb8_only"]
    return out


def soup_line(rt, args, light, cfg, dev, W, H, steps=5):
    """SURVEY 8(d)'s adversarial second input: a uniform random triangle soup of the same face count in the same box, same
    camera, light and settings -- no surface coherence, so rays walk many more leaves before anything stops them."""
    v, i = rt.meshgen.soup(rt.meshgen.STANFORD["dragon"]["faces"], span=cfg["span"], tri_size=0.004, seed=11)
    mesh = rt.TriangleMesh((v, i), rt.ColorInformation((0.9, 0.9, 0.9)), True)
    scene = rt.Scene([mesh], light, cfg["orientation"], rt.PhongLightingModel(light, [mesh]), cfg["fov"], device=dev.index)
    cam = rt.Camera(scene, args.spp, 1, rt.RTStatics.NEAR_PLANE, W, H, rng_seed=2024)
    rgb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    frame = rt._abi.Frame(rgb.data_ptr(), None, None, None, None)
    cam.render_device(frame, collect_stats=True)
    torch.cuda.synchronize()
    st = cam.frame_stats()
    rays = st["primary_rays"] + st["shadow_rays"] + st["secondary_rays"]
    for _ in range(2):
        cam.render_device(frame)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        cam.render_device(frame)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    info = scene.object_info(0)
    scene.close()
    return {"workload": "uniform triangle soup, %d faces (edge %.3g of the span), same box / camera / light / %dx%d / %d spp"
                        % (i.shape[0], 0.004, W, H, args.spp),
            "steps": steps, "ms_per_step": round(dt * 1e3, 4), "value": round(rays / dt / 1e6, 2), "unit": "Mrays/s",
            "rays_per_frame": int(rays), "box_tests_per_ray": round(st["box_tests"] / rays, 1), "tri_tests_per_ray": round(st["tri_tests"] / rays, 1),
            "hierarchy": {k: info[k] for k in ("n_nodes", "n_leaves", "n_leaf_slots", "max_depth")}}


def usable_cores():
    """Cores this process may use: its affinity mask, cut down by a cgroup CPU quota where one is set."""
    affinity = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    return affinity, quota


def cpu_baseline(args, verts, idx, normals, cfg, light):
    """The oracle in reference order (exhaustive both-children traversal), multithreaded, on a bounded
    sample of the same frame: every k-th pixel in row-major order, all of its samples and shadow rays.
    Threads: every core the process may use (the reference starts availableProcessors() threads, Camera.java:129)."""
    import oracle
    import ray_tracer_amd as rt
    affinity, quota = usable_cores()
    threads = args.cpu_threads or max(1, min(affinity, int(np.ceil(quota)) if quota else affinity))
    osc = oracle.OracleScene([(verts, idx, normals, oracle.material(diffuse=(0.9, 0.9, 0.9)))])
    ocam = osc.camera(cfg["orientation"], cfg["fov"], rt.RTStatics.NEAR_PLANE, args.width, args.height)
    ost = oracle.settings(epsilon=osc.epsilon(), ms=args.spp, lighting=1, seed=2024, light_origin=light.origin,
                          light_emission=light.emission, light_ambient=light.ambient)
    # probe at a sparse stride, then size the timed sample to about --cpu-seconds of wall time on these cores:
    # a strided subset of the frame when the whole frame would take longer, else the whole frame, repeated
    stride = 257
    t0 = time.perf_counter()
    o = osc.render(ocam, ost, n_threads=threads, pixel_stride=stride, want_stats=True)
    dt = max(time.perf_counter() - t0, 1e-4)
    full_s = dt * stride
    stride = max(1, int(np.ceil(full_s / args.cpu_seconds)))
    if stride > 1 and stride % 2 == 0:
        stride += 1   # odd strides walk every column of an even-width image
    rays = 0
    repeats = 0
    t0 = time.perf_counter()
    while True:   # at least one pass over the sample, then more until about --cpu-seconds have gone
        o = osc.render(ocam, ost, n_threads=threads, pixel_stride=stride, want_stats=True)
        s = o["stats"]
        rays += s["primary_rays"] + s["shadow_rays"] + s["secondary_rays"]
        repeats += 1
        dt = time.perf_counter() - t0
        if dt * (repeats + 1) / repeats > args.cpu_seconds or repeats >= 64:
            break
    osc.close()
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "threads": threads, "cores_available": affinity,
            "cgroup_cpu_quota": quota, "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS"), "kind": "port",
            "sample": "every %d-th pixel of the same %dx%d frame, %d spp + shadow rays, rendered %d time(s) (%d rays, %.1f s); "
                      "C restatement of the Java algorithm in reference visit order, %d pthreads (all the cores the process may use: "
                      "affinity %d, cgroup quota %s)"
                      % (stride, args.width, args.height, args.spp, repeats, rays, dt, threads, affinity, quota),
            "java_present": bool(any(os.path.exists(os.path.join(p, "java")) for p in os.environ.get("PATH", "").split(":")))}


if __name__ == "__main__":
    main()
